"""The drop-in boundary is a C ABI: a host written in plain C (tests/c_abi/smoke.c — no torch, no Python) is built
against include/tdm_hip.h + libtdm_hip.so and run on the GPU."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_program(tmp_path):
    gcc = shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc / ROCm headers not available on this box")
    libdir = os.path.join(ROOT, "tinydiffusionmodels_amd", "csrc")
    exe = str(tmp_path / "smoke")
    build = subprocess.run([gcc, "-std=c11", "-O1", os.path.join(ROOT, "tests", "c_abi", "smoke.c"),
                            "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
                            "-L", libdir, "-ltdm_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
                            "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe],
                           capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(run.stdout)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "q_sample: 0 mismatching" in run.stdout and "C ABI smoke OK" in run.stdout
