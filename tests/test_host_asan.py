"""The host half of the C-ABI library under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU (SURVEY.md section 5:
sanitizers run on the CPU build only; GPU ASan is not available on the pool).  csrc/*.hip are compiled HOST-ONLY
(`--cuda-host-only`: a second or two per file, no device code) with `-fsanitize=address,undefined`, linked into a scratch
library, and tests/c_abi/host_args.c — argument validation of the launching entry points, the size / layout queries, the
host evaluations of the dropout hash and Philox, with exact-size host buffers — runs against it."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tinydiffusionmodels_amd", "csrc")


def test_host_half_under_asan_ubsan(tmp_path):
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    hipcc = shutil.which("hipcc") or os.path.join(rocm, "bin", "hipcc")
    clang = os.path.join(rocm, "lib", "llvm", "bin", "clang")
    if not (os.path.exists(hipcc) and os.path.exists(clang)):
        pytest.skip("hipcc / the ROCm clang are not available on this box")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    objs = []
    jobs = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        obj = str(tmp_path / (os.path.basename(src)[:-4] + ".o"))
        objs.append(obj)
        jobs.append(subprocess.Popen([hipcc, "--cuda-host-only", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-DNDEBUG",
                                      "-ffp-contract=off", "-Wno-unused-result", *san, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                                      "-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for j in jobs:
        out, _ = j.communicate(timeout=600)
        assert j.returncode == 0, out[-3000:]
    # a host-only object still references its translation unit's device image (`__hip_fatbin_<hash>`, registered lazily and only
    # parsed at the first launch): an empty blob per object satisfies the loader; nothing here ever launches successfully
    nm = subprocess.run(["nm", "-u", *objs], capture_output=True, text=True, timeout=60)
    blobs = sorted({tok for line in nm.stdout.splitlines() for tok in line.split() if tok.startswith("__hip_fatbin_")})
    stub = tmp_path / "fatbin_stub.c"
    stub.write_text("".join(f"const char {b}[64] __attribute__((aligned(4096))) = {{0}};\n" for b in blobs))
    objs.append(str(stub))
    lib = str(tmp_path / "libtdm_host_asan.so")
    link = subprocess.run([hipcc, "-shared", "-fPIC", *san, "-o", lib, *objs, "-ldl"], capture_output=True, text=True, timeout=600)
    assert link.returncode == 0, link.stderr[-3000:]
    exe = str(tmp_path / "host_args")
    build = subprocess.run([clang, "-std=c11", *san, os.path.join(ROOT, "tests", "c_abi", "host_args.c"), "-I", os.path.join(ROOT, "include"),
                            lib, "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe],
                           capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    print(run.stdout[-4000:])
    assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-4000:]
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "host args OK" in run.stdout
