"""Build libtdm_hip.so (gfx950) with hipcc, in-tree.

    python -m tinydiffusionmodels_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  Objects and the library live
next to the sources (git-ignored, but they travel to the GPU box)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libtdm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", INCLUDE, "-I", CSRC,
          "-Wno-unused-result", "-DNDEBUG",
          # HIP defaults to -ffp-contract=fast; the bit-exact elementwise kernels (q_sample, p_sample update,
          # AdamW) must round every mul/add separately like the reference's ATen ops. FMAs are explicit (fmaf).
          "-ffp-contract=off",
          # No SLP vectorisation: on gfx950 / ROCm 7.2 the packed-fp32 code it produced for the LayerNorm backward (v_pk_mul_f32 /
          # v_pk_add_f32 over freshly loaded register pairs) gave DIFFERENT results whenever another kernel stream competed for
          # the GPU — a few rows off by 1e-4 relative, different from run to run (tools/contention_ops.py; quiet runs were always
          # right).  Without it every launch is bit-stable next to foreign work, results are unchanged bit for bit, and the
          # guide's measurement (packed f32 VALU beside MFMAs is an anti-lever) says nothing is lost.
          "-fno-slp-vectorize"]
# TDM_BUILD_DEFINES="-DTDM_DIAG": the diagnostic build (runtime ablation bits / phase probes in the hot kernels, tools/ only)
CFLAGS += os.environ.get("TDM_BUILD_DEFINES", "").split()


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    d.append(os.path.join(INCLUDE, "tdm_hip.h"))
    d.append(os.path.abspath(__file__))   # compiler flags live here
    return d


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    mt = os.path.getmtime(target)
    return any(os.path.getmtime(d) > mt for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    srcs = _sources()
    hdrs = _deps()
    todo = []
    objs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            todo.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC, *CFLAGS, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        return o

    if todo:
        with ThreadPoolExecutor(max_workers=min(4, len(todo))) as ex:
            list(ex.map(cc, todo))
    if todo or force or _stale(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs, "-ldl"]   # comm.hip binds RCCL via dlopen
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
