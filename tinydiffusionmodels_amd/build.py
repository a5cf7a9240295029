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
          # No packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) anywhere in the library, by construction:
          # the device target feature is switched off, so neither clang's SLP vectoriser nor the backend's selection of explicit
          # ext_vector_type arithmetic can emit one (the host pass prints "not a recognized feature" and ignores it), and
          # -fno-slp-vectorize keeps the vectoriser from building the v_mov-assembled register pairs in the first place.
          # Why: on gfx950 `v_pk_add_f32 D, A, B op_sel:[0,1]` (low result taken from the HIGH dword of B) comes
          # back, intermittently, WITHOUT the B term in lanes 48-63 of the low result while the library's token-major GEMM runs on
          # another stream - the instruction alone reproduces it (tools/micro/pk_waw.hip, profiles/r05_pk_opsel_probe.txt), and
          # every wrong row of round 4's SLP-built LayerNorm backward traces to one of its three uses of that form
          # (tools/ln_slp_forensics.py, profiles/r05_ln_slp_forensics.txt).  tests/test_host_logic.py::
          # test_no_packed_fp32_in_any_code_object disassembles every code object and holds this comment to the shipped bits
          # (no v_pk_*_f32, no op_sel half-select of any kind).
          "-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
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


def source_digest() -> str:
    """sha256 (16 hex digits) over the library's sources, headers and compiler flags: what a measurement file under profiles/
    records so that a later bench run can tell whether the numbers were taken on the code it is running."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(_sources() + [d for d in _deps() if not d.endswith("build.py")]):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(f for f in CFLAGS if not os.path.isabs(f)).encode())   # (flags, not the checkout's absolute include paths)
    return h.hexdigest()[:16]


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
