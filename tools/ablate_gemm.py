"""Timing ablation of the NT bf16x3 GEMM (diagnostic; outputs are wrong when a stage is skipped)."""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
def run(M, N, K):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) * 0.05
    bias = torch.zeros(N, device=dev); C = torch.empty(M, N, device=dev)
    res = {}
    for name, abl in (("full", 0), ("no-gload", 1), ("no-mfma", 2), ("no-store", 4), ("no-split", 8), ("no-gload-split", 9),
                      ("mfma-only", 13), ("nothing", 15)):
        relu = abl << 8
        def f(): _lib.check(L.tdm_gemm_f32(_lib.ptr(A), K, 1, _lib.ptr(B), 1, K, _lib.ptr(C), N, _lib.ptr(bias), None, M, N, K,
                                           relu, 1, 0, _lib.stream()))
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: " + "  ".join(f"{n}={v:.0f}us" for n, v in res.items()) + f"  | full = {fl/res['full']/1e6:.0f} TF")
for mode in (1, 2):
    _lib.check(L.tdm_set_gemm_mode(mode)); print("gemm mode", mode)
    run(32768, 2048, 256); run(32768, 256, 2048); run(32768, 768, 256); run(32768, 256, 256)
