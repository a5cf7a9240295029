"""Ablation timing of the LDS-DMA ring GEMM (gemm_ring.hip) on the FFN shapes: full / no stores / no epilogue / no MFMA / no DMA."""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def t(f, it=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
def run(M, N, K):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; bias = torch.zeros(N, device=dev)
    A16, W16 = torch.empty_like(A), torch.empty_like(W); C = torch.empty(M, N, device=dev)
    _lib.check(L.tdm_split_s16_f32(_lib.ptr(A), _lib.ptr(A16), A.numel(), _lib.stream()))
    _lib.check(L.tdm_split_s16_f32(_lib.ptr(W), _lib.ptr(W16), W.numel(), _lib.stream()))
    def nt(fl): return lambda: _lib.check(L.tdm_gemm_f32(_lib.ptr(A16), K, 1, _lib.ptr(W16), 1, K, _lib.ptr(C), N, _lib.ptr(bias), None, M, N, K, fl, 1, 0, _lib.stream()))
    base = 1 | 2 | 4      # relu, S16 in, S16 out
    names = [("full", 0), ("no stores", 4), ("no epilogue", 8), ("no MFMA", 2), ("no DMA", 1), ("no MFMA, no epilogue", 10), ("no DMA, no epilogue", 9), ("only loop+barriers", 11)]
    print(f"M={M} N={N} K={K}: " + "  ".join(f"{n} {t(nt(base | (a << 8))):.0f}us" for n, a in names))
_lib.check(L.tdm_set_gemm_mode(int(sys.argv[1]) if len(sys.argv) > 1 else 1))
run(32768, 2048, 256); run(32768, 256, 2048); run(32768, 768, 256)
