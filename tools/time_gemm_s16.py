"""NT / TN bf16x3 GEMM on the transformer's layer shapes: fp32 operands (split in the loader) vs S16 operands, fp32 vs S16 output."""
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
    def nt(a, w, fl): return lambda: _lib.check(L.tdm_gemm_f32(_lib.ptr(a), K, 1, _lib.ptr(w), 1, K, _lib.ptr(C), N, _lib.ptr(bias), None, M, N, K, fl, 1, 0, _lib.stream()))
    r = [t(nt(A, W, 0)), t(nt(A16, W16, 2)), t(nt(A16, W16, 6)), t(nt(A16, W16, 2 | (4 << 8)))]
    dY = torch.randn(M, N, device=dev); dY16 = torch.empty_like(dY); dW = torch.empty(16, N, K, device=dev)
    _lib.check(L.tdm_split_s16_f32(_lib.ptr(dY), _lib.ptr(dY16), dY.numel(), _lib.stream()))
    sk = max(1, 512 // (((N + 127) // 128) * ((K + 127) // 128))); sk = min(sk, 16)
    def tn(dy, x, fl): return lambda: _lib.check(L.tdm_gemm_f32(_lib.ptr(dy), 1, N, _lib.ptr(x), K, 1, _lib.ptr(dW), K, None, None, N, K, M, fl, sk, N * K, _lib.stream()))
    r += [t(tn(dY, A, 0)), t(tn(dY16, A16, 2))]
    print(f"M={M} N={N} K={K}: NT fp32-in {r[0]:.0f}us  S16-in {r[1]:.0f}us  S16-in+S16-out {r[2]:.0f}us  S16-in no-store {r[3]:.0f}us | TN(splitk {sk}) fp32-in {r[4]:.0f}us  S16-in {r[5]:.0f}us")
_lib.check(L.tdm_set_gemm_mode(1))
run(32768, 2048, 256); run(32768, 256, 2048); run(32768, 768, 256); run(32768, 256, 256); run(32768, 256, 768)
