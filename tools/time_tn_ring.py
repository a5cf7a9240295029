"""Token-major (weight-gradient) bf16x3 GEMMs of the denoiser at config 5 (32,768 tokens): the 128 x 128-tile kernel vs the
256 x 256-tile LDS-DMA ring kernel (flag bit 4 of tdm_gemm_f32), with the ring kernel's ablations (no DMA / no MFMA / no stores)."""
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
def s16(x):
    o = torch.empty_like(x)
    _lib.check(L.tdm_split_s16_f32(_lib.ptr(x), _lib.ptr(o), x.numel(), _lib.stream()))
    return o
def run(T, N, K, sk_old, sk_new):
    g = torch.Generator(device=dev).manual_seed(N + K)
    dY = torch.randn(T, N, device=dev, generator=g); X = torch.randn(T, K, device=dev, generator=g)
    dY16, X16 = s16(dY), s16(X)
    dW = torch.empty(max(sk_old, sk_new), N, K, device=dev)
    def tn(fl, sk): return lambda: _lib.check(L.tdm_gemm_f32(_lib.ptr(dY16), 1, N, _lib.ptr(X16), K, 1, _lib.ptr(dW), K, None, None, N, K, T, fl, sk, N * K, _lib.stream()))
    tn(2, sk_old)(); ref = dW[:sk_old].sum(0)
    want = dY.double().T @ X.double()
    e_old = ((ref.double() - want).norm() / want.norm()).item()
    dW.zero_(); tn(2 | 16, sk_new)(); got = dW[:sk_new].sum(0)
    e_new = ((got.double() - want).norm() / want.norm()).item()
    r = [t(tn(2, sk_old)), t(tn(2 | 16, sk_new))] + [t(tn(2 | 16 | (a << 8), sk_new)) for a in (1, 2, 4, 3, 7, 5, 6, 8, 16, 20)]
    print(f"dW[{N}][{K}] over {T} tokens: 128-tile (sk {sk_old}) {r[0]:.1f} us err {e_old:.1e} | ring (sk {sk_new}) {r[1]:.1f} us err {e_new:.1e} | "
          f"no-DMA {r[2]:.1f}  no-MFMA {r[3]:.1f}  no-store {r[4]:.1f}  no-DMA+MFMA {r[5]:.1f}  nothing {r[6]:.1f}  no-DMA+store {r[7]:.1f}  no-MFMA+store {r[8]:.1f}  stores-to-slab-0 {r[9]:.1f}  L2-resident loads {r[10]:.1f}  L2-resident loads, no store {r[11]:.1f}", flush=True)
_lib.check(L.tdm_set_gemm_mode(1))
T = 32768
run(T, 2048, 256, 16, 32); run(T, 256, 2048, 16, 32)
