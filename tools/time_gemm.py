"""Time the NT bf16 GEMM on the transformer's layer shapes (and the rounding head's logits shape):
    TDM_GEMM_WM=1|2 python tools/time_gemm.py      (0 / unset = the library's own choice)"""
import os, sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
def run(M, N, K, it=20):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) * 0.05
    bias = torch.zeros(N, device=dev); C = torch.empty(M, N, device=dev)
    def f(): _lib.check(L.tdm_gemm_f32(_lib.ptr(A), K, 1, _lib.ptr(B), 1, K, _lib.ptr(C), N, _lib.ptr(bias), None, M, N, K, 0, 1, 0, _lib.stream()))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    ref = (A[:64].double() @ B.double().T).float()
    err = ((C[:64] - ref).abs().max() / ref.abs().max()).item()
    return us, 2.0 * M * N * K / us / 1e6, err
print("TDM_GEMM_WM =", os.environ.get("TDM_GEMM_WM", "auto"))
for mode in (1, 2):
    _lib.check(L.tdm_set_gemm_mode(mode))
    for (M, N, K) in ((32768, 2048, 256), (32768, 256, 2048), (32768, 768, 256), (32768, 256, 256), (32768, 50257, 256)):
        us, tf, err = run(M, N, K, 5 if N > 4096 else 20)
        print(f"mode {mode} M={M} N={N} K={K}: {us:8.1f} us  {tf:6.1f} TFLOP/s algorithmic  (x3 MFMA: {3*tf/2500*100 if mode==1 else tf/2500*100:4.1f}% of bf16 peak)  rel err {err:.1e}")
