"""Stand-alone timing of conv_s16 launches of a given shape (fp32 output only): python tools/time_conv_shapes.py"""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def run(hw, cin, cout, B, k=3):
    x = torch.randn(B, hw, hw, cin, device=dev); w = torch.randn(k, k, cin, cout, device=dev) * 0.05
    b = torch.zeros(cout, device=dev); y = [torch.empty(B, hw, hw, cout, device=dev) for _ in range(2)]
    sc = torch.empty(B * hw * hw * (cin + cout) + 65 * k * k * cin * cout + 128, device=dev)
    _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, None, _lib.ptr(y[0]), None, None, None, _lib.ptr(sc), B, hw, cin, cout, k, 1, _lib.stream()))   # (out_s16 = None: it would have to be a (B, hw, hw, cout) buffer)
    off = (k * k * cin * cout + 63) & ~63; x16 = sc[off:off + x.numel()]
    def f(i): _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x16), _lib.ptr(w), _lib.ptr(b), None, None, _lib.ptr(y[i & 1]), None, None, None, _lib.ptr(sc), B, hw, cin, cout, k, 4 | 8, _lib.stream()))
    for i in range(3): f(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): f(i)
    e1.record(); e1.synchronize()
    print(f"hw={hw} {cin}->{cout} k={k} B={B}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
run(28, 32, 96, 512); run(28, 64, 96, 512); run(28, 32, 32, 512); run(28, 96, 32, 512)
