"""Time the S16 weight-gradient kernel alone.  usage: time_wgrad.py [path/to/libtdm_hip.so]"""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
L = _lib.lib()
dev = torch.device("cuda:0")
def run(hw, cin, cout, B, k=3):
    x = torch.randn(B, hw, hw, cin, device=dev); g = torch.randn(B, hw, hw, cout, device=dev)
    dw = torch.empty(k, k, cin, cout, device=dev)
    sc = torch.empty(B * hw * hw * (cin + cout) + 65 * k * k * cin * cout + 128, device=dev)
    def f(): _lib.check(L.tdm_conv_wgrad_nhwc_s16_f32(_lib.ptr(x), None, _lib.ptr(g), _lib.ptr(dw), _lib.ptr(sc), B, hw, cin, cout, k,
                                                       _lib.stream()))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"wgrad hw={hw} {cin}->{cout} k={k} B={B}: {us:.0f} us per call (incl. 2 to_s16 conversions + slab reduce)")
run(28, 32, 32, 512); run(28, 96, 32, 512); run(14, 64, 64, 512)
