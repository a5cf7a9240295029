"""Timing ablation of the bf16x3 conv kernel (diagnostic; outputs are wrong when a stage is skipped)."""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
def run(hw, cin, cout, B, k=3):
    x = torch.randn(B, hw, hw, cin, device=dev); w = torch.randn(k, k, cin, cout, device=dev) * 0.05
    b = torch.zeros(cout, device=dev); y = torch.empty(B, hw, hw, cout, device=dev); sc = torch.empty(k*k*cin*cout, device=dev)
    res = {}
    for name, abl in (("full", 0), ("no-input-loads", 1), ("no-weight-stage", 2), ("no-mfma", 4), ("nothing", 7)):
        fl = 1 | (abl << 8)
        def f(): _lib.check(L.tdm_conv_nhwc_bf16x3_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, None, _lib.ptr(y), None, _lib.ptr(sc), B, hw, cin, cout, k, fl, _lib.stream()))
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2*k*k*cin*cout*hw*hw*B
    print(f"hw={hw} {cin}->{cout} k={k} B={B}: " + "  ".join(f"{n}={v:.0f}us" for n, v in res.items()) + f"  | full = {fl/res['full']/1e6:.0f} TF-equiv")
run(28, 96, 32, 512); run(28, 32, 32, 512); run(14, 64, 64, 512); run(14, 32, 64, 512); run(28, 32, 32, 4096)
