"""Role ablation of the warp-specialised N = 32 conv kernel (diagnostic; outputs are wrong when a role is skipped):
ablate bits 64 = loaders issue no global loads after the prologue, 512 = no LDS staging, 128 = walkers skip the walks,
256 = consumers skip the MFMAs."""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
def run(hw, cin, cout, B, outs="s16"):
    k = 3
    x = torch.randn(B, hw, hw, cin, device=dev); w = torch.randn(k, k, cin, cout, device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    y = torch.empty(B, hw, hw, cout, device=dev); y16 = torch.empty_like(y); aux = torch.empty_like(y); res = torch.randn_like(y)
    sc = torch.empty(B * hw * hw * (cin + cout) + 65 * k * k * cin * cout + 128, device=dev)
    xs = torch.empty_like(x)
    _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, None, _lib.ptr(y), None, _lib.ptr(xs), None,
                                       _lib.ptr(sc), B, hw, cin, cout, k, 1, _lib.stream()))
    off = (k * k * cin * cout + 63) & ~63
    x16 = sc[off:off + x.numel()]
    full = outs == "all"
    out = {}
    for ws in (0, 1):
        _lib.check(L.tdm_set_conv_ws(ws))
        for name, abl in ((("one-role", 0),) if ws == 0 else (("ws full", 0), ("no preload", 2048), ("no preload/only walk", 2048 | 64 | 512 | 256), ("no loads", 64), ("no loads/stage", 64 | 512), ("no walk", 128), ("no mfma", 256),
                                                                ("only mfma", 64 | 512 | 128), ("only walk", 64 | 512 | 256), ("only load+stage", 128 | 256), ("nothing", 64 | 512 | 128 | 256))):
            fl = 1 | 4 | 8 | (abl << 8)
            def f(): _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x16), _lib.ptr(w), _lib.ptr(b), _lib.ptr(res) if full else None, None,
                                                        _lib.ptr(y) if full else None, _lib.ptr(aux) if full else None, _lib.ptr(y16), None,
                                                        _lib.ptr(sc), B, hw, cin, cout, k, fl, _lib.stream()))
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); e1.synchronize()
            out[name] = e0.elapsed_time(e1) / 20 * 1e3
    print(f"hw={hw} {cin}->{cout} B={B} outs={outs}: " + "  ".join(f"{n}={v:.0f}us" for n, v in out.items()))
run(28, 32, 32, 512); run(28, 32, 32, 512, outs="all"); run(28, 32, 32, 4096)
