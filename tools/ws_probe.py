"""Role timeline of the warp-specialised N = 32 conv kernel (diagnostic, ConvArgs::ablate & 1024): lane 0 of the first
consumer / loader / walker wave stamps the shader clock: tag 1 = arriving at a barrier, 2 = leaving it, 3 = loader: LDS
staging done / walker: inputs arrived, 4 = walker: walk done.  Prints the median time between consecutive stamps."""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def run(hw, cin, cout, B, outs="s16", abl=0):
    k = 3
    x = torch.randn(B, hw, hw, cin, device=dev); w = torch.randn(k, k, cin, cout, device=dev) * 0.05
    b = torch.zeros(cout, device=dev); y = torch.empty(B, hw, hw, cout, device=dev); y16 = torch.empty_like(y); res = torch.randn_like(y)
    sc = torch.empty(B * hw * hw * (cin + cout) + 65 * k * k * cin * cout + 128, device=dev); xs = torch.empty_like(x)
    _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, None, _lib.ptr(y), None, _lib.ptr(xs), None, _lib.ptr(sc), B, hw, cin, cout, k, 1, _lib.stream()))
    off = (k * k * cin * cout + 63) & ~63; x16 = sc[off:off + x.numel()]
    grid = min(256, (B * hw * hw + 255) // 256)
    aux = torch.zeros(y.numel() + grid * 3 * 64 * 2 + 64, device=dev)
    full = outs == "all"
    _lib.check(L.tdm_set_conv_ws(1))
    fl = 1 | 4 | 8 | ((1024 | abl) << 8)
    for _ in range(3):
        _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x16), _lib.ptr(w), _lib.ptr(b), _lib.ptr(res) if full else None, None, _lib.ptr(y) if full else None,
                                           _lib.ptr(aux), _lib.ptr(y16), None, _lib.ptr(sc), B, hw, cin, cout, k, fl, _lib.stream()))
    torch.cuda.synchronize()
    raw = aux[y.numel():y.numel() + grid * 3 * 64 * 2].view(torch.int64).view(grid, 3, 64).cpu()
    tag = (raw >> 56)[0]; d = (raw & ((1 << 56) - 1)).double()
    print(f"hw={hw} {cin}->{cout} B={B} outs={outs} ablate={abl}: tag:median ticks since the previous stamp, stamps 24.. (steady state), {grid} workgroups")
    for r, name in enumerate(("consumer", "helper")):
        n = int((raw[0, r] != 0).sum().item())
        dt = d[:, r, 1:] - d[:, r, :-1]
        print(f"  {name:9s} " + " ".join(f"{int(tag[r, i])}:{dt[:, i - 1].median():.0f}" for i in range(24, min(n, 50))))
    _lib.check(L.tdm_set_conv_ws(0))
run(28, 32, 32, 2048); run(28, 32, 32, 2048, abl=128)
