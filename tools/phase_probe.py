"""Per-workgroup phase timeline of conv_s16_kernel<28,1> (diagnostic, ConvArgs::ablate & 16): shader-clock stamps at
entry / plan+first loads issued / chunk-0 staged / chunk-0 multiplied / chunk-1 staged / chunk-1 multiplied / stores done."""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def run(hw, cin, cout, B, k=3, solo=False):
    x = torch.randn(B, hw, hw, cin, device=dev); w = torch.randn(k, k, cin, cout, device=dev) * 0.05
    b = torch.zeros(cout, device=dev); y = torch.empty(B, hw, hw, cout, device=dev); y16 = torch.empty_like(y)
    sc = torch.empty(B * hw * hw * (cin + cout) + 65 * k * k * cin * cout + 128, device=dev); xs = torch.empty_like(x)
    _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, None, _lib.ptr(y), None, _lib.ptr(xs), None, _lib.ptr(sc), B, hw, cin, cout, k, 1, _lib.stream()))
    off = (k * k * cin * cout + 63) & ~63; x16 = sc[off:off + x.numel()]
    ntiles = (B * hw * hw + 255) // 256
    aux = torch.zeros(y.numel() + ntiles * 32 + 64, device=dev)   # saved-copy output + the stamp table behind it
    fl = 1 | 4 | 8 | ((16 | (32 if solo else 0)) << 8)
    for _ in range(3):
        _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x16), _lib.ptr(w), _lib.ptr(b), None, None, None, _lib.ptr(aux), _lib.ptr(y16), None, _lib.ptr(sc), B, hw, cin, cout, k, fl, _lib.stream()))
    torch.cuda.synchronize()
    d = aux[y.numel():y.numel() + ntiles * 32].view(torch.int64).view(ntiles, 16).cpu().double()
    t0 = d[:, 0].min()
    nch = cin // 16
    names = ["prologue"] + [x for c in range(nch) for x in ((f"wait+stage{c}" if c == 0 else f"stage{c}"), f"compute{c}")] + ["epi: request inputs", "epi: barrier", "epi: to LDS", "epi: walk 0", "epi: walk 1", "store drain"]
    names = names[:15]
    print(d[:3].tolist())
    d = d[d[:, 0] > 0]
    ph = d[:, 1:len(names) + 1] - d[:, 0:len(names)]
    print(f"hw={hw} {cin}->{cout} B={B} tiles={ntiles} {'ONE WORKGROUP PER CU' if solo else ''}; clock ticks (same unit as the stamps); kernel span {float(d[:, 1:16].max() - t0):.0f}")
    print("  start of WG (min/median/max since first):", float((d[:, 0] - t0).min()), float((d[:, 0] - t0).median()), float((d[:, 0] - t0).max()))
    for i, n in enumerate(names):
        print(f"  {n:20s} median {float(ph[:, i].median()):8.0f}  p10 {float(ph[:, i].quantile(0.1)):8.0f}  p90 {float(ph[:, i].quantile(0.9)):8.0f}")
    life = d[:, len(names)] - d[:, 0]
    print(f"  lifetime     median {float(life.median()):8.0f}  p10 {float(life.quantile(0.1)):8.0f}  p90 {float(life.quantile(0.9)):8.0f}")
run(14, 64, 64, 512); run(14, 64, 64, 512, solo=True)
