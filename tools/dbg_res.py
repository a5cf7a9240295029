import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
B, hw, cin, cout, k = 3, 28, 32, 32, 3
g = torch.Generator().manual_seed(0)
x = torch.randn(B, hw, hw, cin, generator=g).to(dev); w = (torch.randn(k, k, cin, cout, generator=g) * 0.05).to(dev)
b = torch.randn(cout, generator=g).to(dev); res = torch.randn(B, hw, hw, cout, generator=g).to(dev)
out = torch.empty(B, hw, hw, cout, device=dev); aux = torch.empty_like(out)
sc = torch.empty(B * hw * hw * (cin + cout) + 65 * k * k * cin * cout + 128, device=dev)
_lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(res), None, _lib.ptr(out), _lib.ptr(aux), None, None, _lib.ptr(sc), B, hw, cin, cout, k, 1, _lib.stream()))
torch.cuda.synchronize()
d = (out - aux)            # should equal res
print("max |out-aux-res|", (d - res).abs().max().item(), " max|out-aux|", d.abs().max().item())
bad = ((d - res).abs() > 1e-4)
print("bad count", bad.sum().item(), "of", bad.numel())
idx = bad.nonzero()[:10]; print(idx.tolist())
print("per-pixel bad (first image, first rows):", bad[0].any(-1).int()[:4].tolist())
