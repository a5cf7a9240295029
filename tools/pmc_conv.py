"""Run the S16 bf16x3 conv kernel alone (rb4.conv1 shape, B=512: 96->32 3x3 @28x28) for rocprofv3
--kernel-trace / --pmc collection: input pre-split and weights pre-packed once, then N timed launches."""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
hw, cin, cout, B, k = 28, 96, 32, 512, 3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
x = torch.randn(B, hw, hw, cin, device=dev); w = torch.randn(k, k, cin, cout, device=dev) * 0.05
b = torch.zeros(cout, device=dev); y = torch.empty(B, hw, hw, cout, device=dev)
woff = (k * k * cin * cout + 63) & ~63
sc = torch.empty(woff + B * hw * hw * cin + 128, device=dev)
def run(inp, flags):
    _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(inp), _lib.ptr(w), _lib.ptr(b), None, None, _lib.ptr(y), None, None, None,
                                       _lib.ptr(sc), B, hw, cin, cout, k, flags, _lib.stream()))
run(x, 1)                       # packs weights and pre-splits the input into the scratch
xs = sc[woff:woff + B * hw * hw * cin]
for _ in range(n):
    run(xs, 1 | 4 | 8)          # the conv kernel alone
torch.cuda.synchronize()
