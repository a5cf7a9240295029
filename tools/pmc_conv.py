"""Run the bf16x3 conv kernel (rb4.conv1 shape) a few times for rocprofv3 --pmc collection."""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
hw, cin, cout, B, k = 28, 96, 32, 512, 3
abl = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn(B, hw, hw, cin, device=dev); w = torch.randn(k, k, cin, cout, device=dev) * 0.05
b = torch.zeros(cout, device=dev); y = torch.empty(B, hw, hw, cout, device=dev); sc = torch.empty(k*k*cin*cout, device=dev)
for _ in range(4):
    _lib.check(L.tdm_conv_nhwc_bf16x3_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, None, _lib.ptr(y), None, _lib.ptr(sc), B, hw, cin, cout, k, 1 | (abl << 8), _lib.stream()))
torch.cuda.synchronize()
