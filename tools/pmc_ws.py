"""One-role vs warp-specialised N = 32 conv under a PMC pass (instruction counts per launch):
   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -- python tools/pmc_ws.py"""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
B, hw, cin, cout, k = 512, 28, 32, 32, 3
x = torch.randn(B, hw, hw, cin, device=dev); w = torch.randn(k, k, cin, cout, device=dev) * 0.05
b = torch.zeros(cout, device=dev); y = torch.empty(B, hw, hw, cout, device=dev); y16 = torch.empty_like(y)
sc = torch.empty(B * hw * hw * (cin + cout) + 65 * k * k * cin * cout + 128, device=dev); xs = torch.empty_like(x)
_lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, None, _lib.ptr(y), None, _lib.ptr(xs), None, _lib.ptr(sc), B, hw, cin, cout, k, 1, _lib.stream()))
off = (k * k * cin * cout + 63) & ~63; x16 = sc[off:off + x.numel()]
for ws in (0, 1):
    _lib.check(L.tdm_set_conv_ws(ws))
    for _ in range(6):
        _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(x16), _lib.ptr(w), _lib.ptr(b), None, None, None, None, _lib.ptr(y16), None, _lib.ptr(sc), B, hw, cin, cout, k, 1 | 4 | 8, _lib.stream()))
    torch.cuda.synchronize()
_lib.check(L.tdm_set_conv_ws(0))
