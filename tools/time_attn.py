"""Per-op attention timing in the three attention modes (config 5: B = 256, L = 128, D = 256, 4 heads)."""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L_ = _lib.lib(); dev = torch.device("cuda:0")
def run(B, L, D, H, p_drop):
    qkv = torch.randn(B, L, 3 * D, device=dev); dO = torch.randn(B, L, D, device=dev)
    o = torch.empty(B, L, D, device=dev); lse = torch.empty(B * H, L, device=dev)
    dqkv = torch.empty_like(qkv); Dv = torch.empty(B * H, L, device=dev)
    out = []
    for mode in (2, 1):
        _lib.check(L_.tdm_set_attn_mode(mode))
        def f(): _lib.check(L_.tdm_attention_fwd_f32(_lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), B, L, D, H, p_drop, 7, 1, _lib.stream()))
        def b(): _lib.check(L_.tdm_attention_bwd_f32(_lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), _lib.ptr(dO), _lib.ptr(dqkv), _lib.ptr(Dv), B, L, D, H, p_drop, 7, 1, _lib.stream()))
        for fn, nm in ((f, "fwd"), (b, "bwd")):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); e1.synchronize()
            out.append(f"mode{mode}.{nm}={e0.elapsed_time(e1) / 20 * 1e3:.0f}us")
    _lib.check(L_.tdm_set_attn_mode(2))
    print(f"B={B} L={L} D={D} H={H} p_drop={p_drop}: " + "  ".join(out))
run(256, 128, 256, 4, 0.1); run(256, 128, 256, 4, 0.0); run(64, 512, 256, 4, 0.0)
