"""Per-op attention timing in the attention modes (config 5: B = 256, L = 128, D = 256, 4 heads).
   python tools/time_attn.py [--ablate 1,2,3,4,6,7]   (ablation bits need a -DTDM_DIAG build: 1 no chunk compute, 2 no staging, 4 no stores)"""
import argparse, sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
ap = argparse.ArgumentParser(); ap.add_argument("--ablate", default=""); ap.add_argument("--modes", default="2,1")
a = ap.parse_args()
L_ = _lib.lib(); dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def run(B, L, D, H, p_drop):
    qkv = torch.randn(B, L, 3 * D, device=dev); dO = torch.randn(B, L, D, device=dev)
    o = torch.empty(B, L, D, device=dev); lse = torch.empty(B * H, L, device=dev)
    dqkv = torch.empty_like(qkv); Dv = torch.empty(B * H, L, device=dev)
    out = []
    for mode in [int(m) for m in a.modes.split(",")]:
        _lib.check(L_.tdm_set_attn_mode(mode))
        def f(): _lib.check(L_.tdm_attention_fwd_f32(_lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), B, L, D, H, p_drop, 7, 1, _lib.stream()))
        def b(): _lib.check(L_.tdm_attention_bwd_f32(_lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), _lib.ptr(dO), _lib.ptr(dqkv), _lib.ptr(Dv), B, L, D, H, p_drop, 7, 1, _lib.stream()))
        o16 = torch.empty_like(o); dq16 = torch.empty_like(qkv)
        def sf(which, out, out16, aux):
            return lambda: _lib.check(L_.tdm_attention_step_form_f32(which, _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), _lib.ptr(dO), _lib.ptr(out) if out is not None else None,
                                                                     _lib.ptr(out16), _lib.ptr(aux), B, L, D, H, p_drop, 7, 1, _lib.stream()))
        forms = ((f, "fwd"), (b, "bwd"))
        if mode == 2:   # the S16-only backward outputs exist in the bf16 kernels only
            forms += ((sf(0, o, o16, lse), "step.fwd"), (sf(1, None, dq16, Dv), "step.dq"), (sf(2, None, dq16, Dv), "step.dkv"))
        for fn, nm in forms:
            s = f"mode{mode}.{nm}={timeit(fn):.0f}us"
            for ab in [int(x) for x in a.ablate.split(",") if x]:
                L_.tdm_attn_set_ablate(ab)
                s += f" [{ab}]{timeit(fn):.0f}"
                L_.tdm_attn_set_ablate(0)
            out.append(s)
    _lib.check(L_.tdm_set_attn_mode(2))
    print(f"B={B} L={L} D={D} H={H} p_drop={p_drop}: " + "  ".join(out), flush=True)
run(256, 128, 256, 4, 0.1); run(256, 128, 256, 4, 0.0); run(64, 512, 256, 4, 0.0)
