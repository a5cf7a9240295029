#!/usr/bin/env python3
"""Which launch of the text step changes its bits next to a side stream of token-major GEMMs?  Forward (saved tensors compared
region by region with the quiet run), then the backward from a quiet forward's workspace (gradients per parameter tensor).
    python tools/contention_bisect.py [--B 256] [--depth 3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinydiffusionmodels_amd import _lib   # noqa: E402
from tinydiffusionmodels_amd import transformer_engine as TE   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--n-side", type=int, default=30)
    ap.add_argument("--p", type=float, default=0.1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    _lib.check(L.tdm_set_gemm_mode(1))
    B, Lq, D, H, depth, F = args.B, 128, 256, 4, args.depth, 2048
    M = B * Lq
    Ms = 32768
    dy = torch.randn(Ms, 2048, device=dev) * 0.01
    xx = torch.randn(Ms, 256, device=dev)
    dy16, x16 = torch.empty_like(dy), torch.empty_like(xx)
    _lib.check(L.tdm_split_s16_f32(_lib.ptr(dy), _lib.ptr(dy16), dy.numel(), _lib.stream()))
    _lib.check(L.tdm_split_s16_f32(_lib.ptr(xx), _lib.ptr(x16), xx.numel(), _lib.stream()))
    slab = torch.empty(8, 2048, 256, device=dev)
    side = torch.cuda.Stream()

    def with_side(fn, n):
        side.wait_stream(torch.cuda.current_stream())
        if n:
            with torch.cuda.stream(side):
                for _ in range(n):
                    _lib.check(L.tdm_gemm_f32(_lib.ptr(dy16), 1, 2048, _lib.ptr(x16), 256, 1, _lib.ptr(slab), 256, None, None, 2048, 256, Ms, 2, 8,
                                              2048 * 256, side.cuda_stream), "tn gemm")
        fn()
        torch.cuda.synchronize()

    # mirror of tt_carve (transformer.hip): names of the saved tensors
    regions, off = [], [0]

    def take(name, n):
        regions.append((name, off[0], n)); off[0] += (n + 63) & ~63

    def mask_elems(M, F): return ((M + 15) // 16) * ((F // 32 + 3) // 4) * 64
    take("that", B); take("tb", B * D); take("abuf", M * D); take("wT", max(F, 3 * D) * D); take("wT2", F * D); take("wT_all", depth * (4 * D * D + 2 * F * D))
    for l in range(depth):
        for nm, n in (("hin", M * D), ("qkv", M * 3 * D), ("o", M * D), ("lse", B * H * Lq), ("s1", M * D), ("mean1", M), ("rstd1", M), ("h1", M * D), ("f1", M * F),
                      ("s2", M * D), ("mean2", M), ("rstd2", M), ("hin16", M * D), ("o16", M * D), ("h1_16", M * D), ("fmask", mask_elems(M, F))):
            take(f"L{l}.{nm}", n)
    cfg = TE.TTConfig(D, H, depth, F)
    torch.manual_seed(0)
    flat = torch.randn(cfg.nparam, device=dev) * 0.05
    x = torch.randn(B, Lq, D, device=dev) * 0.5
    t = torch.randint(0, 1000, (B,), device=dev)
    ws = TE.TTWorkspace(cfg, B, Lq, dev, training=True)
    out = torch.empty_like(x)
    with_side(lambda: TE.tt_forward(cfg, flat, x, t, ws, True, out=out, p_drop=args.p, seed=7), 0)
    ws_ref, out_ref = ws.ws.clone(), out.clone()
    bits = lambda a: a.view(torch.int32)
    for rep in range(2):
        ws.ws.zero_()
        with_side(lambda: TE.tt_forward(cfg, flat, x, t, ws, True, out=out, p_drop=args.p, seed=7), args.n_side)
        print(f"forward rep {rep}: output equal {torch.equal(bits(out), bits(out_ref))}")
        for name, o, n in regions:
            if name in ("abuf", "wT", "wT2", "wT_all"):
                continue
            d = bits(ws.ws[o:o + n]) != bits(ws_ref[o:o + n])
            if d.any():
                print(f"   {name}: {int(d.sum())} of {n} words differ")
    # backward from the quiet forward's workspace
    dout = torch.randn(B, Lq, D, device=dev) * 1e-3
    grads = torch.empty(cfg.nparam, device=dev)
    ws.ws.copy_(ws_ref)
    with_side(lambda: TE.tt_backward(cfg, flat, dout, ws, grads=grads, p_drop=args.p, seed=7), 0)
    g_ref = grads.clone()
    offs = TE.param_offsets(D, depth, F)
    names = ["in_w", "in_b", "out_w", "out_b", "l1_w", "l1_b", "l2_w", "l2_b", "n1_w", "n1_b", "n2_w", "n2_b"]
    for rep in range(3):
        ws.ws.copy_(ws_ref)
        with_side(lambda: TE.tt_backward(cfg, flat, dout, ws, grads=grads, p_drop=args.p, seed=7), args.n_side if rep else 0)
        bad = []
        for i in range(len(offs) - 1):
            if not torch.equal(bits(grads[offs[i]:offs[i + 1]]), bits(g_ref[offs[i]:offs[i + 1]])):
                bad.append(f"L{i // 12}.{names[i % 12]}" if i < 12 * depth else f"te{i - 12 * depth}")
        print(f"backward rep {rep} ({'quiet' if rep == 0 else 'next to TN GEMMs'}): differing tensors: {bad if bad else 'none'}")


if __name__ == "__main__":
    main()
