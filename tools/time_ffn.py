#!/usr/bin/env python3
"""Fused FFN chain (csrc/ffn_chain.hip): results against an fp64 torch reference, and launch times.
   python tools/time_ffn.py [--M 32768] [--F 2048] [--p 0.1] [--iters 30] [--check-rows 2048]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinydiffusionmodels_amd import _lib   # noqa: E402


def s16(t):
    out = torch.empty_like(t)
    _lib.check(_lib.lib().tdm_split_s16_f32(_lib.ptr(t), _lib.ptr(out), t.numel(), _lib.stream()), "split")
    return out


def from_s16(t16, rows, cols):
    """hi + lo of an S16 tensor -> fp32 (host)."""
    raw = t16.cpu().numpy().view(np.uint16).reshape(rows, cols // 16, 2, 16).astype(np.uint32)
    hi = (raw[:, :, 0, :] << 16).view(np.float32)
    lo = (raw[:, :, 1, :] << 16).view(np.float32)
    return torch.from_numpy((hi + lo).reshape(rows, cols))


def keep_rows(p, seed, site, rows, width):
    """keep mask [len(rows)][width] of dropout site `site` over a (M, width) tensor (host evaluation of the library's hash)."""
    out = np.ones((len(rows), width), dtype=bool)
    if p > 0:
        k = np.empty(width, dtype=np.uint8)
        for i, r in enumerate(rows):
            _lib.check(_lib.lib().tdm_dropout_keep_u8(p, seed, site, int(r) * width, width, k.ctypes.data), "keep")
            out[i] = k.astype(bool)
    return torch.from_numpy(out)


def mask_to_gate(mask, M, F, rows):
    """sign-mask words [tblk][word][lane] (uint32) -> bool [len(rows)][F]: byte b of word w of lane l = hidden block 4 w + b, bit e
    of the byte = hidden unit 32 block + 8 (l >> 4) + e of token 16 tblk + (l & 15)"""
    nfb = F // 32
    mk = mask.cpu().numpy().view(np.uint32).reshape(-1, (nfb + 3) // 4, 64)
    gate = np.zeros((len(rows), F), dtype=bool)
    e = np.arange(8)
    for i, tok in enumerate(rows):
        tb, c = int(tok) // 16, int(tok) % 16
        for gg in range(4):
            w = mk[tb, :, c + 16 * gg]                                               # [words]
            by = ((w[:, None] >> (8 * np.arange(4)[None, :])) & 0xff).reshape(-1)[:nfb]   # [nfb]
            bits = (by[:, None] >> e[None, :]) & 1                                   # [nfb][8]
            f = 32 * np.arange(nfb)[:, None] + 8 * gg + e[None, :]
            gate[i, f.reshape(-1)] = bits.reshape(-1).astype(bool)
    return torch.from_numpy(gate)


def chain(mode, nprod, x16, wa16, ba, wb16, bb, y, mid16, mask, gs, p, seed, M, D, F):
    L = _lib.lib()
    _lib.check(L.tdm_ffn_chain_f32(mode, nprod, _lib.ptr(x16), _lib.ptr(wa16), _lib.ptr(ba), _lib.ptr(wb16), _lib.ptr(bb), _lib.ptr(y),
                                   _lib.ptr(mid16), _lib.ptr(mask), gs, p, seed, 3, 4, M, D, F, _lib.stream()), "ffn_chain")


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=32768)
    ap.add_argument("--F", type=int, default=2048)
    ap.add_argument("--p", type=float, default=0.1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--check-rows", type=int, default=1024)
    ap.add_argument("--ablate", type=str, default="", help="comma list of ablation bit sets to time (diagnostics), e.g. 1,2,4,6,8,16,32")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, D, F, p, seed = a.M, 256, a.F, a.p, 0x1234567
    L = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(M, D, device=dev, generator=g)
    W1 = torch.randn(F, D, device=dev, generator=g) * (1 / D ** 0.5)
    b1 = torch.randn(F, device=dev, generator=g) * 0.1
    W2 = torch.randn(D, F, device=dev, generator=g) * (1 / F ** 0.5)
    b2 = torch.randn(D, device=dev, generator=g) * 0.1
    gy = torch.randn(M, D, device=dev, generator=g)
    x16, w1_16, w2_16, gy16 = s16(x), s16(W1), s16(W2), s16(gy)
    w2t16, w1t16 = s16(W2.t().contiguous()), s16(W1.t().contiguous())
    y, dx = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev)
    h16, dz16 = torch.empty(M, F, device=dev), torch.empty(M, F, device=dev)
    mask = torch.zeros(L.tdm_ffn_chain_mask_count(M, F), dtype=torch.int32, device=dev)
    gs = 1.0 / (1 - p) if p > 0 else 1.0
    R = min(a.check_rows, M)
    rows = sorted(set(range(R // 2)) | set(range(M - R // 2, M)))      # head and tail of the token range
    ridx = torch.tensor(rows, dtype=torch.long, device=dev)
    for nprod in (3, 1):
        if R > 0:
            chain(1, nprod, x16, w1_16, b1, w2_16, b2, y, h16, mask, 1.0, p, seed, M, D, F)
            chain(2, nprod, gy16, w2t16, None, w1t16, None, dx, dz16, mask, gs, 0.0, 0, M, D, F)
            torch.cuda.synchronize()
            xs, W1d, W2d = x[ridx].double().cpu(), W1.double().cpu(), W2.double().cpu()
            hid = torch.relu(xs @ W1d.T + b1.double().cpu())
            hid = hid * keep_rows(p, seed, 3, rows, F) / (1 - p) if p > 0 else hid
            yref = hid @ W2d.T + b2.double().cpu()
            yref = yref * keep_rows(p, seed, 4, rows, D) / (1 - p) if p > 0 else yref
            err_y = ((y[ridx].cpu().double() - yref).abs().max() / yref.abs().max()).item()
            hgot = from_s16(h16[ridx], len(rows), F).double()
            err_h = ((hgot - hid).abs().max() / hid.abs().max()).item()
            gate = mask_to_gate(mask, M, F, rows)
            flips = int((gate != (hid > 0)).sum())            # sign flips of near-zero pre-activations are legitimate
            stored = int((gate != (hgot > 0)).sum())          # ... but the mask must agree with the hidden tensor that was stored
            dz_ref = (gy[ridx].double().cpu() @ W2d) * gate * gs
            dx_ref = dz_ref @ W1d
            e_dz = ((from_s16(dz16[ridx], len(rows), F).double() - dz_ref).abs().max() / dz_ref.abs().max()).item()
            e_dx = ((dx[ridx].cpu().double() - dx_ref).abs().max() / dx_ref.abs().max()).item()
            print(f"nprod {nprod}: y {err_y:.2e}, hidden {err_h:.2e}, mask vs stored hidden {stored} mismatches (vs fp64 signs {flips}), "
                  f"dZ {e_dz:.2e}, dX {e_dx:.2e}   [max-abs error / max-abs reference, {len(rows)} rows]")
        for mode, nm in ((0, "fwd (nothing saved)"), (1, "fwd + hidden S16 + masks"), (2, "data gradient")):
            if mode == 2:
                fn = lambda: chain(2, nprod, gy16, w2t16, None, w1t16, None, dx, dz16, mask, gs, 0.0, 0, M, D, F)   # noqa: E731
            else:
                fn = lambda mode=mode: chain(mode, nprod, x16, w1_16, b1, w2_16, b2, y, h16, mask, 1.0, p if mode else 0.0, seed, M, D, F)   # noqa: E731
            us = timeit(fn, a.iters)
            fl = 2 * 2.0 * M * D * F * (3 if nprod == 3 else 1)
            extra = ""
            for ab in [int(x) for x in a.ablate.split(",") if x]:
                L.tdm_ffn_chain_set_ablate(ab)
                extra += f"  [{ab}] {timeit(fn, a.iters):.0f}"
                L.tdm_ffn_chain_set_ablate(0)
            print(f"  nprod {nprod} mode {mode} {nm:28s}: {us:8.1f} us   {fl / us / 1e6:7.1f} TFLOP/s of bf16 issue{extra}")


if __name__ == "__main__":
    main()
