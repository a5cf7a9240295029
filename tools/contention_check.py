#!/usr/bin/env python3
"""Do the train steps give the same bits when ANOTHER kernel stream competes for the GPU?  Each step (UNet B = 512, text
denoiser B = 64 / 256) is run quietly, then again while a side stream runs back-to-back fp32 GEMMs (torch.mm) that touch none of
its memory; gradients are compared bit for bit.  A difference means some kernel's result depends on timing (a missing wait or
barrier that an undisturbed run never exposes).    python tools/contention_check.py [--reps 3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinydiffusionmodels_amd import _lib, mnist as M, unet_engine as E   # noqa: E402
from tinydiffusionmodels_amd import transformer_engine as TE   # noqa: E402
from tinydiffusionmodels_amd.shakespeare import TinyTransformer, DenoiserTrainer   # noqa: E402


def noisy(fn, side, a, b, c, n):
    """fn() on the current stream while `side` runs n GEMMs"""
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(n):
            torch.mm(a, b, out=c)
    fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    _lib.check(L.tdm_set_bwd_overlap(0))
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev); c = torch.empty(4096, 4096, device=dev)
    # UNet
    torch.manual_seed(0)
    m = M.SimpleUNet().to(dev)
    tr = M.DDPMTrainer(m, 512, lr=1e-3, graph=False)
    x0 = torch.rand(512, 1, 28, 28, device=dev) * 2 - 1
    t = torch.randint(0, 1000, (512,), device=dev); nz = torch.randn(512, 1, 28, 28, device=dev)
    st = tr.state
    E.loss_and_grad(tr.flat, st, x0, nz, t); torch.cuda.synchronize()
    ref = st.grads.clone()
    for r in range(args.reps):
        noisy(lambda: E.loss_and_grad(tr.flat, st, x0, nz, t), side, a, b, c, 12)
        print(f"UNet B=512 rep {r}: grads equal under contention: {torch.equal(ref, st.grads)}", flush=True)
    # text
    for gm in (1, 2):
        _lib.check(L.tdm_set_gemm_mode(gm))
        for B in (64, 256):
            torch.manual_seed(0)
            tm = TinyTransformer(256, dropout=0.1).to(dev); tm.train()
            ttr = DenoiserTrainer(tm, B, 128, lr=1e-4, weight_decay=1e-4, graph=False)
            x = torch.randn(B, 128, 256, device=dev) * 0.02
            s = ttr.state
            rng0 = ttr.rng_state.clone()

            def step():
                ttr.rng_state.copy_(rng0)
                TE.tt_loss_and_grad_philox(ttr.flat, s, x, ttr.seed, ttr.rng_state, p_drop=0.1, drop_seed=ttr.drop_seed)
            step(); torch.cuda.synchronize()
            ref = s.grads.clone()
            step(); torch.cuda.synchronize()
            print(f"text gemm_mode {gm} B={B}: quiet repeat equal: {torch.equal(ref, s.grads)}", flush=True)
            for r in range(args.reps):
                noisy(step, side, a, b, c, 40 if B == 256 else 16)
                d = (ref - s.grads).abs().max().item()
                print(f"text gemm_mode {gm} B={B} rep {r}: grads equal under contention: {torch.equal(ref, s.grads)} (max diff {d:.2e})", flush=True)
    _lib.check(L.tdm_set_gemm_mode(1)); _lib.check(L.tdm_set_bwd_overlap(1))


if __name__ == "__main__":
    main()
