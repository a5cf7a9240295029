#!/usr/bin/env python3
"""Text denoiser step (and the UNet step) next to a side stream of the library's OWN token-major (weight-gradient) GEMMs on unrelated buffers:
gradients compared bit for bit with the quiet step.  python tools/contention_tn.py [--B 256] [--reps 4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinydiffusionmodels_amd import _lib   # noqa: E402
from tinydiffusionmodels_amd import transformer_engine as TE   # noqa: E402
from tinydiffusionmodels_amd.shakespeare import TinyTransformer, DenoiserTrainer   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--n-side", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    _lib.check(L.tdm_set_gemm_mode(1))
    B = args.B
    M = 32768
    # side work: dW[2048][256] = dY[M][2048]^T X[M][256], S16 operands, 8 splits (the linear1 weight gradient's shape)
    dy = torch.randn(M, 2048, device=dev) * 0.01
    xx = torch.randn(M, 256, device=dev)
    dy16, x16 = torch.empty_like(dy), torch.empty_like(xx)
    _lib.check(L.tdm_split_s16_f32(_lib.ptr(dy), _lib.ptr(dy16), dy.numel(), _lib.stream()))
    _lib.check(L.tdm_split_s16_f32(_lib.ptr(xx), _lib.ptr(x16), xx.numel(), _lib.stream()))
    slab = torch.empty(8, 2048, 256, device=dev)
    side = torch.cuda.Stream()

    def side_work(n):
        with torch.cuda.stream(side):
            for _ in range(n):
                _lib.check(L.tdm_gemm_f32(_lib.ptr(dy16), 1, 2048, _lib.ptr(x16), 256, 1, _lib.ptr(slab), 256, None, None, 2048, 256, M, 2, 8,
                                          2048 * 256, side.cuda_stream), "tn gemm")
    torch.manual_seed(0)
    tm = TinyTransformer(256, dropout=0.1).to(dev)
    tm.train()
    ttr = DenoiserTrainer(tm, B, 128, lr=1e-4, weight_decay=1e-4, graph=False)
    x = torch.randn(B, 128, 256, device=dev) * 0.02
    s = ttr.state
    rng0 = ttr.rng_state.clone()

    def step():
        ttr.rng_state.copy_(rng0)
        TE.tt_loss_and_grad_philox(ttr.flat, s, x, ttr.seed, ttr.rng_state, p_drop=0.1, drop_seed=ttr.drop_seed)
    step(); torch.cuda.synchronize()
    ref = s.grads.clone()
    side_work(2); torch.cuda.synchronize()
    sref = slab.clone()
    for r in range(args.reps):
        side.wait_stream(torch.cuda.current_stream())
        side_work(args.n_side)
        step()
        torch.cuda.synchronize()
        d = (ref - s.grads).abs().max().item()
        print(f"B={B} rep {r}: step grads equal next to TN GEMMs: {torch.equal(ref, s.grads)} (max diff {d:.2e}); side GEMM result equal: {torch.equal(sref, slab)}", flush=True)
    # the UNet train step (B = 512, one and two queues in the backward) next to the same side work
    from tinydiffusionmodels_amd import mnist as MN, unet_engine as E
    for ov in (0, 1):
        _lib.check(L.tdm_set_bwd_overlap(ov))
        torch.manual_seed(0)
        m = MN.SimpleUNet().to(dev)
        tr = MN.DDPMTrainer(m, 512, lr=1e-3, graph=False)
        x0 = torch.rand(512, 1, 28, 28, device=dev) * 2 - 1
        t = torch.randint(0, 1000, (512,), device=dev)
        nz = torch.randn(512, 1, 28, 28, device=dev)
        us = tr.state
        E.loss_and_grad(tr.flat, us, x0, nz, t)
        torch.cuda.synchronize()
        uref = us.grads.clone()
        ok = []
        for r in range(args.reps):
            side.wait_stream(torch.cuda.current_stream())
            side_work(10)
            E.loss_and_grad(tr.flat, us, x0, nz, t)
            torch.cuda.synchronize()
            ok.append(torch.equal(uref, us.grads))
        print(f"UNet B=512, {'two queues' if ov else 'one queue'}: grads equal next to TN GEMMs: {ok}", flush=True)
    _lib.check(L.tdm_set_bwd_overlap(1))


if __name__ == "__main__":
    main()
