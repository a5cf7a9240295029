#!/usr/bin/env python3
"""Workload for rocprofv3 passes over single kernels of the train step: after one full step has filled two
workspaces, replays the chosen launches (tdm_unet_replay_launch_f32, in-pipeline arguments, alternating the
workspaces so the 256 MB Infinity Cache cannot serve one launch's inputs to the next) and one NT GEMM of the text
denoiser's FFN shape.

    python tools/pmc_replay.py [--iters 5] [--ids 15,8,2,...] [--B 512] [--names-out names.json]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--ids", type=str, default="15,8,2,12,6,13,11,17,10,16")
    ap.add_argument("--names-out", type=str, default=None)
    ap.add_argument("--B", type=int, default=512)
    args = ap.parse_args()
    from tinydiffusionmodels_amd import _lib, unet_engine as E
    from tinydiffusionmodels_amd.mnist import SimpleUNet
    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = SimpleUNet().to(dev)
    flat = model.flat.detach()
    B = args.B
    g = torch.Generator(device=dev).manual_seed(5)
    x0 = torch.rand(B, 1, 28, 28, device=dev, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), device=dev, generator=g)
    nz = torch.randn(B, 1, 28, 28, device=dev, generator=g)
    sts = [E.TrainState(flat, B), E.TrainState(flat, B)]
    for st in sts:
        E.loss_and_grad(flat, st, x0, nz, t)
    slabs = E.slabs_for(dev)
    gs = torch.empty_like(sts[0].grads)
    torch.cuda.synchronize()
    if args.names_out:
        import json
        json.dump({v: L.tdm_unet_launch_name(int(v)).decode() for v in args.ids.split(",")}, open(args.names_out, "w"))
    for lid in [int(v) for v in args.ids.split(",")]:
        print("replaying", lid, L.tdm_unet_launch_name(lid).decode(), flush=True)
        for i in range(args.iters):
            st = sts[i & 1]
            _lib.check(L.tdm_unet_replay_launch_f32(_lib.ptr(flat), _lib.ptr(st.x_noisy), _lib.ptr(t), _lib.ptr(st.eps),
                                                    _lib.ptr(st.deps), _lib.ptr(nz), _lib.ptr(gs), _lib.ptr(st.ws.ws), _lib.ptr(slabs), B, lid,
                                                    _lib.stream()), "replay")
        torch.cuda.synchronize()
    # text denoiser FFN1 shape on the NT bf16x3 GEMM in its in-pipeline form (gemm_nt_bf16_kernel<3,...,true>): pre-split
    # (S16) operands, ReLU, S16 output — linear1 of the encoder layer
    M, N, K = 32768, 2048, 256
    A = [torch.randn(M, K, device=dev) for _ in range(2)]
    Bm = torch.randn(N, K, device=dev) * 0.05
    A16 = [torch.empty_like(a) for a in A]
    B16 = torch.empty_like(Bm)
    for a, a16 in zip(A, A16):
        _lib.check(L.tdm_split_s16_f32(_lib.ptr(a), _lib.ptr(a16), a.numel(), _lib.stream()))
    _lib.check(L.tdm_split_s16_f32(_lib.ptr(Bm), _lib.ptr(B16), Bm.numel(), _lib.stream()))
    bias = torch.zeros(N, device=dev)
    C = [torch.empty(M, N, device=dev) for _ in range(2)]
    for i in range(args.iters):
        _lib.check(L.tdm_gemm_f32(_lib.ptr(A16[i & 1]), K, 1, _lib.ptr(B16), 1, K, _lib.ptr(C[i & 1]), N, _lib.ptr(bias), None, M, N, K,
                                  1 | 2 | 4, 1, 0, _lib.stream()))
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
