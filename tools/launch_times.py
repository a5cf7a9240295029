#!/usr/bin/env python3
"""Per-launch times of the default UNet train step at a given batch size (tdm_unet_replay_launch_f32):
    python tools/launch_times.py --B 512 [--B 256 ...] [--ids 5,6,7] [--iters 20]
Each launch is replayed alone on two alternating, fully populated workspaces and timed with HIP events."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (LAUNCH_WORK, time_events)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, action="append")
    ap.add_argument("--ids", type=str, default="")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from tinydiffusionmodels_amd import _lib, unet_engine as E
    from tinydiffusionmodels_amd.mnist import SimpleUNet
    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = SimpleUNet().to(dev)
    flat = model.flat.detach()
    nl = L.tdm_unet_launch_count()
    ids = [int(v) for v in args.ids.split(",")] if args.ids else list(range(nl))
    res = {}
    for B in (args.B or [512]):
        g = torch.Generator(device=dev).manual_seed(5)
        x0 = torch.rand(B, 1, 28, 28, device=dev, generator=g) * 2 - 1
        t = torch.randint(0, 1000, (B,), device=dev, generator=g)
        nz = torch.randn(B, 1, 28, 28, device=dev, generator=g)
        sts = [E.TrainState(flat, B), E.TrainState(flat, B)]
        for st in sts:
            E.loss_and_grad(flat, st, x0, nz, t)
        slabs = E.slabs_for(dev)
        gs = torch.empty_like(sts[0].grads)
        for lid in ids:
            def call(i, lid=lid):
                st = sts[i & 1]
                _lib.check(L.tdm_unet_replay_launch_f32(_lib.ptr(flat), _lib.ptr(st.x_noisy), _lib.ptr(t), _lib.ptr(st.eps),
                                                        _lib.ptr(st.deps), _lib.ptr(nz), _lib.ptr(gs), _lib.ptr(st.ws.ws), _lib.ptr(slabs), B, lid,
                                                        _lib.stream()), "replay")
            res[(B, lid)] = bench.time_events(call, args.iters) * 1e3
        del sts
    Bs = args.B or [512]
    print("id " + " ".join(f"{'B=' + str(b):>9s}" for b in Bs) + "  launch")
    for lid in ids:
        print(f"{lid:2d} " + " ".join(f"{res[(b, lid)]:9.2f}" for b in Bs) + "  " + L.tdm_unet_launch_name(lid).decode())
    print("sum" + " ".join(f"{sum(res[(b, l)] for l in ids):9.1f}" for b in Bs))


if __name__ == "__main__":
    main()
