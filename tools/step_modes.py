#!/usr/bin/env python3
"""UNet train step (epoch mode) at several batch sizes in the three issue modes: one hipGraph replay per step(s) without the
backward's side stream, eager launches without it, eager launches with it (tdm_set_bwd_overlap).
    python tools/step_modes.py [--B 64 --B 512] [--steps 300]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinydiffusionmodels_amd import _lib, mnist as M   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, action="append")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    modes = [("graph", True, 0), ("graph+side", True, 1), ("eager", False, 0), ("eager+side", False, 1)]
    for B in (args.B or [64, 128, 256, 512]):
        data = torch.rand(32 * B, 1, 28, 28, device=dev) * 2 - 1
        perm = torch.randperm(32 * B).to(dev)
        res = {}
        for rep in range(args.reps):
            for name, graph, ov in modes:
                _lib.check(L.tdm_set_bwd_overlap(ov), "overlap")
                torch.manual_seed(0)
                m = M.SimpleUNet().to(dev)
                tr = M.DDPMTrainer(m, B, lr=1e-3, graph=graph)

                def run(n):
                    while n > 0:
                        tr.begin_epoch(data, perm)
                        k = min(n, 32)
                        tr.steps_epoch(k)
                        n -= k
                run(40)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(args.steps)
                torch.cuda.synchronize()
                res.setdefault(name, []).append((time.perf_counter() - t0) / args.steps * 1e3)
                del tr, m
        print(f"B={B:4d}: " + "  ".join(f"{k} {min(v):.4f} ms" for k, v in res.items()), flush=True)
    L.tdm_set_bwd_overlap(1)


if __name__ == "__main__":
    main()
