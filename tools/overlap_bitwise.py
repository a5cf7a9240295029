#!/usr/bin/env python3
"""Two-queue backward vs one queue, bit for bit, over many steps at several batch sizes (small ones keep whole tensors in L2
across steps — the case a missing cache acquire would show up in):  python tools/overlap_bitwise.py [--B 4 --B 512] [--steps 100]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinydiffusionmodels_amd import _lib, mnist as M   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, action="append")
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    bad = 0
    for B in (args.B or [1, 2, 4, 8, 16, 25, 37, 64, 128, 512]):
        data = torch.rand(8 * B, 1, 28, 28, device=dev) * 2 - 1
        perm = torch.randperm(8 * B).to(dev)
        out = []
        for ov in (0, 1, 1):
            _lib.check(L.tdm_set_bwd_overlap(ov))
            torch.manual_seed(0)
            m = M.SimpleUNet().to(dev)
            tr = M.DDPMTrainer(m, B, lr=1e-3, graph=False)
            n = args.steps
            while n > 0:
                tr.begin_epoch(data, perm)
                k = min(n, 8)
                tr.steps_epoch(k)
                n -= k
            torch.cuda.synchronize()
            out.append(m.flat.detach().clone())
        ok = torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])
        bad += 0 if ok else 1
        print(f"B={B:4d} steps={args.steps}: two queues == one queue (two runs): {ok}", flush=True)
    _lib.check(L.tdm_set_bwd_overlap(1))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
