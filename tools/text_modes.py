#!/usr/bin/env python3
"""Text denoiser train step (DenoiserTrainer) in four issue modes: hipGraph replay / eager launches, with and without the
backward's side queue (tdm_set_bwd_overlap); --check compares the weights after the run across modes and runs (same draws).
    python tools/text_modes.py [--B 32 --B 256] [--steps 60] [--gemm-mode 1] [--check]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinydiffusionmodels_amd import _lib   # noqa: E402
from tinydiffusionmodels_amd.shakespeare import TinyTransformer, DenoiserTrainer   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, action="append")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--gemm-mode", type=int, default=1)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    _lib.check(L.tdm_set_gemm_mode(args.gemm_mode))
    modes = [("graph", True, 0), ("graph+side", True, 1), ("eager", False, 0), ("eager+side", False, 1)]
    torch.manual_seed(123)
    for B in (args.B or [32, 256]):
        x = torch.randn(B, 128, 256, device=dev) * 0.02
        res, finals = {}, {}
        for rep in range(args.reps):
            for name, graph, ov in modes:
                _lib.check(L.tdm_set_bwd_overlap(ov), "overlap")
                torch.manual_seed(0)
                m = TinyTransformer(256, dropout=0.1).to(dev)
                m.train()
                tr = DenoiserTrainer(m, B, 128, lr=1e-4, weight_decay=1e-4, graph=graph)
                for _ in range(4):
                    tr.step(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    tr.step(x)
                torch.cuda.synchronize()
                res.setdefault(name, []).append((time.perf_counter() - t0) / args.steps * 1e3)
                finals.setdefault(name, []).append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone())
                del tr, m
        line = f"B={B:4d}: " + "  ".join(f"{k} {min(v):.4f} ms" for k, v in res.items())
        if args.check:
            ref = finals["eager"][0]
            line += "  | equal to eager: " + " ".join(f"{k}={all(bool(torch.equal(v, ref)) for v in vs)}" for k, vs in finals.items())
        print(line, flush=True)
    L.tdm_set_bwd_overlap(1)


if __name__ == "__main__":
    main()
