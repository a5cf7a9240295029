#!/usr/bin/env python3
"""The FULL text train step (TextTrainStep, V = 50,257) as a hipGraph replay and issued eagerly with the backward's side queue:
    python tools/text_full_modes.py [--B 32 --B 256] [--steps 40]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinydiffusionmodels_amd import _lib, shakespeare as S   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, action="append")
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    V, D, Lq = 50257, 256, 128
    for B in (args.B or [32, 256]):
        ids = torch.randint(0, V, (B, Lq), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        res = {}
        for rep in range(2):
            for name, graph, ov in (("graph", True, 0), ("eager", False, 0), ("eager+side", False, 1)):
                _lib.check(L.tdm_set_bwd_overlap(ov))
                torch.manual_seed(0)
                m = S.TinyTransformer(D, dropout=0.1).to(dev)
                m.train()
                emb, rnd = S.LearnedEmbedding(V, D).to(dev), S.LearnedRounding(D, V).to(dev)
                st = S.TextTrainStep(m, rnd, emb, lr=1e-4, graph=graph)
                for _ in range(4):
                    st.step(ids)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    st.step(ids)
                torch.cuda.synchronize()
                res.setdefault(name, []).append((time.perf_counter() - t0) / args.steps * 1e3)
                del st, m, emb, rnd
        print(f"B={B:4d}: " + "  ".join(f"{k} {min(v):.4f} ms" for k, v in res.items()), flush=True)
    L.tdm_set_bwd_overlap(1)


if __name__ == "__main__":
    main()
