#!/usr/bin/env python3
"""ms per FULL text train step (TextTrainStep) at B sequences of 128 tokens: python tools/time_text_full.py [B=256] [steps=30] [graph=1]"""
import sys, time
import torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import shakespeare as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
graph = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
dev = torch.device("cuda:0")
torch.manual_seed(0)
V, D, L = 50257, 256, 128
m = S.TinyTransformer(D, dropout=0.1).to(dev); m.train()
emb, rnd = S.LearnedEmbedding(V, D).to(dev), S.LearnedRounding(D, V).to(dev)
st = S.TextTrainStep(m, rnd, emb, lr=1e-4, graph=graph)
g = torch.Generator(device=dev).manual_seed(3)
ids = torch.randint(0, V, (B, L), device=dev, generator=g)
for _ in range(5):
    st.step(ids)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(n):
        st.step(ids)
    torch.cuda.synchronize()
    print(f"B={B} graph={int(graph)} rep {rep}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)
