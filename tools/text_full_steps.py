#!/usr/bin/env python3
"""A few FULL text train steps (TextTrainStep: embedding + denoiser + rounding head at V = 50,257 + AdamW on all tensors) for profiling:
   python tools/text_full_steps.py [B=32] [steps=6] [graph=0]"""
import sys
import torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import shakespeare as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
graph = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
dev = torch.device("cuda:0")
torch.manual_seed(0)
V, D, L = 50257, 256, 128
m = S.TinyTransformer(D, dropout=0.1).to(dev); m.train()
emb, rnd = S.LearnedEmbedding(V, D).to(dev), S.LearnedRounding(D, V).to(dev)
st = S.TextTrainStep(m, rnd, emb, lr=1e-4, graph=graph)
g = torch.Generator(device=dev).manual_seed(3)
ids = torch.randint(0, V, (B, L), device=dev, generator=g)
for _ in range(n):
    st.step(ids)
torch.cuda.synchronize()
print("losses", st.losses.tolist())
