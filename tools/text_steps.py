#!/usr/bin/env python3
"""A few text-denoiser train steps at config 5's size for profiling:  python tools/text_steps.py [dropout] [steps] [gemm_mode]"""
import sys
import torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
from tinydiffusionmodels_amd.shakespeare import TinyTransformer, DenoiserTrainer
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
_lib.check(_lib.lib().tdm_set_gemm_mode(int(sys.argv[3]) if len(sys.argv) > 3 else 1))
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = TinyTransformer(256, dropout=p).to(dev); m.train()
tr = DenoiserTrainer(m, 256, 128, lr=1e-4, graph=False)
x0 = torch.randn(256, 128, 256, device=dev) * 0.02
for _ in range(n):
    tr.step(x0)
torch.cuda.synchronize()
