"""Achieved forward / gradient error of the default (bf16x3, S16) pipeline against the committed goldens."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from tinydiffusionmodels_amd import unet_engine as E, mnist
from oracle import ddpm_oracle as O
dev = torch.device("cuda:0")
g = {k: torch.from_numpy(v) for k, v in np.load("tests/golden/unet_forward.npz").items()}
model = mnist.SimpleUNet().to(dev)
sd = {k[2:]: v for k, v in g.items() if k.startswith("w.")}
model.load_state_dict(sd)
x, t = g["x_noisy"].to(dev), g["t"].to(dev)
ws = E.UNetWorkspace(x.shape[0], dev, training=True)
eps = E.unet_forward(model.flat.detach(), x, t, ws, save=True)
print("forward rel err (max|d| / max|ref|):", {k: float(O.rel_err(E.get_activation(ws, k).cpu(), g[k])) for k in ("h1", "h2", "h3", "h4")},
      "eps", float(O.rel_err(eps.cpu(), g["eps"])))
