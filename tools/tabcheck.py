import hashlib, numpy as np, torch, sys
sys.path.insert(0, ".")
from tinydiffusionmodels_amd.schedule import make_tables
t = make_tables()
g = np.load("tests/golden/schedule.npz")
print(torch.__config__.show().split("\n")[0:3], torch.backends.cpu.get_cpu_capability())
for k in g.files:
    a = t[k].numpy(); b = g[k]
    d = (a.view(np.int32) - b.view(np.int32))
    print(k, hashlib.sha256(a.tobytes()).hexdigest()[:16], "ndiff", int((d != 0).sum()), "max ulp", int(np.abs(d).max()))
