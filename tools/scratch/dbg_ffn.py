import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import time_ffn as T
from tinydiffusionmodels_amd import _lib
dev = torch.device('cuda:0'); L = _lib.lib()
M, D, F = 384, 256, 2048
g = torch.Generator(device=dev).manual_seed(3)
gy = torch.randn(M, D, device=dev, generator=g)
W1 = torch.randn(F, D, device=dev, generator=g) / 16
W2 = torch.randn(D, F, device=dev, generator=g) / 16
mask = torch.zeros(L.tdm_ffn_chain_mask_count(M, F), dtype=torch.int16, device=dev)
# random masks
mk = torch.randint(0, 65536, mask.shape, generator=torch.Generator().manual_seed(5)).to(torch.int32)
mask.copy_(torch.from_numpy(mk.numpy().astype(np.uint16).view(np.int16)).to(dev))
gy16, w2t16, w1t16 = T.s16(gy), T.s16(W2.t().contiguous()), T.s16(W1.t().contiguous())
dz16, dx = torch.empty(M, F, device=dev), torch.empty(M, D, device=dev)
T.chain(2, 3, gy16, w2t16, None, w1t16, None, dx, dz16, mask, 2.0, 0.0, 0, M, D, F)
torch.cuda.synchronize()
mkn = mk.numpy().reshape(-1, F // 32, 64)
gate = np.zeros((M, F), dtype=bool)
for tb in range(M // 32):
    for fb in range(F // 32):
        for lane in range(64):
            w = int(mkn[tb, fb, lane]); tok = tb * 32 + (lane & 31)
            for r in range(16):
                gate[tok, 32 * fb + 16 * (r >> 3) + 8 * (lane >> 5) + (r & 7)] = (w >> r) & 1
dz_ref = (gy.double().cpu() @ W2.double().cpu()) * torch.from_numpy(gate) * 2.0
got = T.from_s16(dz16, M, F).double()
err = (got - dz_ref).abs()
print('dz max err', err.max().item(), 'ref max', dz_ref.abs().max().item())
bad = (err > 1e-3).numpy()
print('bad count', bad.sum(), 'of', bad.size)
if bad.sum():
    idx = np.argwhere(bad)
    print('bad by hidden block', np.bincount(idx[:, 1] // 32, minlength=F // 32))
    print('bad by token block', np.bincount(idx[:, 0] // 32, minlength=M // 32))
    # is it a gating problem? compare with ungated
    ung = (gy.double().cpu() @ W2.double().cpu()) * 2.0
    g_got = (got.abs() > 0).numpy()
    print('gate mismatches', (g_got != gate).sum(), ' ungated-value err where got nonzero', (torch.where(torch.from_numpy(g_got), got - ung, torch.zeros_like(got))).abs().max().item())
dx_ref = dz_ref @ W1.double().cpu()
print('dx rel err', ((dx.cpu().double() - dx_ref).abs().max() / dx_ref.abs().max()).item())
