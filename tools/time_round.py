"""Rounding head (src/shakespeare.py:239-240 + gradients): the fused chained-MFMA form (csrc/ce_chain.hip: logits in registers) against
fp64 torch at a small size and against the stored-logits / vocabulary-chunked forms at config-5 size (32,768 x 50,257, D = 256).
   python tools/time_round.py [--small-only] [--nseg 1,3,5]"""
import argparse
import sys

import torch

sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib   # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
ap = argparse.ArgumentParser()
ap.add_argument("--small-only", action="store_true")
ap.add_argument("--nseg", type=str, default="1,3,5")
args = ap.parse_args()


def t(f, it=5):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it


def make(M, V, D, seed=1, bscale=0.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(M, D, device=dev, generator=g) * 0.5
    W = torch.randn(V, D, device=dev, generator=g) / 16
    b = torch.randn(V, device=dev, generator=g) * bscale
    ids = torch.randint(0, V, (M,), device=dev, generator=g)
    return x, W, b, ids


def fused(x, W, b, ids, scale, nseg):
    M, D = x.shape; V = W.shape[0]
    loss, dx, dW, db = torch.empty(1, device=dev), torch.empty_like(x), torch.empty_like(W), torch.empty_like(b)
    ws = torch.empty(L.tdm_round_workspace_fused_floats(M, V, D, nseg), device=dev)
    f = lambda: _lib.check(L.tdm_round_ce_loss_grad_fused_f32(_lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(ids), scale, _lib.ptr(loss),   # noqa: E731
                                                              _lib.ptr(dx), _lib.ptr(dW), _lib.ptr(db), _lib.ptr(ws), M, V, D, nseg, _lib.stream()), "fused")
    return f, (loss, dx, dW, db), ws


rel = lambda a, r: ((a.double().cpu() - r.double().cpu()).abs().max() / r.double().cpu().abs().max()).item()   # noqa: E731
# ---- small sizes vs fp64 torch (ragged M and V: tails of both passes)
for (M, V, nseg) in ((300, 1003, 1), (1000, 2077, 3), (128, 64, 2)):
    x, W, b, ids = make(M, V, 256, seed=M, bscale=0.3)
    f, (loss, dx, dW, db), ws = fused(x, W, b, ids, 0.7, nseg)
    f(); torch.cuda.synchronize()
    xd, Wd, bd = (v.double().cpu().requires_grad_(True) for v in (x, W, b))
    lr = torch.nn.functional.cross_entropy(xd @ Wd.T + bd, ids.cpu())
    (0.7 * lr).backward()
    print(f"M={M} V={V} nseg={nseg}: loss {abs(loss.item() - lr.item()) / lr.item():.1e}, dx {rel(dx, xd.grad):.1e}, dW {rel(dW, Wd.grad):.1e}, db {rel(db, bd.grad):.1e}")
    del ws
if args.small_only:
    sys.exit(0)
M, V, D = 32768, 50257, 256
x, W, b, ids = make(M, V, D)
loss, dx, dW, db = torch.empty(1, device=dev), torch.empty_like(x), torch.empty_like(W), torch.empty_like(b)
n = L.tdm_round_workspace_floats(M, V, D); ws = torch.empty(n, device=dev)
ms = t(lambda: _lib.check(L.tdm_round_ce_loss_grad_f32(_lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(ids), 1.0, _lib.ptr(loss), _lib.ptr(dx), _lib.ptr(dW), _lib.ptr(db), _lib.ptr(ws), M, V, D, _lib.stream())))
print(f"stored logits: {ms:.2f} ms, workspace {n * 4 / 2**30:.2f} GiB, loss {loss.item():.4f}")
ref = (dx.clone(), dW.clone(), db.clone(), loss.item())
del ws
Vc = 8192
n = L.tdm_round_workspace_chunked_floats(M, V, D, Vc); ws = torch.empty(n, device=dev)
ms = t(lambda: _lib.check(L.tdm_round_ce_loss_grad_chunked_f32(_lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(ids), 1.0, _lib.ptr(loss), _lib.ptr(dx), _lib.ptr(dW), _lib.ptr(db), _lib.ptr(ws), M, V, D, Vc, _lib.stream())))
print(f"chunk {Vc}: {ms:.2f} ms, workspace {n * 4 / 2**30:.2f} GiB, loss {loss.item():.4f}, dx / dW vs stored form {rel(dx, ref[0]):.1e} / {rel(dW, ref[1]):.1e}")
del ws
for nseg in [int(v) for v in args.nseg.split(",")]:
    f, (loss, dx, dW, db), ws = fused(x, W, b, ids, 1.0, nseg)
    ms = t(f)
    print(f"fused (logits in registers), nseg {nseg}: {ms:.2f} ms, workspace {ws.numel() * 4 / 2**30:.2f} GiB, loss {loss.item():.4f} (stored form {ref[3]:.4f}), "
          f"dx / dW / db vs stored form {rel(dx, ref[0]):.1e} / {rel(dW, ref[1]):.1e} / {rel(db, ref[2]):.1e}")
    del ws
