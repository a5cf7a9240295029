"""Rounding head at config-5 size (32,768 tokens x V = 50,257, D = 256): stored-logits form vs vocabulary-chunked forms."""
import sys, torch
sys.path.insert(0, ".")
from tinydiffusionmodels_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
M, V, D = 32768, 50257, 256
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(M, D, device=dev, generator=g) * 0.5; W = torch.randn(V, D, device=dev, generator=g) / 16; b = torch.zeros(V, device=dev)
ids = torch.randint(0, V, (M,), device=dev, generator=g)
loss, dx, dW, db = torch.empty(1, device=dev), torch.empty_like(x), torch.empty_like(W), torch.empty_like(b)
def t(f, it=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
n = L.tdm_round_workspace_floats(M, V, D); ws = torch.empty(n, device=dev)
ms = t(lambda: _lib.check(L.tdm_round_ce_loss_grad_f32(_lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(ids), 1.0, _lib.ptr(loss), _lib.ptr(dx), _lib.ptr(dW), _lib.ptr(db), _lib.ptr(ws), M, V, D, _lib.stream())))
print(f"stored logits: {ms:.2f} ms, workspace {n * 4 / 2**30:.2f} GiB, loss {loss.item():.4f}")
ref = (dx.clone(), dW.clone())
del ws
for Vc in (2048, 4096, 8192, 16384):
    n = L.tdm_round_workspace_chunked_floats(M, V, D, Vc); ws = torch.empty(n, device=dev)
    ms = t(lambda: _lib.check(L.tdm_round_ce_loss_grad_chunked_f32(_lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(ids), 1.0, _lib.ptr(loss), _lib.ptr(dx), _lib.ptr(dW), _lib.ptr(db), _lib.ptr(ws), M, V, D, Vc, _lib.stream())))
    e1 = ((dx - ref[0]).abs().max() / ref[0].abs().max()).item(); e2 = ((dW - ref[1]).abs().max() / ref[1].abs().max()).item()
    print(f"chunk {Vc}: {ms:.2f} ms, workspace {n * 4 / 2**30:.2f} GiB, loss {loss.item():.4f}, dx / dW vs stored form {e1:.1e} / {e2:.1e}")
    del ws
