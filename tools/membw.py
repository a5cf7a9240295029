"""Achievable HBM bandwidth of plain torch fill / copy / read-reduce kernels (reference points for roofline fractions)."""
import torch
dev = torch.device("cuda:0")
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 268, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(x)); tr = t(lambda: x.sum())
    print(f"{mb:5d} MB: fill {mb/1024/tf/1e3*1.073741824:.2f} TB/s ({tf*1e6:.0f} us)  copy {2*mb/1024/tc/1e3*1.073741824:.2f} TB/s ({tc*1e6:.0f} us)  sum {mb/1024/tr/1e3*1.073741824:.2f} TB/s ({tr*1e6:.0f} us)")
