#!/usr/bin/env python3
"""Dev: what plain streaming reads reach on this box (torch reductions / copies over a 2.4 GB tensor)."""
import torch
x = torch.empty(1200 * 1024 * 1024, dtype=torch.bfloat16, device="cuda").normal_()
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
nb = x.numel() * 2
xi = x.view(torch.int32)
for name, fn, bytes_ in (("sum bf16", lambda: x.sum(), nb), ("sum int32 view", lambda: xi.sum(), nb), ("max", lambda: x.max(), nb),
                         ("copy (r+w)", lambda: y.copy_(x), 2 * nb)):
    s = t(fn)
    print(f"{name:16s} {bytes_ / s / 1e12:6.2f} TB/s  ({s * 1e3:.2f} ms)")
