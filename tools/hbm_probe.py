#!/usr/bin/env python3
"""Achievable HBM bandwidth on this box (torch copy / fill / read-reduce of 300 MB..1.2 GB fp32 buffers): the yardstick for the
HBM-bound 1x1 layers (NiN cccp1/2: 297 MB in + 297 MB out)."""
import torch, time
torch.cuda.set_device(0)
for mb in (297, 1188):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    def t(f, it=20):
        for _ in range(5): f()
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): f()
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
    c = t(lambda: y.copy_(x)); f = t(lambda: y.fill_(1.0)); r = t(lambda: x.sum())
    print(f"{mb} MB: copy {c:.4f} ms = {2*mb*1.048576/c:.0f} GB/s (R+W) | fill {f:.4f} ms = {mb*1.048576/f:.0f} GB/s (W) | sum {r:.4f} ms = {mb*1.048576/r:.0f} GB/s (R)")
