#!/usr/bin/env python3
"""Device copy time vs size (what a streaming launch of a given footprint can cost at best on this box)."""
import torch
dev = torch.device("cuda:0")
for mb in (8, 16, 32, 48, 64, 96, 128, 256, 512, 1024):
    n = mb * (1 << 20) // 2          # mb = bytes read + bytes written
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    for _ in range(20): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): b.copy_(a)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 100
    print(f"{mb:5d} MB moved (r+w): {us:7.1f} us  {mb * 1.048576 / us * 1e3 / 1e3:6.2f} TB/s", flush=True)
