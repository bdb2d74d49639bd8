#!/usr/bin/env python3
"""Do two streams read in lockstep conflict in HBM depending on the distance between their bases?
c[i] = a[i] + b[i] over 160 MB arrays carved from one pool; the distance base(b) - base(a) is varied."""
import torch
dev = torch.device("cuda:0")
MiB = 1 << 20
n = 40_000_000                      # float32 elements: 160 MB per array
pool = torch.zeros(6 * 1024 * MiB, dtype=torch.uint8, device=dev)
base = (-pool.data_ptr()) % (2 * MiB)


def view(off_bytes):
    return pool[base + off_bytes: base + off_bytes + 4 * n].view(torch.float32)


def timeit(a, b, c, reps=30):
    for _ in range(5):
        torch.add(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.add(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


size = 4 * n
span = ((size + 2 * MiB - 1) // (2 * MiB)) * 2 * MiB          # 2 MiB-rounded array size (154 MiB)
a = view(0)
c_off = 4096 * MiB
c = view(c_off)
print(f"array {size / MiB:.1f} MiB; a at 0, c at {c_off // MiB} MiB", flush=True)
for d in [span, span + 2 * MiB, span + 6 * MiB, 160 * MiB, 192 * MiB, 256 * MiB, 258 * MiB, 320 * MiB, 384 * MiB, 512 * MiB, 514 * MiB, 640 * MiB,
          768 * MiB, 1024 * MiB, 1026 * MiB, 1536 * MiB, 2048 * MiB, 2050 * MiB, 3072 * MiB]:
    b = view(d)
    us = timeit(a, b, c)
    print(f"b at +{d / MiB:7.1f} MiB: {us:7.1f} us  {3 * size / us / 1e6:6.2f} TB/s", flush=True)
