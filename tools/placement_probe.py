#!/usr/bin/env python3
"""Is a launch's speed a property of WHERE the driver put the buffers?  tools/placement_probe.py [molecules] [pools]
Carves all state arrays from a fresh pool (earlier pools stay allocated, so every pool is different physical memory),
times the streaming launches on each, then goes back over the pools to see whether the figure sticks to the pool."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from openmm_drudenose_amd import synth, _lib
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext, FLAG_DEFER_SCALE, _check

mol = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
npools = int(sys.argv[2]) if len(sys.argv) > 2 else 10
system, group, ngroups = synth.water_box(mol)
it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
it.setMaxDrudeDistance(0.02)
ctx = HipContext(system, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE)
names = ["velm", "force", "posq", "posq_corr", "x0", "pos_delta"]
state = {n: getattr(ctx, n).clone() for n in names}
MB2 = 2 << 20
slots, off = {}, 0
for n in names:
    slots[n] = off
    off += (state[n].numel() * state[n].element_size() + MB2 - 1) // MB2 * MB2
pools = []


def use(pool):
    base = (-pool.data_ptr()) % MB2
    for n in names:
        o = state[n]
        nbytes = o.numel() * o.element_size()
        v = pool[base + slots[n]: base + slots[n] + nbytes].view(o.dtype).view(o.shape)
        v.copy_(o)
        setattr(ctx, n, v)
    _check(ctx.lib.tgnh_bind_buffers(ctx.h, ctx.posq.data_ptr(), ctx.posq_corr.data_ptr(), ctx.velm.data_ptr(),
                                     ctx.force.data_ptr(), ctx.pos_delta.data_ptr()))


def measure(tag):
    ctx.step(40)
    torch.cuda.synchronize()
    ctx.timing(True)
    ctx.step(200)
    torch.cuda.synchronize()
    ctx.timing(False)
    ks = {kid: ctx.timing_read(kid) for kid in (_lib.KID_SKD, _lib.KID_KICK_KE, _lib.KID_FORCE)}
    for n in names:
        state[n] = getattr(ctx, n).clone()
    print(f"{tag}: " + " | ".join(f"{_lib.KERNEL_NAMES[k]} {ms / n * 1e3:7.2f}" for k, (ms, n) in ks.items()), flush=True)


for i in range(npools):
    pools.append(torch.zeros(off + MB2, dtype=torch.uint8, device=ctx.dev))
    use(pools[-1])
    measure(f"pool {i:2d} @ {pools[-1].data_ptr():#x}")
for i in range(npools):
    use(pools[i])
    measure(f"again {i:2d} @ {pools[i].data_ptr():#x}")
