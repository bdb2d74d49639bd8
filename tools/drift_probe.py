#!/usr/bin/env python3
"""Does the dominant launch's duration drift in time (clock ramp, power state)?  tools/drift_probe.py [molecules]
One context, windows of 200 steps back to back, then the same after idle gaps of 0.5-5 s."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from openmm_drudenose_amd import synth, _lib
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext, FLAG_DEFER_SCALE

mol = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
system, group, ngroups = synth.water_box(mol)
it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
it.setMaxDrudeDistance(0.02)
ctx = HipContext(system, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE)


def window(tag, steps=200):
    ctx.timing(2 + 0)
    t0 = time.perf_counter()
    ctx.step(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.timing(False)
    ms, n = ctx.timing_read(0)
    print(f"{tag}: {steps / dt:8.1f} steps/s | scale+kick+drift {ms / n * 1e3:7.2f} us", flush=True)


for i in range(12):
    window(f"back to back {i:2d}")
for gap in (0.5, 1.0, 2.0, 5.0, 5.0):
    time.sleep(gap)
    window(f"after {gap:3.1f} s idle, first 100 steps", 100)
    window("                  next 200 steps      ")
