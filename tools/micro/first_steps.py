#!/usr/bin/env python3
"""How the one-launch step's duration settles over the first steps of a fresh context (metric size): HIP events, chunks of 5 steps."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext, _lib
from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP
s, g, ng = synth.water_box(1000000)
it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
it.setMaxDrudeDistance(0.02)
it.addTempGroup(); it._particleTempGroup = g.astype("int32")
ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)
done = 0
for chunk in [5] * 12 + [50] * 4 + [200] * 3:
    ctx.timing(True); ctx.step(chunk); torch.cuda.synchronize(); ctx.timing(False)
    ms, n = ctx.timing_read(_lib.KID_STEP); fm, fn = ctx.timing_read(_lib.KID_FORCE)
    done += chunk
    print(f"steps {done - chunk + 1:5d}-{done:5d}: step kernel {ms / n * 1e3:7.2f} us  force {fm / fn * 1e3:6.2f} us", flush=True)
ctx.close()
