#!/usr/bin/env python3
"""dualNH with chains of 2-4 links inside the launches (useDrudeNHChains): C2 (32 k slots) and the 8-GPU shard size (625 k), one
launch per step, hipGraph replay, best of three.  Run once with TGNH_LIB=<a build before run_dualnh_pair> and once without."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext
from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP
for mols in (6400, 125000):
    s, g, ng = synth.water_box(mols)
    for chains in (1, 2, 3, 4):
        it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, True, True)
        it.setMaxDrudeDistance(0.02)
        ctx = HipContext(s, it, mode="dualNH", precision="mixed", flags=FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)
        ctx.step(60); torch.cuda.synchronize()
        best = 0.0
        for _ in range(3):
            rep = ctx.capture_steps(10); rep(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100): rep()
            torch.cuda.synchronize(); best = max(best, 1000 / (time.perf_counter() - t0))
        print(f"dualNH {5 * mols} slots, {chains} links: {best:.0f} steps/s ({os.environ.get('TGNH_LIB', 'this build')})", flush=True)
        ctx.close()
