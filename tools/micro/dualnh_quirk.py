#!/usr/bin/env python3
"""dualNH WITHOUT useDrudeNHChains (the reference's C++ default: the coupled chain of Ref :476-503) and 2-4 links: steps/s at C2
(32 k slots) and at the 8-GPU shard size (625 k), the reference's pass structure (chain_kernel) and the deferred one (the chain
inside the rescale launch), hipGraph replay, best of three.  Run once with TGNH_LIB=<a build before dualnh_quirk_fast> and once
without (round 5: profiles/r05_dualnh_quirk.txt)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext
from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE
for mols in (6400, 125000):
    s, g, ng = synth.water_box(mols)
    for flags, name in ((0, "plain"), (FLAG_DEFER_SCALE, "defer")):
        for chains in (2, 3, 4, 10):
            it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, False, True)
            it.setMaxDrudeDistance(0.02)
            ctx = HipContext(s, it, mode="dualNH", precision="mixed", flags=flags)
            ctx.step(60); torch.cuda.synchronize()
            best = 0.0
            for _ in range(3):
                rep = ctx.capture_steps(10); rep(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100): rep()
                torch.cuda.synchronize(); best = max(best, 1000 / (time.perf_counter() - t0))
            print(f"dualNH coupled chain, {5 * mols} slots, {name}, {chains} links: {best:.0f} steps/s ({os.environ.get('TGNH_LIB', 'this build')})", flush=True)
            ctx.close()
