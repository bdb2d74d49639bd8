#!/usr/bin/env python3
"""What the molecular centre-of-mass decomposition costs the one-launch step: the same box with and without the COM temperature
group (useCOMTempGroup), hipGraph, by size."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from openmm_drudenose_amd import synth, _lib
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP

for mol in [int(x) for x in sys.argv[1:]] or [6400, 125000, 1000000]:
    s, g, ng = synth.water_box(mol)
    for com in (True, False):
        it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, com)
        it.setMaxDrudeDistance(0.02)
        ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)
        ctx.step(50); torch.cuda.synchronize()
        rep = ctx.capture_steps(10); rep(); torch.cuda.synchronize()
        best = 0.0
        for _ in range(3):
            n = 2000 if mol < 500000 else 500
            t0 = time.perf_counter()
            for _ in range(n // 10): rep()
            torch.cuda.synchronize()
            best = max(best, n / (time.perf_counter() - t0))
        ctx.timing(True); ctx.step(100); torch.cuda.synchronize(); ctx.timing(False)
        ms, k = ctx.timing_read(_lib.KID_STEP)
        print(f"{mol:8d} molecules, COM group {'on ' if com else 'off'}: {best:8.0f} steps/s, step kernel {ms * 1e3 / max(k, 1):6.2f} us", flush=True)
        ctx.close()
