#!/usr/bin/env python3
"""Is the 90 % fill rule for wave tiles (tgnh_create) the right one?  BASELINE configs 3 and 4 with the wave-tile kernels forced
(TGNH_FLAG_WAVE_TILES) against the library's choice, the 512-slot tile kernels.  Round 4: C3 (fill 0.70) 42.1 k against 39.7 k forced,
C4 (fill 0.83) 23.8 k against 23.6 k (defer: 23.3 k against 21.2 k) -- the rule stands."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext
from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_WAVE_TILES
for name, build, hw in (("C3 il 100k", lambda: synth.ionic_liquid(2222), 0.0), ("C4 mixed 500k", lambda: synth.mixed(60000, 4444), 0.02)):
    s, g, ng = build()
    for wave in (0, FLAG_WAVE_TILES):
        for var, fl in (("defer", FLAG_DEFER_SCALE), ("resident", FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)):
            it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
            it.setMaxDrudeDistance(hw)
            for _ in range(ng): it.addTempGroup()
            it._particleTempGroup = g.astype("int32")
            ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=fl | wave)
            ctx.step(50); torch.cuda.synchronize()
            best = 0.0
            for _ in range(3):
                rep = ctx.capture_steps(10); rep(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50): rep()
                torch.cuda.synchronize(); best = max(best, 500 / (time.perf_counter() - t0))
            wt = ctx.topology(9).reshape(-1, 2)
            fill = s.num_particles / (64.0 * (len(wt) - 1)) if len(wt) > 1 else 0
            print(f"{name} wave_flag={bool(wave)} {var}: {best:.0f} steps/s  kernel={ctx.resident_kernel()} wave-tile fill {fill:.2f}", flush=True)
            ctx.close()
