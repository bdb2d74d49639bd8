#!/usr/bin/env python3
"""Is tgnh_create's rule for wave tiles (taken when they would be >= 90 % full) right for the configurations it turns away?
C3 (ionic liquid: 35-slot cations) and C4 (mixed box) with TGNH_FLAG_WAVE_TILES forced against the default (512-slot tiles),
deferred and one-launch variants, hipGraph replays of 50 steps, best of three, interleaved on one box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext
from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_WAVE_TILES, TgnhError
CFG = [("C3 ionic liquid 100k", lambda: synth.ionic_liquid(2222), 0.0), ("C4 mixed 500k + hard wall", lambda: synth.mixed(60000, 4444), 0.02),
       ("ionic liquid 1M", lambda: synth.ionic_liquid(22222), 0.0)]
for name, build, hw in CFG:
    s, g, ng = build()
    for var, base in (("plain", 0), ("defer", FLAG_DEFER_SCALE), ("resident", FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)):
        rates = {}
        ctxs = {}
        for wave in (False, True):
            it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
            it.setMaxDrudeDistance(hw)
            for _ in range(ng): it.addTempGroup()
            it._particleTempGroup = g.astype("int32")
            try:
                ctxs[wave] = HipContext(s, it, mode="TGNH", precision="mixed", flags=base | (FLAG_WAVE_TILES if wave else 0))
            except TgnhError as e:
                print(f"{name} {var} wave={wave}: refused ({str(e)[:120]})"); continue
            ctxs[wave].step(50)
        torch.cuda.synchronize()
        for rnd in range(3):
            for wave, ctx in ctxs.items():
                rep = ctx.capture_steps(50); rep(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10): rep()
                torch.cuda.synchronize()
                rates[wave] = max(rates.get(wave, 0.0), 500 / (time.perf_counter() - t0))
        info = {w: (c.num_wave_tiles() if hasattr(c, "num_wave_tiles") else None, c.resident_kernel()) for w, c in ctxs.items()}
        print(f"{name} ({s.num_particles} slots) {var}: default {rates.get(False, 0):.0f} steps/s, wave tiles forced {rates.get(True, 0):.0f}  {info}", flush=True)
        for c in ctxs.values(): c.close()
