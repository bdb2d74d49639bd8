#!/usr/bin/env python3
"""Where should chains of 2-4 links leave the streaming launches for chain_kernel?  tgnh_create keeps them inside below 1 M slots
(instantiations with the links' registers: two work-groups per compute unit instead of three / four).  With every thermostat in
one pass (chain_both_fast, round 4) the chain inside a launch takes half the time it did when the limit was set: the limit again,
three links, TGNH_INLINE_MULTI_MAX of a -DTGNH_TUNING build (TGNH_LIB=build_variants/lib_tuning.so) at 0 (never inside) and 2^30
(always), 625 k to 5 M slots, hipGraph replay, best of three."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext
from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP
assert os.environ.get("TGNH_LIB"), "run with TGNH_LIB=build_variants/lib_tuning.so (a -DTGNH_TUNING build reads the knob)"
for mols in (125000, 250000, 400000, 1000000):
    s, g, ng = synth.water_box(mols)
    for chains in (3, 2, 4) if mols in (125000, 1000000) else (3,):
        for var, fl in (("resident", FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP), ("defer", FLAG_DEFER_SCALE)):
            row = []
            for limit in ("0", str(1 << 30)):
                os.environ["TGNH_INLINE_MULTI_MAX"] = limit
                it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, True, True)
                it.setMaxDrudeDistance(0.02)
                for _ in range(ng): it.addTempGroup()
                it._particleTempGroup = g.astype("int32")
                ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=fl)
                ctx.step(60); torch.cuda.synchronize()
                best = 0.0
                n = 40 if mols >= 400000 else 100
                for _ in range(3):
                    rep = ctx.capture_steps(10); rep(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n): rep()
                    torch.cuda.synchronize(); best = max(best, 10 * n / (time.perf_counter() - t0))
                row.append(best)
                ctx.close()
            print(f"{5 * mols} slots, {chains} links, {var}: chain_kernel {row[0]:.0f}  inside the launches {row[1]:.0f} steps/s  ({row[1] / row[0] - 1:+.1%})", flush=True)
