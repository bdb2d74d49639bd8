#!/usr/bin/env python3
"""What a chain of C links costs a step (GPU): C2 (32 k slots) and the metric size, one launch per step + chain_kernel /
chain_long_kernel<C>, eagerly and under a hipGraph; the chain's own launch time from HIP events.  -> profiles/r04_chain_cost.md"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext, _lib
from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP
print("| system | numNHChains | steps/s eager | steps/s hipGraph | chain launch us (events) | launches per step |")
print("|---|---|---|---|---|---|")
for name, mols, hw in (("C2 SWM4 32k", 6400, 0.0), ("metric 5 M slots", 1000000, 0.02)):
    s, g, ng = synth.water_box(mols)
    for chains in (1, 3, 5, 10, 16):
        it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, True, True)
        it.setMaxDrudeDistance(hw)
        for _ in range(ng): it.addTempGroup()
        it._particleTempGroup = g.astype("int32")
        ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)
        ctx.step(50); torch.cuda.synchronize()
        n, eager, graph = 400, 0.0, 0.0
        for _ in range(2):
            t0 = time.perf_counter(); ctx.step(n); torch.cuda.synchronize(); eager = max(eager, n / (time.perf_counter() - t0))
            rep = ctx.capture_steps(10); rep(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n // 10): rep()
            torch.cuda.synchronize(); graph = max(graph, n / (time.perf_counter() - t0))
        ctx.timing(True); ctx.step(100); torch.cuda.synchronize(); ctx.timing(False)
        ms, cnt = ctx.timing_read(_lib.KID_CHAIN)
        total = sum(ctx.timing_read(k)[1] for k in range(8))
        print(f"| {name} | {chains} | {eager:.0f} | {graph:.0f} | {ms / cnt * 1e3 if cnt else 0:.1f} x {cnt / 100:.0f} | {total / 100:.0f} |", flush=True)
        ctx.close()
