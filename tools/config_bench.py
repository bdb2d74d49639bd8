#!/usr/bin/env python3
"""GPU-side: steps/s of the BASELINE.json configs on ONE MI355X (C4/C5 are multi-GPU configs; their 1-GPU rate
is what can be measured on a 1-GPU box).  Prints a markdown table for BASELINE.md section 5."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext, _lib
from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP
CFG = [("C1 nacl", synth.nacl, 0.02), ("C2 SWM4 32k", lambda: synth.water_box(6400), 0.0),
       ("C3 ionic liquid 100k", lambda: synth.ionic_liquid(2222), 0.0),
       ("C3 as worded: + SHAKE on its 33 330 X-H bonds (split path, harness solver)", lambda: synth.ionic_liquid(2222, constrained=True), 0.0),
       ("C4 mixed 500k + hard wall", lambda: synth.mixed(60000, 4444), 0.02),
       ("C5 SWM4 2M", lambda: synth.water_box(400000), 0.0), ("metric SWM4 1M pairs", lambda: synth.water_box(1000000), 0.02)]
print("| config | N slots | pairs | groups | precision | variant | numNHChains | steps/s eager | steps/s hipGraph | B_step model | GB/s vs model |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for name, build, hw in CFG:
    s, g, ng = build()
    runs = [("mixed", "plain", 1), ("mixed", "defer", 1), ("mixed", "resident", 1), ("single", "resident" if s.num_particles < 2000000 else "defer", 1)]
    constrained = len(s.constraints) > 0
    if constrained:                              # the constraint call-outs need stored velocities: the reference's structure, or its halves as one launch each
        runs = [("mixed", "plain", 1), ("mixed", "plain-resident", 1)]
    if name.startswith("C2") or name.startswith("metric"):       # longer chains: 3 links (in-kernel below 2 M slots) and the reference test's own 10
        runs += [("mixed", "resident", 3), ("mixed", "defer", 3), ("mixed", "resident", 10), ("mixed", "defer", 10)]
    elif not constrained and (name.startswith("C3") or name.startswith("C4")):     # ... three links inside the launches with 2 / 4 temperature groups
        runs += [("mixed", "resident", 3)]
    for prec, var, chains in runs:
        it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, True, True)
        it.setMaxDrudeDistance(hw)
        for _ in range(ng): it.addTempGroup()
        it._particleTempGroup = g.astype("int32")
        flags = {"plain": 0, "plain-resident": FLAG_RESIDENT_STEP, "defer": FLAG_DEFER_SCALE, "resident": FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP}[var]
        ctx = HipContext(s, it, mode="TGNH", precision=prec, flags=flags)   # plain: the reference's pass structure, what the OpenMM glue runs
        ctx.step(50); torch.cuda.synchronize()
        n, eager, graph = 500, 0.0, 0.0
        for _ in range(3):                       # best of three: a row now and then catches a host stall (C2 single: 13.9 k once, 71 k alone)
            t0 = time.perf_counter(); ctx.step(n); torch.cuda.synchronize(); eager = max(eager, n / (time.perf_counter() - t0))
            rep = ctx.capture_steps(10); rep(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n // 10): rep()
            torch.cuda.synchronize(); graph = max(graph, n / (time.perf_counter() - t0))
        V = 16 if prec == "single" else 32
        b = s.num_particles * (7 * V + 48 + 2 * V)
        print(f"| {name} | {s.num_particles} | {s.num_pairs} | {ng} | {prec} | {var} | {chains} | {eager:.0f} | {graph:.0f} | {b/1e6:.1f} MB | {b*max(eager,graph)/1e9:.0f} |", flush=True)
        ctx.close()
