#!/usr/bin/env python3
"""Why does `bench.py --steps 20 --warmup 5` (the driver's shape) read 3-9 % under the steady state, and more on some boxes than
on others?  Two candidates: the state (a fresh box is away from equilibrium: the chain's arguments are larger for the first ~60
steps -- round 4's finding, 1-2 %) and the device (clocks that have dropped while the host built the system for seconds and need
more than 6 ms of work to come back).  A fresh process per run (same state every time): the metric system, 5 warm-up steps, then
20 timed steps in 4 blocks of 5 -- as is, after 300 ms of device-to-device copies (work, not steps: the state does not move),
and after 100 extra steps (state and device both warm)."""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for rep in range(2):
        for mode in ("asis", "spin", "steps"):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True)
            print(mode, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:], flush=True)
    sys.exit(0)
import numpy as np, torch
from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext
from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP
mode = sys.argv[1]
s, g, ng = synth.water_box(1000000)
it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
it.setMaxDrudeDistance(0.02)
it.addTempGroup(); it._particleTempGroup = np.ascontiguousarray(g, np.int32)
ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)
torch.cuda.synchronize()
if mode == "spin":
    a = torch.empty(64 << 20, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        b.copy_(a); torch.cuda.synchronize()
elif mode == "steps":
    ctx.step(100)
ctx.step(5); torch.cuda.synchronize()
out = []
for blk in range(4):
    t0 = time.perf_counter(); ctx.step(5); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 5 * 1e6)
print(" ".join(f"{x:.1f}" for x in out), f"us per step in blocks of 5; 20 steps: {20e6 / (sum(out) * 5):.0f} steps/s")
