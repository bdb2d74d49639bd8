import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext, _lib
from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP
s, g, ng = synth.water_box(6400)
for prec in ("single", "mixed", "single"):
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
    it.setMaxDrudeDistance(0.0)
    for _ in range(ng): it.addTempGroup()
    it._particleTempGroup = g.astype("int32")
    ctx = HipContext(s, it, mode="TGNH", precision=prec, flags=FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)
    ctx.step(50); torch.cuda.synchronize()
    for rep_i in range(3):
        n = 500
        t0 = time.perf_counter(); ctx.step(n); torch.cuda.synchronize(); eager = n / (time.perf_counter() - t0)
        rep = ctx.capture_steps(10); rep(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n // 10): rep()
        torch.cuda.synchronize(); graph = n / (time.perf_counter() - t0)
        print(prec, rep_i, f"eager {eager:.0f} graph {graph:.0f}", "check", ctx.check(), "pending", hex(ctx.pending_state()), flush=True)
    ctx.timing(True); ctx.step(100); torch.cuda.synchronize()
    print({_lib.KERNEL_NAMES[k]: ctx.timing_read(k) for k in range(8)})
    ctx.timing(False)
    ctx.close()
