#!/usr/bin/env python3
"""Phase timeline of step_kernel (tuning build with -DTGNH_TRACE -DTGNH_TUNING):
   python tools/build_variant.py build_variants/lib_trace.so -DTGNH_TRACE -DTGNH_TUNING
   TGNH_LIB=build_variants/lib_trace.so python tools/step_trace.py [--precision single|mixed|double] [molecules ...]
Slots: 0 entry, 1 pass 1 done, 2 row handed in, 7 (work-group 0) rows collected, 8 sums received, 9 chain done,
15 exit (pass 2 done).  Times in us from the first work-group's entry."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openmm_drudenose_amd import synth, _lib
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP

NAMES = {0: "entry", 1: "pass 1 done", 2: "row handed in", 6: "poll starts*", 10: "rows all seen*", 7: "rows collected*", 13: "sums sent*", 14: "chain loop 1 in", 11: "chain loop 2 out", 12: "chain out", 8: "sums received", 9: "chain done",
         3: "p1 t0 data in", 4: "p1 t0 prepared", 5: "p2 t0 pre-store", 15: "exit"}
argv = sys.argv[1:]
precision = "mixed"
if argv and argv[0] == "--precision":
    precision, argv = argv[1], argv[2:]
for mol in [int(x) for x in argv] or [125000]:
    s, g, ng = synth.water_box(mol)
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
    it.setMaxDrudeDistance(0.02)
    ctx = HipContext(s, it, mode="TGNH", precision=precision, flags=FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)
    lib = _lib.load()
    it.step(50)
    import torch
    torch.cuda.synchronize()
    for rep in range(2):
        assert lib.tgnh_debug_clear_trace() == 0
        ctx.step_begin()
        torch.cuda.synchronize()
        buf = np.zeros(2048 * 16, np.uint64)
        assert lib.tgnh_debug_read_trace(buf.ctypes.data_as(C.c_void_p)) == 0
        tr = buf.reshape(2048, 16).astype(np.int64)
        live = tr[:, 0] > 0
        tr = tr[live]
        base = tr[:, 0].min()
        print(f"--- {mol} molecules, {precision}, {ctx.resident_kernel()}: {live.sum()} work-groups, span {(tr[:, 15].max() - base) / 100:.2f} us")
        for sl in (0, 3, 4, 1, 2, 6, 10, 7, 13, 8, 14, 11, 12, 9, 5, 15):
            col = tr[:, sl]
            m = col >= base
            if m.sum() == 0:
                continue
            us = (col[m] - base) / 100.0
            print(f"  {NAMES[sl]:16s} n={m.sum():5d}  min {us.min():6.2f}  p50 {np.median(us):6.2f}  p90 {np.percentile(us, 90):6.2f}  max {us.max():6.2f}")
        if os.environ.get("TGNH_TRACE_BY_XCD"):              # where the spread comes from: work-group b runs on XCD b mod 8, on compute unit (b div 8) mod 32 of it
            idx = np.flatnonzero(live)
            for sl in (1, 15):
                us = (tr[:, sl] - base) / 100.0
                print(f"  {NAMES[sl]:12s} by XCD (p50): " + " ".join(f"{np.median(us[idx % 8 == x]):6.2f}" for x in range(8)) +
                      "   | by slot on the CU (first / second work-group): " + " ".join(f"{np.median(us[(idx // 256) == k]):6.2f}" for k in range(2)))
        ctx.compute_forces(); ctx.step_end()
    ctx.close()
