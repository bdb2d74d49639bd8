#!/usr/bin/env python3
"""Phase timeline of the tile kernels (tuning build with -DTGNH_TRACE):
   TGNH_LIB=build_variants/lib_trace.so python tools/trace_probe.py [molecules ...]
Slots: 0 entry, 1 first loads issued, 2 scale factors ready (chain prologue), per tile i: 3+4i data arrived,
4+4i rescaled, 5+4i before stores, 6+4i tile done; 15 kernel exit.  Times in us from the first work-group's entry."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openmm_drudenose_amd import synth, _lib
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext, FLAG_DEFER_SCALE

NAMES = {0: "entry", 1: "loads issued", 2: "scale ready", 15: "exit"}
for i in range(3):
    NAMES.update({3 + 4 * i: f"t{i} data in", 4 + 4 * i: f"t{i} rescaled", 5 + 4 * i: f"t{i} pre-store", 6 + 4 * i: f"t{i} done"})
NAMES.update({13: "chain start*", 14: "chain end*"})      # in-kernel chain (shares slots with a third tile's)


def show(lib, label, torch):
    torch.cuda.synchronize()
    buf = np.zeros(2048 * 16, np.uint64)
    assert lib.tgnh_debug_read_trace(buf.ctypes.data_as(C.c_void_p)) == 0
    tr = buf.reshape(2048, 16).astype(np.int64)
    live = tr[:, 0] > 0
    tr = tr[live]
    base = tr[:, 0].min()
    print(f"--- {label}: {live.sum()} work-groups, span {(tr[:, 15].max() - base) / 100:.2f} us")
    hw0, hw1 = tr[:, 11], tr[:, 12]
    if hw0.any():
        def dec(h): return dict(xcc=(h >> 32) & 15, se=(h >> 13) & 7, sh=(h >> 12) & 1, cu=(h >> 8) & 15, simd=(h >> 4) & 3, wave=h & 15)
        d0, d1 = dec(hw0), dec(hw1)
        cuid = d0["xcc"] * 1000 + d0["se"] * 100 + d0["sh"] * 50 + d0["cu"]
        ids = np.flatnonzero(live)
        print("  wave0 simd histogram", np.bincount(d0["simd"], minlength=4), " wave1 simd", np.bincount(d1["simd"], minlength=4))
        print("  distinct CUs", len(set(cuid.tolist())), " work-groups per CU histogram", np.bincount(np.bincount(np.unique(cuid, return_inverse=True)[1])))
        for lo in range(0, len(ids), 256):
            sel = (ids >= lo) & (ids < lo + 256)
            if sel.sum() == 0: continue
            ready = (tr[sel, 2] - base) / 100.0
            print(f"  blockIdx {lo:4d}..{lo + 255:4d}: scale ready p50 {np.median(ready):6.2f} max {ready.max():6.2f}; xcc of first 16: {d0['xcc'][sel][:16].tolist()}; cu of first 16: {(cuid % 1000)[sel][:16].tolist()}")
    for s in range(16):
        if s in (11, 12) and hw0.any(): continue
        col = tr[:, s]
        m = col >= base
        if m.sum() == 0 or s not in NAMES: continue
        us = (col[m] - base) / 100.0
        print(f"  {NAMES[s]:14s} n={m.sum():5d}  min {us.min():6.2f}  p50 {np.median(us):6.2f}  p90 {np.percentile(us, 90):6.2f}  max {us.max():6.2f}")
    assert lib.tgnh_debug_clear_trace() == 0


for mol in [int(x) for x in sys.argv[1:]] or [125000, 250000]:
    s, g, ng = synth.water_box(mol)
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, int(os.environ.get("TGNH_PROBE_S", "20")), 1, True, True)
    it.setMaxDrudeDistance(0.02)
    ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE)
    lib = _lib.load()
    it.step(50)
    import torch
    torch.cuda.synchronize()
    # zero the trace, then one launch of each kernel
    assert lib.tgnh_debug_clear_trace() == 0
    ctx.step_begin(); show(lib, f"{mol} molecules: scale+kick+drift", torch)
    ctx.compute_forces(); torch.cuda.synchronize()
    ctx.step_end(); show(lib, f"{mol} molecules: kick+KE", torch)
    for _ in range(3):                      # again, after the caches have seen the same sequence
        ctx.step_begin(); ctx.compute_forces(); ctx.step_end()
    torch.cuda.synchronize(); assert lib.tgnh_debug_clear_trace() == 0
    ctx.step_begin(); show(lib, f"{mol} molecules: scale+kick+drift (again)", torch)
    ctx.close()
