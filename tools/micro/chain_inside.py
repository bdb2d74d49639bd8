#!/usr/bin/env python3
"""chain_kernel from the inside (tuning build with -DTGNH_TRACE -DTGNH_TUNING, TGNH_INLINE_MULTI_MAX=0 so that chains of 3 links
take their own launch): its own wall clock and cycle counter at entry, after the prologue and at exit, next to the HIP-event time
of the launch, at several system sizes.
   python tools/build_variant.py build_variants/lib_trace.so -DTGNH_TRACE -DTGNH_TUNING
   TGNH_LIB=build_variants/lib_trace.so TGNH_INLINE_MULTI_MAX=0 python tools/micro/chain_inside.py [molecules ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from openmm_drudenose_amd import synth, _lib
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext, FLAG_DEFER_SCALE
import torch

for mol in [int(x) for x in sys.argv[1:]] or [6400, 25000, 125000, 1000000]:
    s, g, ng = synth.water_box(mol)
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 3, True, True)
    it.setMaxDrudeDistance(0.02)
    ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE)
    lib = _lib.load()
    ctx.step(60)
    torch.cuda.synchronize()
    inside, pro, mhz = [], [], []
    for rep in range(20):
        ctx.step(1)
        torch.cuda.synchronize()
        buf = np.zeros(8, np.uint64)
        assert lib.tgnh_debug_read_chain_trace(buf.ctypes.data_as(C.c_void_p)) == 0
        t = buf.astype(np.int64)
        inside.append((t[4] - t[0]) / 100.0); pro.append((t[2] - t[0]) / 100.0)
        mhz.append((t[5] - t[1]) / max(1, (t[4] - t[0])) * 100.0)
    ctx.step(20)
    torch.cuda.synchronize()
    buf = np.zeros(8, np.uint64)
    assert lib.tgnh_debug_read_chain_trace(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.astype(np.int64)
    print(f"         last launch of 20 steps enqueued back to back: inside {(t[4] - t[0]) / 100.0:6.2f} us")
    dbg = np.zeros(4)
    assert lib.tgnh_debug_read_chain_dbg(dbg.ctypes.data_as(C.c_void_p)) == 0
    print(f"         largest exponent argument: real thermostats {dbg[0]:.4g}, Drude {dbg[1]:.4g} (fast path holds below 0.0156); range left {int(dbg[2])} / {int(dbg[3])} times")
    ctx.timing(True)
    ctx.step(50)
    torch.cuda.synchronize()
    ms, n = ctx.timing_read(_lib.KID_CHAIN)
    ctx.timing(False)
    tm = f"{ms * 1e3 / max(n, 1):.2f} us x {n}"
    print(f"{mol:8d} molecules: chain_kernel inside {np.median(inside):6.2f} us (prologue {np.median(pro):5.2f}), its cycle counter / wall clock = {np.median(mhz):6.1f} MHz; "
          f"HIP events around the launch: {tm}")
    ctx.close()
