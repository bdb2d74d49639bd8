#!/usr/bin/env python3
"""Tile-size sweep at the metric size: tools/cap_sweep.py [molecules] -- steps/s and the rescale+kick+drift launch by
TGNH_TILE_CAP (slots per tile; tiles end at molecule boundaries, so 512 means 510 for 5-slot waters).  One process,
contexts built one after the other."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from openmm_drudenose_amd import synth, _lib
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext, FLAG_DEFER_SCALE

mol = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
caps = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512, 505, 500, 495, 490, 485, 480, 475, 470, 460, 450, 440, 430, 420, 410, 400, 380, 360, 340, 320]
system, group, ngroups = synth.water_box(mol)
for rep in range(2):
    for cap in caps:
        os.environ["TGNH_TILE_CAP"] = str(cap)
        it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
        it.setMaxDrudeDistance(0.02)
        ctx = HipContext(system, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE)
        ctx.step(60)
        torch.cuda.synchronize()
        ctx.timing(2 + 0)
        t0 = time.perf_counter()
        ctx.step(400)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ctx.timing(False)
        ms, n = ctx.timing_read(0)
        tiles = len(ctx.topology(7)) - 1
        print(f"cap {cap:4d} tiles {tiles:6d}: {400 / dt:8.1f} steps/s | scale+kick+drift {ms / n * 1e3:7.2f} us", flush=True)
        ctx.close()
