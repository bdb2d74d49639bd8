#!/usr/bin/env python3
"""The Drude thermostat of bench.py's synthetic water box by size: its kinetic energy over its target and its etaDot after the
same number of steps -- intensive quantities; why does the chain's exponent argument grow with the box?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from openmm_drudenose_amd import synth
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext

for mol in [int(x) for x in sys.argv[1:]] or [6400, 125000, 1000000]:
    s, g, ng = synth.water_box(mol)
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
    it.setMaxDrudeDistance(0.02)
    ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=0)
    dof = ctx.dof() if hasattr(ctx, "dof") else None
    for steps in (1, 10, 50, 100):
        ctx.step(steps)
        ke = ctx.last_kinetic_energies()
        ed = ctx.thermostat_state(1)
        print(mol, "after +%d steps" % steps, "KE bins", ke, "etaDot", ed, "scale", ctx.last_scale_factors(), flush=True)
    print("dof", dof)
    ctx.close()
