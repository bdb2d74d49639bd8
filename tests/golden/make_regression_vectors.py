#!/usr/bin/env python3
"""Generates tests/golden/oracle_regression.npz from THIS repository's oracle (oracle/tgnh_oracle.c).

These are regression vectors of the oracle itself, NOT reference outputs: the reference cannot be run here
(DESIGN.md section 6).  They freeze what the oracle computes today, so that a later edit of the oracle, of the synthetic
system builders or of the harness force that changes any result is caught by tests/test_oracle.py, and they travel to
the GPU box as data.  Inputs are regenerated from seeds (openmm_drudenose_amd.synth, seed 20191024); stored are the
per-half-step kinetic energies and scale factors, the final thermostat variables and the final coordinates /
velocities of the first 64 particles.

    python tests/golden/make_regression_vectors.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

from openmm_drudenose_amd import synth, DrudeTGNHIntegrator          # noqa: E402
from oracle import Oracle, MODE_DUALNH, MODE_TGNH                    # noqa: E402
from helpers import oracle_run                                       # noqa: E402

CASES = {
    # name: (builder, mode, chains, useDrudeNHChains, useCOM, hard wall, steps)
    "nacl_tgnh": (synth.nacl, MODE_TGNH, 1, True, True, 0.02, 40),
    "nacl_dualnh_coupled": (synth.nacl, MODE_DUALNH, 3, False, True, 0.02, 40),
    "mixed_tgnh_4groups": (lambda: synth.mixed(60, 5), MODE_TGNH, 2, True, True, 0.0, 40),
    "pnm_dualnh": (synth.pair_normal_massless, MODE_DUALNH, 2, True, True, 0.0, 40),
}


def run(name):
    build, mode, chains, drude_chains, com, hw, steps = CASES[name]
    s, g, ng = build()
    if mode == MODE_DUALNH:
        g, ng = np.zeros_like(g), 1
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, drude_chains, com)
    it.setMaxDrudeDistance(hw)
    o = Oracle.from_integrator(s, it, g, ng, mode)
    pos, vel, ke, sc = oracle_run(o, s, steps, record=True)
    return {"ke": ke, "scale": sc, "eta": o.chain(0), "etaDot": o.chain(1), "pos64": pos[:64], "vel64": vel[:64]}


if __name__ == "__main__":
    out = {}
    for name in CASES:
        for k, v in run(name).items():
            out[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "oracle_regression.npz"), **out)
    print("wrote", len(out), "arrays")
