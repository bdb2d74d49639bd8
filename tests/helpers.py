"""Shared helpers for the parity tests (oracle = checker, HIP library = thing under test)."""
import numpy as np

from openmm_drudenose_amd import synth
from oracle import Oracle, MODE_DUALNH, MODE_TGNH

ONE_4PI_EPS0 = 138.935456          # OpenMM SimTKOpenMMRealType.h
MODES = {"dualNH": MODE_DUALNH, "TGNH": MODE_TGNH}


def rel_err(a, b):
    """max-norm relative error: max|a-b| / max|b|."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))


def make_oracle(system, group, ngroups, mode, integ):
    return Oracle.from_integrator(system, integ, group, ngroups, MODES[mode])


def oracle_run(o, system, nsteps, k_drude=synth.K_DRUDE, k_tether=synth.K_TETHER, record=False, x0=None):
    """Runs the oracle with the harness force; returns final (pos, vel) and optionally per-step KE/scale.
    x0 = tether sites; pass ctx.sites() so the oracle sees the sites exactly as the HIP harness stores them."""
    pos, vel = system.positions.copy(), system.velocities.copy()
    x0 = system.positions.copy() if x0 is None else np.ascontiguousarray(x0, np.float64)
    f = o.harness_force(pos, x0, k_drude, k_tether)
    if not record:
        o.run_harness(pos, vel, f, x0, k_drude, k_tether, nsteps)
        return pos, vel
    kes, scs = [], []
    for _ in range(nsteps):
        ke, sc = o.propagate_nhc(vel)
        kes.append(ke); scs.append(sc)
        o.half_kick(vel, f)
        o.drift(pos, vel)
        o.hardwall(pos, vel)
        f = o.harness_force(pos, x0, k_drude, k_tether)
        o.half_kick(vel, f)
        ke, sc = o.propagate_nhc(vel)
        kes.append(ke); scs.append(sc)
    return pos, vel, np.array(kes), np.array(scs)


def to_internal(x, mode):
    """Oracle thermostat vectors -> the library's NT layout (dualNH: [real, 0, drude])."""
    x = np.asarray(x)
    if mode == "dualNH":
        return np.array([x[0], 0.0, x[1]]) if x.ndim == 1 else np.stack([x[:, 0], 0 * x[:, 0], x[:, 1]], 1)
    return x
