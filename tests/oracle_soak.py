#!/usr/bin/env python3
"""The HIP path against the CPU oracle over RANDOM CONFIGURATIONS, for a time budget (not a test: run by hand on a GPU box; the
suite's parity tests walk the configurations somebody thought of).  Lives under tests/ because it uses the oracle as its checker.

    python tests/oracle_soak.py --minutes 8 --seed0 0 > gpurun_out/oracle_soak.txt

Per case, drawn from one seeded generator: a system (one of 200 ragged random topologies of tests/helpers.py::random_topology --
with or without random constraint clusters --, or one of the synthetic boxes: mixed, nacl, ionic liquid with and without its
constraints, water plain and rigid, polymer in water, 6 / 12 / 32 temperature groups), the mode (TGNH / dualNH), the precision
(double / mixed), one of the six flag sets of the call-sequence tests with or without forced wave tiles, numNHChains in 1-6, 10,
16, useDrudeNHChains, useCOMTempGroup, the hard wall on or off, a CMMotionRemover in the System, the step size (1 / 0.5 fs), the
sub-steps (20 / 5 / 1), temperatures and coupling times.  The integrator's temperature groups are handed to the library in BOTH
modes (dualNH has to ignore them: the oracle's dualNH mode gets none).  Then `--steps` (30) steps on both sides with the harness
force -- through the split entry points and the harness' constraint call-outs when the system has constraints -- and positions,
velocities (north_star's 1e-6; the worst seen is printed) and the thermostat variables (rtol 1e-6 as in test_100_step_parity)
are compared.  A refused configuration (TGNH_ERR_UNSUPPORTED at create: wave tiles for a long molecule, a pair across a tile cut
...) is counted as skipped; anything else that raises or disagrees is a FAIL line with the case's seed."""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from oracle.binding import OracleError  # noqa: E402
from helpers import (make_oracle, oracle_run, rel_err, to_internal, random_topology, random_clusters,  # noqa: E402
                     drudes_at_the_end, onion, far_pairs, interleaved)
from openmm_drudenose_amd import synth, HipContext, _lib  # noqa: E402
from openmm_drudenose_amd.drudetgnhplugin import (DrudeTGNHIntegrator, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_WAVE_TILES,  # noqa: E402
                                                   FLAG_TRUST_STATE_CHANGED, FLAG_GATHER, TgnhError)

BOXES = {
    "mixed-200-15": lambda: synth.mixed(200, 15),
    "nacl": synth.nacl,
    "ionic-40": lambda: synth.ionic_liquid(40),
    "ionic-40-shake": lambda: synth.ionic_liquid(40, constrained=True),
    "water-343": lambda: synth.water_box(343),
    "water-rigid-216": lambda: synth.water_box(216, rigid=True),
    "polymer-600+200": lambda: synth.polymer_in_water(600, 200),
    "groups-6": lambda: synth.many_groups(200, 15, 6),
    "groups-12": lambda: synth.many_groups(200, 15, 12),
    "groups-32": lambda: synth.many_groups(200, 15, 32),
    # what the tiles cannot hold: the gather path (tgnh_gather.hip)
    "groups-33": lambda: synth.many_groups(200, 15, 33),
    "groups-100": lambda: synth.many_groups(300, 15, 100),
    "onion-700": onion,
    "far-pairs": far_pairs,
    "interleaved-60": interleaved,
    "water-8000": lambda: synth.water_box(8000),                    # 40 000 slots: every work-group of the one-launch step holds several tiles
    "mixed-6000-400": lambda: synth.mixed(6000, 400),               # 48 000 slots, four groups
}
REFUSED, GATHER = {}, {}                                           # refusal message -> cases ; gather-path reason -> cases
SINGLE = False                                                     # --single: every case in single precision, at GATES["single"]
# positions / velocities (relative, max norm), thermostat rtol, kinetic-energy query.  Single precision is not a parity gate of
# the suite (measured and reported); here it is run for what a loose gate still catches -- NaN, a wrong launch, a status bit
GATES = {"double": (1e-6, 1e-6, 1e-6, 1e-7), "mixed": (1e-6, 1e-6, 1e-6, 1e-7), "single": (1e-4, 2e-2, 5e-2, 1e-3)}
FLAG_SETS = [0, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP, FLAG_TRUST_STATE_CHANGED,
             FLAG_TRUST_STATE_CHANGED | FLAG_RESIDENT_STEP]


def ragged(k, clusters):
    mass, pd, pp, resid, group, ngroups, cons, sizes, first, rng = random_topology(k)
    pos = rng.uniform(0.0, 3.0, (len(mass), 3))
    if clusters:
        ca, cd = random_clusters(rng, mass, pd, pp, sizes, first, pos)
    s, g, ng = synth._finish(mass, np.array(pd, np.int32), np.array(pp, np.int32), resid, pos, group, ngroups, rng, 300.0, 1.0, f"ragged{k}")
    if clusters and len(ca):
        s.set_clusters(ca, cd)
    return s, g, ng


def one_case(rng, nsteps, info):
    u = int(rng.integers(0, 10))
    if u < 4:
        k, clusters = int(rng.integers(0, 200)), bool(rng.integers(0, 3) == 0)
        name, (s, g, ng) = f"ragged-{k}{'-clusters' if clusters else ''}", ragged(k, clusters)
    else:
        name = list(BOXES)[int(rng.integers(0, len(BOXES)))]
        s, g, ng = BOXES[name]()
    mode = "dualNH" if rng.integers(0, 10) < 3 else "TGNH"
    precision = "mixed" if rng.integers(0, 10) < 4 else "double"
    if SINGLE:
        precision = "single"
    flags = FLAG_SETS[int(rng.integers(0, len(FLAG_SETS)))] | (FLAG_WAVE_TILES if rng.integers(0, 2) else 0)
    chains = int(rng.choice([1, 1, 2, 3, 4, 5, 6, 10, 16]))
    drude_chains, com = bool(rng.integers(0, 4)), bool(rng.integers(0, 4))
    hardwall = 0.0 if rng.integers(0, 4) == 0 else 0.02
    s.has_cm_motion_remover = bool(rng.integers(0, 4) == 0)
    dt = float(rng.choice([0.001, 0.0005]))
    sub = int(rng.choice([20, 20, 5, 1]))
    temp, tau, dtemp, dtau = [(300.0, 0.1, 1.0, 0.005), (350.0, 0.05, 5.0, 0.01), (280.0, 0.2, 1.0, 0.1)][int(rng.integers(0, 3))]
    what = (f"system={name} mode={mode} precision={precision} flags={flags} chains={chains} drude_chains={drude_chains} com={com} "
            f"hardwall={hardwall} cmm={s.has_cm_motion_remover} dt={dt} substeps={sub} T={temp}/{dtemp} tau={tau}/{dtau}")
    info["what"] = what
    it = DrudeTGNHIntegrator(temp, tau, dtemp, dtau, dt, sub, chains, drude_chains, com)
    it.setMaxDrudeDistance(hardwall)
    it.setConstraintTolerance(1e-10)
    for _ in range(ng):
        it.addTempGroup()
    it._particleTempGroup = np.ascontiguousarray(g, np.int32)         # handed over in both modes
    try:
        ctx = HipContext(s, it, mode=mode, precision=precision, flags=flags)
    except TgnhError as e:
        if e.status == _lib.ERR_UNSUPPORTED:
            REFUSED[str(e)[:90]] = REFUSED.get(str(e)[:90], 0) + 1
            return "skip", what + f"  ({str(e)[:120]})", 0.0, 0.0
        raise
    if ctx.step_path()[0] == "gather":
        GATHER[ctx.step_path()[1]] = GATHER.get(ctx.step_path()[1], 0) + 1
    try:
        og, ong = (g, ng) if mode == "TGNH" else (np.zeros_like(g), 1)
        o = make_oracle(s, og, ong, mode, it)
        pos_o, vel_o, x0 = s.positions.copy(), s.velocities.copy(), ctx.sites()
        gate_p, gate_v, gate_t, gate_q = GATES[precision]
        query = not ctx.constrained        # (A12 with constraints needs the host's projection of the shifted velocities: include/drude_tgnh.h)
        if query:                          # the kinetic-energy query before the first step (computed: ke_sum_valid false) ...
            f = o.harness_force(pos_o, x0, synth.K_DRUDE, synth.K_TETHER)
            ke_h, ke_o = ctx.kinetic_energy(), o.kinetic_energy_query(vel_o, f, False)
            assert abs(ke_h - ke_o) <= gate_q * abs(ke_o), ("kinetic energy query at the start", ke_h, ke_o)
        if ctx.constrained:
            f = o.harness_force(pos_o, x0, synth.K_DRUDE, synth.K_TETHER)
            o.run_harness_constrained(pos_o, vel_o, f, x0, synth.K_DRUDE, synth.K_TETHER, 1e-10, nsteps)
        else:
            pos_o, vel_o = oracle_run(o, s, nsteps, x0=x0)
        ctx.step(nsteps)
        if query:                          # ... and after the last (TGNH: the cached bins of the last thermostat half; dualNH: shifted by half a kick)
            f = o.harness_force(pos_o, x0, synth.K_DRUDE, synth.K_TETHER)
            ke_h, ke_o = ctx.kinetic_energy(), o.kinetic_energy_query(vel_o, f, True)
            assert abs(ke_h - ke_o) <= gate_q * abs(ke_o), ("kinetic energy query after the steps", ke_h, ke_o)
        status = ctx.check()
        ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
        assert status == 0, ("status word", status)
        assert ep <= gate_p and ev <= gate_v, ("positions / velocities", ep, ev)
        if flags & FLAG_DEFER_SCALE and ctx.pending_state() & 0x8:
            # a rescale is owed: the deferred structure has run the NEXT step's first thermostat half already (its rescale rides
            # on that step's first launch) -- the oracle takes that half step too, on a copy of its velocities
            o.propagate_nhc(vel_o.copy())
        for which in (0, 1):
            a, b = ctx.thermostat_state(which), o.chain(which)
            assert np.allclose(a, b, rtol=gate_t, atol=gate_t * 1e-3 * max(1.0, np.abs(b).max())), ("thermostat", which, float(np.abs(a - b).max()))
        return "ok", what, ep, ev
    finally:
        ctx.close()


def sharded_case(rng, nsteps, info):
    """The particle-sharded path with the 'ranks' as two handles on this one GPU, each on its own stream, the kinetic
    energies exchanged through the library's mailboxes (as test_particle_sharded_mailbox_exchange_on_one_gpu): the shards'
    positions and velocities, put together, against the oracle's run of the WHOLE system; thermostats bit-identical on the ranks."""
    import torch
    from openmm_drudenose_amd.system import shard_bounds
    u = int(rng.integers(0, 10))
    if u < 4:
        k = int(rng.integers(0, 200))
        name, (s, g, ng) = f"ragged-{k}", ragged(k, False)
    else:
        name = [n for n in BOXES if "shake" not in n and "rigid" not in n][int(rng.integers(0, len(BOXES) - 2))]
        s, g, ng = BOXES[name]()
    mode = "dualNH" if rng.integers(0, 10) < 3 else "TGNH"
    precision = "mixed" if rng.integers(0, 10) < 4 else "double"
    flags = [0, FLAG_DEFER_SCALE, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP, FLAG_TRUST_STATE_CHANGED][int(rng.integers(0, 4))]
    flags |= FLAG_WAVE_TILES if rng.integers(0, 2) else 0
    chains = int(rng.choice([1, 1, 2, 3, 4, 6, 10]))
    drude_chains, com = bool(rng.integers(0, 4)), bool(rng.integers(0, 4))
    hardwall = 0.0 if rng.integers(0, 4) == 0 else 0.02
    s.has_cm_motion_remover = bool(rng.integers(0, 4) == 0)
    dt, sub = float(rng.choice([0.001, 0.0005])), int(rng.choice([20, 20, 5, 1]))
    nranks = 2
    what = (f"sharded x{nranks} system={name} mode={mode} precision={precision} flags={flags} chains={chains} drude_chains={drude_chains} "
            f"com={com} hardwall={hardwall} cmm={s.has_cm_motion_remover} dt={dt} substeps={sub}")
    info["what"] = what

    def integrator(group):
        it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, dt, sub, chains, drude_chains, com)
        it.setMaxDrudeDistance(hardwall)
        for _ in range(ng):
            it.addTempGroup()
        it._particleTempGroup = np.ascontiguousarray(group, np.int32)
        return it
    b = shard_bounds(s, nranks)
    if min(np.diff(b)) == 0:
        return "skip", what + "  (an empty shard)", 0.0, 0.0
    parts = []
    try:
        try:
            for r in range(nranks):
                loc = s.slice_molecules(b[r], b[r + 1])
                loc.has_cm_motion_remover = s.has_cm_motion_remover
                parts.append(HipContext(loc, integrator(g[b[r]:b[r + 1]]), mode=mode, precision=precision, flags=flags))
            # Every rank of a sharded run must step in the same pass structure (the same number of exchanges per step), and the
            # gather path ignores the flags that change it: if any shard's topology took it, all shards do (include/drude_tgnh.h)
            if len({c.step_path()[0] for c in parts}) > 1:
                for r in range(nranks):
                    if parts[r].step_path()[0] != "gather":
                        parts[r].close()
                        loc = s.slice_molecules(b[r], b[r + 1])
                        loc.has_cm_motion_remover = s.has_cm_motion_remover
                        parts[r] = HipContext(loc, integrator(g[b[r]:b[r + 1]]), mode=mode, precision=precision, flags=flags | FLAG_GATHER)
            for c in parts:
                if flags & FLAG_RESIDENT_STEP:
                    c.set_resident_share(nranks)
        except TgnhError as e:
            if e.status == _lib.ERR_UNSUPPORTED:
                REFUSED[str(e)[:90]] = REFUSED.get(str(e)[:90], 0) + 1
                return "skip", what + f"  ({str(e)[:120]})", 0.0, 0.0
            raise
        streams = [torch.cuda.Stream(priority=-r) for r in range(nranks)]      # two different hardware queues
        total = sum(c.local_dof_terms() for c in parts)
        try:
            boxes = [c.exchange_create(nranks, r)[1] for r, c in enumerate(parts)]
        except TgnhError as e:                              # more than 34 thermostats: the mailboxes do not hold them (an all-reduce hook does:
            if "too many thermostats" in str(e):            # tests/test_gather_gpu.py::test_sharded_gather_path_with_an_allreduce_hook)
                return "skip", what + "  (mailbox: <= 34 thermostats)", 0.0, 0.0
            raise
        for c in parts:
            c.set_global_dof_terms(total)
            c.exchange_attach_pointers(boxes)
        torch.cuda.synchronize()
        x0 = np.concatenate([c.sites() for c in parts])
        og, ong = (g, ng) if mode == "TGNH" else (np.zeros_like(g), 1)
        o = make_oracle(s, og, ong, mode, integrator(g))
        pos_o, vel_o = oracle_run(o, s, nsteps, x0=x0)
        for _ in range(nsteps):
            for c, st in zip(parts, streams):
                with torch.cuda.stream(st):
                    c.step_begin(); c.compute_forces(); c.step_end()
        for c, st in zip(parts, streams):                            # settle every rank's pending half before any rank waits on a query
            with torch.cuda.stream(st):
                assert c.lib.tgnh_flush(c.h, c._stream()) == 0
        torch.cuda.synchronize()
        for c in parts:
            assert c.check() == 0, ("status word", c.status_flags())
        ep = rel_err(np.concatenate([c.getPositions() for c in parts]), pos_o)
        ev = rel_err(np.concatenate([c.getVelocities() for c in parts]), vel_o)
        assert ep <= 1e-6 and ev <= 1e-6, ("positions / velocities", ep, ev)
        for which in (0, 1):
            a = parts[0].thermostat_state(which)
            for c in parts[1:]:
                assert np.array_equal(a, c.thermostat_state(which)), ("thermostats differ between the ranks", which)
        return "ok", what, ep, ev
    finally:
        for c in parts:
            try:
                c.exchange_detach()
            except Exception:  # noqa: BLE001
                pass
            c.close()


def checkpoint_case(rng, nsteps, info):
    """State persistence over random configurations: positions, velocities, the thermostat variables and the clock
    (serialization.save_thermostat) are taken between two steps and put into a FRESH handle; `nsteps` further steps on both must
    agree bit for bit -- positions, velocities, thermostats.  Flag sets without DEFER_SCALE (a deferred handle holds the coming
    step's first thermostat half between steps and says so: serialization.py) and without TRUST_STATE_CHANGED (the restored
    handle sums its kinetic energies again where the running one carried them: equal to rounding only)."""
    from openmm_drudenose_amd.serialization import save_thermostat, load_thermostat
    u = int(rng.integers(0, 10))
    if u < 4:
        k = int(rng.integers(0, 200))
        name, mk = f"ragged-{k}", (lambda: ragged(k, False))
    else:
        name = [n for n in BOXES if "shake" not in n and "rigid" not in n][int(rng.integers(0, len(BOXES) - 2))]
        mk = BOXES[name]
    mode = "dualNH" if rng.integers(0, 10) < 3 else "TGNH"
    precision = "mixed" if rng.integers(0, 10) < 4 else "double"
    flags = int(rng.choice([0, FLAG_RESIDENT_STEP])) | (FLAG_WAVE_TILES if rng.integers(0, 2) else 0)
    chains = int(rng.choice([1, 1, 2, 3, 4, 6, 10]))
    drude_chains, com = bool(rng.integers(0, 4)), bool(rng.integers(0, 4))
    hardwall = 0.0 if rng.integers(0, 4) == 0 else 0.02
    cmm = bool(rng.integers(0, 4) == 0)
    dt, sub = float(rng.choice([0.001, 0.0005])), int(rng.choice([20, 20, 5, 1]))
    before = int(rng.integers(1, 12))                                # steps before the checkpoint (odd and even: the sweep direction)
    what = (f"checkpoint after {before} steps system={name} mode={mode} precision={precision} flags={flags} chains={chains} "
            f"drude_chains={drude_chains} com={com} hardwall={hardwall} cmm={cmm} dt={dt} substeps={sub}")
    info["what"] = what

    def context():
        s, g, ng = mk()
        s.has_cm_motion_remover = cmm
        it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, dt, sub, chains, drude_chains, com)
        it.setMaxDrudeDistance(hardwall)
        for _ in range(ng):
            it.addTempGroup()
        it._particleTempGroup = np.ascontiguousarray(g, np.int32)
        return HipContext(s, it, mode=mode, precision=precision, flags=flags)
    try:
        a = context()
    except TgnhError as e:
        if e.status == _lib.ERR_UNSUPPORTED:
            REFUSED[str(e)[:90]] = REFUSED.get(str(e)[:90], 0) + 1
            return "skip", what + f"  ({str(e)[:120]})", 0.0, 0.0
        raise
    b = None
    try:
        a.step(before)
        state, pos, vel = save_thermostat(a), a.getPositions(), a.getVelocities()
        a.step(nsteps)
        b = context()
        b.setPositions(pos); b.setVelocities(vel); b.compute_forces()
        load_thermostat(b, state)
        b.step(nsteps)
        assert a.check() == 0 and b.check() == 0 and a.time() == b.time()
        assert np.array_equal(a.getPositions(), b.getPositions()), "positions differ after the restore"
        assert np.array_equal(a.getVelocities(), b.getVelocities()), "velocities differ after the restore"
        for which in range(4):
            assert np.array_equal(a.thermostat_state(which), b.thermostat_state(which)), ("thermostat differs after the restore", which)
        return "ok", what, 0.0, 0.0
    finally:
        a.close()
        if b is not None:
            b.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", action="store_true", help="checkpoint / restore into a fresh handle: bit for bit")
    ap.add_argument("--sharded", action="store_true", help="two ranks on this GPU with the mailbox exchange, against the oracle's run of the whole system")
    ap.add_argument("--minutes", type=float, default=8.0)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--only", default="", help="comma-separated case seeds")
    ap.add_argument("--single", action="store_true", help="single precision throughout, loose gates (see GATES)")
    a = ap.parse_args()
    global SINGLE
    SINGLE = a.single
    only = [int(x) for x in a.only.split(",") if x]
    t_end, n, count, worst = time.time() + 60.0 * a.minutes, 0, {"ok": 0, "skip": 0, "FAIL": 0}, {"double": [0.0, 0.0], "mixed": [0.0, 0.0], "single": [0.0, 0.0]}
    while time.time() < t_end and (not only or n < len(only)):
        seed = only[n] if only else a.seed0 + n
        n += 1
        rng = np.random.default_rng(seed)
        t0 = time.time()
        info = {"what": ""}
        try:
            kind, what, ep, ev = (sharded_case if a.sharded else checkpoint_case if a.checkpoint else one_case)(rng, a.steps, info)
            if kind == "ok":
                w = worst[what.split("precision=")[1].split()[0]]
                w[0], w[1] = max(w[0], ep), max(w[1], ev)
            print(f"{kind:5s} seed={seed} {what}  pos {ep:.1e} vel {ev:.1e}  {time.time() - t0:.1f}s", flush=True)
        except OracleError as e:                             # random clusters the oracle's own SHAKE gives up on: no verdict
            kind = "skip"
            print(f"skip  seed={seed} {info['what']}  (oracle: {e})", flush=True)
        except Exception as e:  # noqa: BLE001  (a soak: log and go on)
            kind = "FAIL"
            print(f"FAIL  seed={seed} {info['what']}  {type(e).__name__}: {str(e)[:600]}", flush=True)
            traceback.print_exc(limit=3, file=sys.stdout)
        count[kind] += 1
    refused = sum(REFUSED.values())
    print(f"{n} cases: {count['ok']} ok, {refused} refused as unsupported at create, {count['skip'] - refused} without a verdict (the oracle's own SHAKE "
          f"gave up on a random cluster / an empty shard / a mailbox asked for more than 34 thermostats), {count['FAIL']} failed; worst pos / vel error: "
          f"double {worst['double'][0]:.1e} / {worst['double'][1]:.1e}, mixed {worst['mixed'][0]:.1e} / {worst['mixed'][1]:.1e}, "
          f"single {worst['single'][0]:.1e} / {worst['single'][1]:.1e}", flush=True)
    # every refusal by name: only what the reference itself cannot run may appear here (a massless pair member, Ref :132; a molecule
    # without mass under the COM group, K :86-104; dualNH without a pair, Ref :181; DEFER_SCALE asked for a topology it cannot hold)
    for msg, k in sorted(REFUSED.items(), key=lambda kv: -kv[1]):
        print(f"  refused x{k}: {msg}", flush=True)
    for why, k in sorted(GATHER.items(), key=lambda kv: -kv[1]):
        print(f"  on the gather path x{k}: {why}", flush=True)
    return 1 if count["FAIL"] else 0


if __name__ == "__main__":
    sys.exit(main())
