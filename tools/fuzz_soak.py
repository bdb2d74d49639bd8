#!/usr/bin/env python3
"""A soak of tests/test_abi_sequences_gpu.py's random walks: more seeds, more systems, for a time budget (run once on a GPU box;
the suite itself keeps its 27 fixed walks).  Every walk drives a handle of a random (flags, exchange, chain length, tile kind)
over 500 random entry-point calls against a plain handle of the same tile kind.  A failing walk is logged with what reproduces
it (system, flags, exchange, chains, wave, seed) and the soak goes on.

The gates.  Chains of several links far from equilibrium -- the synthetic boxes -- are chaotic: on the CPU oracle ONE ulp in one
velocity component grows to 2e-4 in the velocities within 700 steps of six links, 2e-6 with four (mixed-60-6, water-400), and the
variants differ from the plain handle by such ulps (a kinetic energy carried as s^2 KE instead of summed again, a sum collected by
another launch).  The suite's fixed gates (1e-10, its walks have one and three links) cannot tell that from a fault, so every walk
here carries a TWIN: a second plain handle whose starting velocities are all two ulps larger (the same sign everywhere, so that
the kinetic energies move by 9e-16 relative as a carried sum's do; signs at random cancel in the sums, and that twin read 20-130
times less than the deferred variants with four links).  What the twin has
drifted from the plain handle by is what ONE rounding does to this walk; a variant rounds differently at EVERY step (300-700 of
them in a walk) and its error grows with the twin's.  `--trace` shows it call by call: on three walks that stood out (seeds 60485,
61474, 61498 of --seed0 60000) the variant's error and the twin's rise and fall together, jump at the same calls, and stay within
a factor 1-8 of each other from call 10 to call 499 (profiles/r04_fuzz_soak.md).  Compared every 25 calls only, variants were
seen at up to 800 x the twin (7 of 3 272 walks beyond 200 x, 14 of 3 441 beyond 20 x; all with three to six links, none before
call 174, the twin already grown from 4e-16 to 1e-12..1e-9).  So: the variant must stay within max(suite gate, TWIN_FACTOR =
2000 x the twin's drift), and a walk whose twin has left 1e-6 is given up as chaotic (counted, not failed).  A wrong launch shows
as 1e-3 at the first check after it; the twin is below 1e-11 for a walk's first ~150 calls and below 5e-7 always.
"""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import pytest  # noqa: E402

import test_abi_sequences_gpu as T  # noqa: E402
from openmm_drudenose_amd import synth, HipContext  # noqa: E402
from openmm_drudenose_amd.drudetgnhplugin import (DrudeTGNHIntegrator, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_WAVE_TILES,  # noqa: E402
                                                   FLAG_TRUST_STATE_CHANGED, FLAG_GATHER, TgnhError)
from openmm_drudenose_amd import _lib  # noqa: E402

SYSTEMS = {
    "mixed-60-6": lambda: synth.mixed(60, 6),                       # the suite's own
    "mixed-700-40": lambda: synth.mixed(700, 40),                   # ~5 300 slots: a dozen 512-slot tiles, ~90 wave tiles
    "water-400": lambda: synth.water_box(400),                      # one pattern throughout
    "nacl": lambda: synth.nacl(),
    "ionic-40": lambda: synth.ionic_liquid(40),                     # 35-slot cations: wave tiles refused unless forced
    "polymer-300+200": lambda: synth.polymer_in_water(300, 200),    # one molecule of 900 slots: the COM table of long molecules
    "groups-12": lambda: synth.many_groups(150, 10, 12),            # > 8 temperature groups: the LDS bins
    # --modes only (drawn less often: a walk takes ~1 s): enough tiles for every work-group of the one-launch step to hold several
    "water-8000": lambda: synth.water_box(8000),                    # 40 000 slots, 625 wave tiles
    "mixed-6000-400": lambda: synth.mixed(6000, 400),               # 48 000 slots, four groups, two kinds of tile pattern
    # --modes only: constraint clusters + virtual sites (step() = the split entry points around the harness SHAKE / velocity stage;
    # the walk's own split steps and fused pieces run without the call-outs, on both handles alike)
    "water-rigid-300": lambda: synth.water_box(300, rigid=True),
    "ionic-30-shake": lambda: synth.ionic_liquid(30, constrained=True),
}
BIG = ("water-8000", "mixed-6000-400")
CONSTRAINED = ("water-rigid-300", "ionic-30-shake")


def ragged(k):
    """tests/helpers.py::random_topology(k): molecules of 1-40 slots (some seeds: two longer than a tile), Drudes before or after
    their parents and up to 30 slots away, massless sites, up to 6 groups -- as test_random_ragged_topologies_against_the_oracle"""
    from helpers import random_topology
    mass, pd, pp, resid, group, ngroups, cons, sizes, first, rng = random_topology(k)
    pos = rng.uniform(0.0, 3.0, (len(mass), 3))
    return synth._finish(mass, np.array(pd, np.int32), np.array(pp, np.int32), resid, pos, group, ngroups, rng, 300.0, 1.0, f"ragged{k}")

COM, MODE, PRECISION, DRUDE_CHAINS, HARDWALL, CMM = True, "TGNH", "double", True, 0.02, False
TWIN_FACTOR = 2000.0


def make_build(name):
    def build(flags, exchange, chains, wave=False):
        s, g, ng = ragged(int(name[7:])) if name.startswith("ragged-") else SYSTEMS[name]()
        s.has_cm_motion_remover = CMM
        it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, DRUDE_CHAINS, COM)
        it.setMaxDrudeDistance(HARDWALL)
        for _ in range(ng):
            it.addTempGroup()
        for gi in g:
            it.addParticleTempGroup(int(gi))
        ctx = HipContext(s, it, mode=MODE, precision=PRECISION, flags=flags | (FLAG_WAVE_TILES if wave else 0))
        if exchange == "hook":
            ctx.set_allreduce(lambda t: None)
        elif exchange == "mailbox":
            _, ptr = ctx.exchange_create(1, 0)
            ctx.exchange_attach_pointers([ptr])
        elif exchange == "rccl":
            ctx.rccl_init(1, 0, ctx.rccl_unique_id())
        return s, it, ctx
    return build


class SoakWalk(T.Walk):
    """T.Walk with a twin of the plain handle (see the module text)."""
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        _, _, self.twin = T.build(0, None, a[2], wave=kw["wave"])
        v = self.twin.getVelocities()
        self.twin.setVelocities(v * (1.0 + 4e-16))
        self.twin.compute_forces()
        self.chaotic = False

    def both(self, fn):
        fn(self.ctx); fn(self.ref); fn(self.twin)

    def refused_or(self, call_on, apply_ref):
        ok = super().refused_or(call_on, apply_ref)
        if ok:
            apply_ref(self.twin)
        return ok

    def op_graph(self):
        before = self.ref.time()[1]
        super().op_graph()
        self.twin.step(self.ref.time()[1] - before)

    trace = None                                             # --trace: a file; every call's errors are written there, nothing asserted

    def compare(self, where):
        if self.trace is not None:
            ref_v = self.ref.getVelocities()
            row = [T.rel_err(self.ctx.getVelocities(), ref_v), T.rel_err(self.twin.getVelocities(), ref_v)]
            if not self.flags & FLAG_DEFER_SCALE:
                for which in (0, 1):
                    a, b, t = (c.thermostat_state(which) for c in (self.ctx, self.ref, self.twin))
                    row += [float(np.abs(a - b).max()), float(np.abs(t - b).max())]
            self.trace.write(f"{where} {self.log[-1] if not self.log[-1].startswith('  ') else self.log[-2] + ' refused'} | "
                             + " ".join(f"{x:.3e}" for x in row) + "\n")
            return
        ref_p, ref_v = self.ref.getPositions(), self.ref.getVelocities()
        tp, tv = T.rel_err(self.twin.getPositions(), ref_p), T.rel_err(self.twin.getVelocities(), ref_v)
        if tv > 1e-6:
            self.chaotic = True
            raise StopIteration
        ep, ev = T.rel_err(self.ctx.getPositions(), ref_p), T.rel_err(self.ctx.getVelocities(), ref_v)
        assert ep <= max(1e-12, TWIN_FACTOR * tp) and ev <= max(self.gate_v, TWIN_FACTOR * tv), (where, ep, ev, "twin", tp, tv, self.log[-12:])
        assert self.ctx.time() == pytest.approx(self.ref.time(), rel=1e-12) and self.ctx.check() == 0
        if not self.flags & FLAG_DEFER_SCALE:                # (deferred: the chain has run the next step's first half already)
            for which in (0, 1):
                a, b, t = (c.thermostat_state(which) for c in (self.ctx, self.ref, self.twin))
                tt = float(np.abs(t - b).max())
                gate = dict(rtol=self.gate_t["rtol"], atol=max(self.gate_t["atol"], TWIN_FACTOR * tt))
                assert np.allclose(a, b, **gate), (where, "thermostat", which, float(np.abs(a - b).max()), "twin", tt, self.log[-12:])

    def run(self):
        try:
            super().run()
        except StopIteration:
            pass
        finally:
            for c in (self.ctx, self.ref, self.twin):
                try:
                    c.close()
                except Exception:  # noqa: BLE001
                    pass


def main():
    global COM, MODE, PRECISION, DRUDE_CHAINS, HARDWALL, CMM
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--seed0", type=int, default=5000)
    ap.add_argument("--calls", type=int, default=T.CALLS_PER_WALK)
    ap.add_argument("--modes", action="store_true", help="also draw dualNH / TGNH, mixed / double precision, useDrudeNHChains, the hard wall, "
                    "a CMMotionRemover in the System, ragged random topologies and two 40-50 k-slot boxes")
    ap.add_argument("--constrained", action="store_true", help="with --modes: rigid water and the constrained ionic liquid too")
    ap.add_argument("--gather", action="store_true", help="half of the walks on the gather path (TGNH_FLAG_GATHER) against the tiled plain handle: "
                    "every sum in another order, the suite's cross-kind gates")
    ap.add_argument("--trace", default="", help="with --only: compare after every call and write the errors (variant, twin) here")
    ap.add_argument("--only", default="", help="comma-separated seeds: run just these walks of the sequence --seed0 defines")
    a = ap.parse_args()
    T.CALLS_PER_WALK = a.calls
    only = {int(x) for x in a.only.split(",") if x}
    pick = np.random.default_rng(a.seed0)
    flag_sets = [0, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP, FLAG_TRUST_STATE_CHANGED,
                 FLAG_TRUST_STATE_CHANGED | FLAG_RESIDENT_STEP]
    t_end, n, bad, chaotic = time.time() + 60.0 * a.minutes, 0, 0, 0
    names = [n for n in SYSTEMS if n not in BIG + CONSTRAINED]
    while time.time() < t_end:
        seed = a.seed0 + n
        name = names[int(pick.integers(0, len(names)))]
        flags = flag_sets[int(pick.integers(0, len(flag_sets)))]
        exchange = [None, None, "hook", "mailbox", "rccl"][int(pick.integers(0, 5))]
        chains = int(pick.choice([1, 1, 2, 3, 4, 6]))
        wave = bool(pick.integers(0, 2))
        COM = bool(pick.integers(0, 4))                              # the COM group off in a quarter of the walks
        if a.modes:                                                  # (drawn only when asked for, so that the earlier runs' --seed0 sequences stay what they were)
            MODE = "dualNH" if pick.integers(0, 10) < 3 else "TGNH"  # the Reference platform's algorithm in 30 % of the walks,
            PRECISION = "mixed" if pick.integers(0, 10) < 4 else "double"
            DRUDE_CHAINS = bool(pick.integers(0, 4))                 # ... its coupled-chain quirk (useDrudeNHChains = false) in a quarter
            HARDWALL = 0.0 if pick.integers(0, 4) == 0 else 0.02
            CMM = pick.integers(0, 4) == 0                           # System holds a CMMotionRemover: three degrees of freedom fewer
            u = int(pick.integers(0, 20))
            if u < 6:
                name = f"ragged-{int(pick.integers(0, 200))}"
            elif u < 8:
                name = BIG[u - 6]
            elif u < 10 and a.constrained:
                name = CONSTRAINED[u - 8]
        if a.gather and pick.integers(0, 2):
            flags |= FLAG_GATHER
        what = (f"system={name} flags={flags} exchange={exchange} chains={chains} wave={wave} com={COM} seed={seed}"
                + (f" mode={MODE} precision={PRECISION} drude_chains={DRUDE_CHAINS} hardwall={HARDWALL} cmm={CMM}" if a.modes else ""))
        if only and seed not in only:
            n += 1
            if seed > max(only):
                break
            continue
        T.build = make_build(name)
        t0 = time.time()
        try:
            w = SoakWalk(flags, exchange, chains, seed=seed, wave=wave)
            if flags & FLAG_GATHER:                                  # (tests/test_abi_sequences_gpu.py::test_random_call_sequence_on_the_gather_path's gates)
                w.gate_v, w.gate_t = 5e-10, dict(rtol=1e-7, atol=1e-9)
            if a.trace:
                T.CHECK_EVERY = 1
                w.trace = open(a.trace, "a")
                w.trace.write(f"# {what}: call, op | velocities variant twin [| thermostat 0 variant twin, thermostat 1 variant twin]\n")
            w.run()
            chaotic += w.chaotic
            print(f"{'chaos' if w.chaotic else 'ok   '} {what}  {time.time() - t0:.1f}s", flush=True)
        except Exception as e:  # noqa: BLE001  (a soak: log and go on)
            if isinstance(e, TgnhError) and e.status == _lib.ERR_UNSUPPORTED:
                print(f"skip  {what}  ({str(e)[:200]})", flush=True)   # e.g. wave tiles asked for a molecule longer than a wavefront
                n += 1
                continue
            bad += 1
            print(f"FAIL  {what}  {type(e).__name__}: {str(e)[:1500]}", flush=True)
            traceback.print_exc(limit=4, file=sys.stdout)
        n += 1
    print(f"{n} walks, {bad} failed, {chaotic} given up as chaotic", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
