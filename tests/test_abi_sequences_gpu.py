"""Random call sequences against the C ABI's host state machine (DrudeTGNHIntegrator.cpp:166-194: anything may happen between
two steps -- queries, setters, a changed step size -- and the integrator must go on as if nothing had).

The library steers its launches with a handful of "still owed" flags (tgnh_get_pending_state) x 6 flag combinations x 4 kinds
of exchange; the deterministic tests walk the transitions somebody thought of.  Here a seeded random walk over the entry
points drives a handle of every combination, and a PLAIN handle (flags 0, no exchange: the reference's own pass structure) of
the SAME TILE KIND -- wave tiles or 512-slot tiles, so that the kinetic-energy sums of the two are added in the same order
wherever the two run the same pass -- is fed the same physical sequence: the same steps, the same accepted setters at the same
step boundaries; queries do not change physics.  Every 25 calls positions must agree to 1e-12, velocities to 1e-10 and (where
the variant does not run the chain ahead) the thermostats to rtol 1e-9 / atol 1e-12; a call the variant refuses must be refused
with TGNH_ERR_STATE, and is then not made on the plain handle either -- a refusal is allowed, a wrong trajectory is not.
One more walk compares ACROSS the tile kinds (wave-tile handle, 512-slot-tile reference: different orders of every sum) at
the gate that comparison needs, 5e-10 / 1e-8."""
import ctypes as C

import numpy as np
import pytest

from helpers import rel_err
from openmm_drudenose_amd import synth, HipContext, _lib
from openmm_drudenose_amd.drudetgnhplugin import (DrudeTGNHIntegrator, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_WAVE_TILES,
                                                   FLAG_TRUST_STATE_CHANGED, FLAG_GATHER, TgnhError)

pytestmark = pytest.mark.gpu

CALLS_PER_WALK = 500
CHECK_EVERY = 25


def build(flags, exchange, chains, wave=False):
    s, g, ng = synth.mixed(60, 6)
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, True, True)
    it.setMaxDrudeDistance(0.02)
    for _ in range(ng):
        it.addTempGroup()
    for i, gi in enumerate(g):
        it.addParticleTempGroup(int(gi))
    ctx = HipContext(s, it, mode="TGNH", precision="double", flags=flags | (FLAG_WAVE_TILES if wave else 0))
    if exchange == "hook":
        ctx.set_allreduce(lambda t: None)                    # one rank: the all-reduce is the identity
    elif exchange == "mailbox":
        _, ptr = ctx.exchange_create(1, 0)
        ctx.exchange_attach_pointers([ptr])
    elif exchange == "rccl":
        ctx.rccl_init(1, 0, ctx.rccl_unique_id())
    return s, it, ctx


class Walk:
    def __init__(self, flags, exchange, chains, seed, wave=None, cross=False):
        self.rng = np.random.default_rng(seed)
        wave = seed % 2 == 1 if wave is None else wave                                       # (wave-tile and 512-slot-tile kernels, mixed over the grid)
        self.s, self.it, self.ctx = build(flags, exchange, chains, wave=wave)
        _, self.rit, self.ref = build(0, None, chains, wave=wave != cross)                   # like with like, but for the one cross-kind walk
        self.flags = flags
        self.gate_v, self.gate_t = ((5e-10, dict(rtol=1e-8, atol=1e-9)) if cross else (1e-10, dict(rtol=1e-9, atol=1e-12)))
        if flags & FLAG_TRUST_STATE_CHANGED and exchange is None and not cross:
            # a carried bin is KE prod exp(-dtc etaDot) (Cu :574), the plain handle's is summed from the velocities again: equal to
            # rounding, not bit for bit -- two different sums, as across the tile kinds, and the same gate: the higher links of a
            # three-link chain amplify a rounding of etaDot_0 by Q_0 / Q_1 ~ the number of degrees of freedom (Cu :588; measured
            # 1.3e-9 relative in the third link's etaDot after ~700 steps, 3.9e-12 absolute in eta); a sum that survived a write of
            # the velocities shows as 1e-3
            self.gate_t = dict(rtol=1e-8, atol=1e-9)
        self.replay = None
        self.log = []

    def both(self, fn):
        fn(self.ctx); fn(self.ref)

    def refused_or(self, call_on, apply_ref):
        """a setter: made on the variant first; refused (ERR_STATE) = skipped on both, anything else must succeed"""
        try:
            call_on(self.ctx)
        except TgnhError as e:
            assert e.status == _lib.ERR_STATE, (self.log[-5:], e)
            self.log.append("  (refused)")
            return False
        apply_ref(self.ref)
        return True

    # ---- physical operations (made on both handles)
    def op_step(self):
        n = int(self.rng.integers(1, 4))
        self.both(lambda c: c.step(n))

    def op_step_pieces(self):
        self.both(lambda c: (c.step_begin(), c.compute_forces(), c.step_end()))

    def op_split_step(self):
        def split(c):
            lib, h, st = c.lib, c.h, c._stream
            assert lib.tgnh_step_begin_kick(h, st()) == 0
            assert lib.tgnh_step_begin_move(h, st()) == 0
            c.compute_forces()
            assert lib.tgnh_step_end_kick(h, st()) == 0
            assert lib.tgnh_step_end_thermo(h, st()) == 0
            c.ke_sum_valid = True
        self.both(split)

    def op_graph(self):
        """steps recorded into a hipGraph on the variant, the same number of eager steps on the plain handle"""
        k, reps = int(self.rng.integers(1, 4)), int(self.rng.integers(1, 4))
        before = self.ctx.time()[1]
        rep = self.ctx.capture_steps(k)
        for _ in range(reps):
            rep()
        self.ctx.torch.cuda.synchronize()
        taken = self.ctx.time()[1] - before                  # (capture_steps takes real steps first when the handle is not in its steady state)
        assert taken >= k * reps
        self.ref.step(taken)

    def op_set_step_size(self):
        dt, old = float(self.rng.choice([0.001, 0.0005, 0.00075])), self.ctx.integrator.getStepSize()
        if not self.refused_or(lambda c: c.integrator.setStepSize(dt), lambda c: c.integrator.setStepSize(dt)):
            self.ctx.integrator._stepSize = old              # (the setter pushes to the handle: refused, the integrator keeps its value)

    def op_set_drude_steps(self):
        n, old = int(self.rng.choice([20, 10, 5])), self.ctx.integrator.getDrudeStepsPerRealStep()
        if not self.refused_or(lambda c: c.integrator.setDrudeStepsPerRealStep(n), lambda c: c.integrator.setDrudeStepsPerRealStep(n)):
            self.ctx.integrator._drudeSteps = old

    def op_set_velocities(self):
        """Context::setVelocities (stateChanged, DrudeTGNHIntegrator.cpp:166-170): the current velocities scaled by 0.999"""
        v = self.ctx.getVelocities() * 0.999
        if self.refused_or(lambda c: c.setVelocities(v), lambda c: c.setVelocities(c.getVelocities() * 0.999)):
            self.both(lambda c: c.compute_forces())

    def op_remove_cm(self):
        """the harness' CMMotionRemover: velocities change behind the integrator, the library's own call-out knows it"""
        def cm(c):
            if self.flags & FLAG_DEFER_SCALE:                # deferred: velocities lag between steps and the next half step has run
                c._state_changed()                           # already -- the contract is stateChanged first, which such a handle refuses
                c.getVelocities()
            assert c.lib.tgnh_harness_remove_cm_motion(c.h, c._stream()) == 0
        self.refused_or(cm, lambda c: c.lib.tgnh_harness_remove_cm_motion(c.h, c._stream()))

    def op_rebind(self):
        """tgnh_bind_buffers: the same arrays again (what the OpenMM glue does at every step), or the velocities in another array"""
        def rebind(c, move):
            if move:
                c.torch.cuda.synchronize(c.dev)
                c.velm = c.velm.clone()
            rc = c.lib.tgnh_bind_buffers(c.h, c.posq.data_ptr(), c.posq_corr.data_ptr() if c.posq_corr is not None else None,
                                         c.velm.data_ptr(), c.force.data_ptr(), c.pos_delta.data_ptr())
            if rc != 0:
                raise TgnhError(rc, c.lib.tgnh_last_error().decode())
        move = bool(self.rng.integers(0, 2))
        if move:
            self.ctx.getVelocities()                         # (a moved array holds the reference's end-of-step velocities: settle what is pending first)
        self.refused_or(lambda c: rebind(c, move), lambda c: rebind(c, move))

    def op_set_thermostat(self):
        which = int(self.rng.integers(0, 2))
        try:
            cur = self.ctx.thermostat_state(which)
        except TgnhError as e:
            assert e.status == _lib.ERR_STATE
            return
        self.refused_or(lambda c: c.set_thermostat_state(which, cur), lambda c: None)     # (its own values: no change of physics)

    # ---- queries (variant only: they must not change its trajectory)
    def op_flush(self):
        assert self.ctx.lib.tgnh_flush(self.ctx.h, self.ctx._stream()) == 0

    def op_queries(self):
        c = self.ctx
        pick = int(self.rng.integers(0, 6))
        if pick == 0:
            c.getVelocities()
        elif pick == 1:
            c.kinetic_energy()
        elif pick == 2:
            c.last_kinetic_energies(); c.last_scale_factors()
        elif pick == 3:
            c.thermostat_state(int(self.rng.integers(0, 4)))
        elif pick == 4:
            c.compute_kinetic_energies()
        else:
            assert c.check() == 0 and c.time()[1] == self.ref.time()[1]

    OPS = [("step", 6), ("step_pieces", 3), ("split_step", 2), ("graph", 1), ("set_step_size", 1), ("set_drude_steps", 1),
           ("set_velocities", 1), ("set_thermostat", 1), ("remove_cm", 1), ("rebind", 1), ("flush", 2), ("queries", 5)]

    def compare(self, where):
        pos, vel = self.ctx.getPositions(), self.ctx.getVelocities()
        ep, ev = rel_err(pos, self.ref.getPositions()), rel_err(vel, self.ref.getVelocities())
        assert ep <= 1e-12 and ev <= self.gate_v, (where, ep, ev, self.log[-12:])      # (a wrong launch shows as 1e-3)
        assert self.ctx.time() == pytest.approx(self.ref.time(), rel=1e-12) and self.ctx.check() == 0
        if not self.flags & FLAG_DEFER_SCALE:                # (deferred: the chain has run the next step's first half already)
            for which in (0, 1):
                a, b = self.ctx.thermostat_state(which), self.ref.thermostat_state(which)
                worst = float(np.max(np.abs(a - b) / (self.gate_t["atol"] + self.gate_t["rtol"] * np.abs(b))))
                assert np.allclose(a, b, **self.gate_t), (where, which, worst, float(np.abs(a - b).max()), self.log[-12:])

    def run(self):
        names = [n for n, w in self.OPS for _ in range(w)]
        for i in range(CALLS_PER_WALK):
            name = names[int(self.rng.integers(0, len(names)))]
            self.log.append(f"{i}: {name} (owed {self.ctx.pending_state():#x})")
            getattr(self, "op_" + name)()
            if (i + 1) % CHECK_EVERY == 0:
                self.compare(i)
        self.ctx.close(); self.ref.close()


TRUST = FLAG_TRUST_STATE_CHANGED


@pytest.mark.parametrize("exchange", [None, "hook", "mailbox", "rccl"])
@pytest.mark.parametrize("flags", [0, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP, TRUST, TRUST | FLAG_RESIDENT_STEP])
def test_random_call_sequences(flags, exchange):
    """(TRUST with an exchange attached is ignored by the library: those walks check exactly that)"""
    chains = 1 if (flags + (0 if exchange is None else 1)) % 2 == 0 else 3          # one- and three-link chains, mixed over the grid
    Walk(flags, exchange, chains, seed=1000 + 16 * flags + len(exchange or "")).run()


@pytest.mark.parametrize("chains", [1, 3])
def test_kinetic_energies_carried_over_survive_random_call_sequences(chains):
    """TGNH_FLAG_TRUST_STATE_CHANGED without an exchange, both tile kinds: every call that may write velocities between two steps
    (setVelocities, the split entry points, the harness' CMMotionRemover, velocities bound to another array, a thermostat set)
    must make the next half step sum its kinetic energies again -- a carried sum that survived one of them shows as 1e-3."""
    for wave in (False, True):
        w = Walk(TRUST, None, chains, seed=77 + chains, wave=wave)
        w.run()


def test_random_call_sequence_across_tile_kinds():
    """a wave-tile handle against a 512-slot-tile reference: every sum in another order, the loose gate"""
    Walk(FLAG_DEFER_SCALE, None, 3, seed=4242, wave=True, cross=True).run()


@pytest.mark.parametrize("flags,exchange,chains", [(FLAG_GATHER, None, 3), (FLAG_GATHER | FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP, "hook", 1)])
def test_random_call_sequence_on_the_gather_path(flags, exchange, chains):
    """The gather path (tgnh_gather.hip: the reference's un-fused kernels by global index, here forced with TGNH_FLAG_GATHER on a
    topology the tiles can hold) against a TILED plain reference over a random walk of the entry points -- steps, split steps with
    the harness' constraint call-outs, setters, queries, a rebind: every sum in another order, the cross-kind gate.  The flags that
    change the pass structure are ignored on it (nothing is owed between steps, setters are never refused)."""
    w = Walk(flags, exchange, chains, seed=9000 + flags, wave=False, cross=True)
    # the third link of a three-link chain amplifies a rounding of etaDot_0 by ~ the number of degrees of freedom (Walk.__init__):
    # 1.5e-8 relative in its etaDot at call 474 of this walk with the gather path's particle-order sums, positions 1e-12 and
    # velocities < 5e-10 throughout; a wrong launch shows as 1e-3
    w.gate_t = dict(rtol=1e-7, atol=1e-9)
    assert w.ctx.step_path()[0] == "gather" and w.ref.step_path()[0] == "tiled"
    w.run()
