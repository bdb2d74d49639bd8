"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, plus size-independent properties at the full 1 M-pair size.

Tolerances (BASELINE.json north_star): pair/group indexing bit-exact; positions and
velocities within 1e-6 relative (max-norm) over 100 steps in the precisions that carry
fp64 velocities (mixed, double).  Single precision (float4 state) cannot hold 1e-6 against a
double-precision oracle with a stiff Drude spring; its deviation is measured and bounded
by the looser figure written in test_single_precision_deviation.
"""
import numpy as np
import pytest

from openmm_drudenose_amd import synth, _lib
from openmm_drudenose_amd.drudetgnhplugin import (DrudeTGNHIntegrator, HipContext, TgnhError,
                                                   FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_WAVE_TILES, FLAG_TRUST_STATE_CHANGED, FLAG_GATHER)
from helpers import make_oracle, oracle_run, rel_err, to_internal

pytestmark = pytest.mark.gpu

TOL = 1e-6          # north_star: positions / velocities, 100 steps
TOL_KE = 1e-6       # KE[] and scale[] per step on the GPU


def integ(chains=3, drude_chains=True, com=True, hardwall=0.0, dt=0.001):
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, dt, 20, chains, drude_chains, com)
    it.setMaxDrudeDistance(hardwall)
    return it


def bind_groups(it, group, ngroups):
    for _ in range(ngroups):
        it.addTempGroup()
    for g in group:
        it.addParticleTempGroup(int(g))


def bind_groups_array(it, group, ngroups):
    """the same for millions of particles: the array goes in whole (a Python list that long is slow to build and to walk)"""
    for _ in range(ngroups):
        it.addTempGroup()
    it._particleTempGroup = np.ascontiguousarray(group, np.int32)


SYSTEMS = {
    "groups6": lambda: synth.many_groups(300, 20, 6),           # 5-8 groups: the widest register-bin instantiation
    "groups12": lambda: synth.many_groups(300, 20, 12),         # > 8 groups: KE bins in LDS
    "groups32": lambda: synth.many_groups(300, 20, 32),
    "polymer": lambda: synth.polymer_in_water(700, 300),       # one 2100-slot molecule (longer than a tile) + waters
    "pair+normal+massless": lambda: synth.pair_normal_massless(),
    "water27": lambda: synth.water_box(27),
    "water1000": lambda: synth.water_box(1000),
    "nacl": lambda: synth.nacl(),
    "il40": lambda: synth.ionic_liquid(40),
    "mixed": lambda: synth.mixed(300, 20),
    "ragged6": lambda: ragged_system(6),                       # random molecule sizes / pair placements, 4 groups
}


def ragged_system(seed):
    from helpers import random_topology
    mass, pd, pp, resid, group, ngroups, cons, sizes, first, rng = random_topology(seed)
    pos = rng.uniform(0.0, 3.0, (len(mass), 3))
    return synth._finish(mass, np.array(pd, np.int32), np.array(pp, np.int32), resid, pos, group, ngroups, rng, 300.0, 1.0,
                         f"ragged{seed}")


def make(sysname, mode, precision, flags=0, tiles="wave", **kw):
    """tiles = "wave": the wave-tile kernels wherever the topology has wave tiles (TGNH_FLAG_WAVE_TILES; the library's own rule
    takes them only where they are >= 90 % full, i.e. not for the ion-pair systems here); "lds": the library's rule -- the
    512-slot tile kernels for those.  Tests that build their HipContext themselves run under the library's rule."""
    if tiles == "wave":
        flags |= FLAG_WAVE_TILES
    s, g, ng = SYSTEMS[sysname]()
    it = integ(**kw)
    if mode == "TGNH":
        bind_groups(it, g, ng)
    else:
        g, ng = np.zeros_like(g), 1
    ctx = HipContext(s, it, mode=mode, precision=precision, flags=flags)
    return s, g, ng, it, ctx


# ---------------------------------------------------------------------------
# A1 / A2: indexing bit-exact, dof
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("sysname", ["pair+normal+massless", "nacl", "il40", "mixed"])
def test_topology_bit_exact(sysname):
    s, g, ng, it, ctx = make(sysname, "TGNH", "double")
    o = make_oracle(s, g, ng, "TGNH", it)
    assert np.array_equal(ctx.topology(0), o.normal_particles())
    assert np.array_equal(ctx.topology(1), s.pair_drude) and np.array_equal(ctx.topology(2), s.pair_parent)
    assert np.array_equal(ctx.topology(3), g) and np.array_equal(ctx.topology(4), s.resid)
    count = np.bincount(s.resid, minlength=s.num_residues)
    first = np.array([np.flatnonzero(s.resid == r)[0] for r in range(s.num_residues)])
    assert np.array_equal(ctx.topology(5), count) and np.array_equal(ctx.topology(6), first)
    # tiles never cut a Drude pair or a molecule; packed words decode to the same topology
    ts = ctx.topology(7)
    tile_of = np.searchsorted(ts, np.arange(s.num_particles), side="right") - 1
    assert np.all(tile_of[s.pair_drude] == tile_of[s.pair_parent])
    assert np.all(np.diff(ts) <= 512) and ts[0] == 0 and ts[-1] == s.num_particles
    for r in range(s.num_residues):
        assert len(set(tile_of[s.resid == r])) == 1
    meta = ctx.topology(8).view(np.uint32)
    role = meta & 3
    assert np.array_equal(np.flatnonzero(role == 1), np.sort(s.pair_drude))
    assert np.array_equal(np.flatnonzero(role == 2), np.sort(s.pair_parent))
    assert np.array_equal(((meta >> 2) & 255).astype(np.int32), g)
    off = ((meta >> 10) & 2047).astype(np.int64) - 1024
    assert np.array_equal((np.arange(s.num_particles) + off)[s.pair_drude], s.pair_parent)
    ctx.close()


@pytest.mark.parametrize("mode", ["dualNH", "TGNH"])
@pytest.mark.parametrize("com", [True, False])
def test_dof_and_thermostat_masses(mode, com):
    s, g, ng, it, ctx = make("mixed", mode, "double", com=com)
    s.has_cm_motion_remover = False
    o = make_oracle(s, g, ng, mode, it)
    dof_o, nkt_o = o.dof()
    dof, nkt = ctx.dof()
    assert np.allclose(dof, to_internal(dof_o, mode), rtol=1e-14, atol=0)
    assert np.allclose(nkt, to_internal(nkt_o, mode), rtol=1e-14, atol=0)
    assert np.allclose(ctx.thermostat_state(3), o.chain(3), rtol=1e-14, atol=0)       # etaMass
    assert np.allclose(ctx.thermostat_state(2), o.chain(2), rtol=1e-14, atol=0)       # etaDotDot init
    ctx.close()


def test_group_mismatch_is_an_error():
    s, g, ng = synth.water_box(4)
    it = integ()
    g = g.copy()
    g[1] = 1
    bind_groups(it, g, 2)
    with pytest.raises(TgnhError, match="Temperature group for drude particle") as e:     # Cu :145-146
        HipContext(s, it, mode="TGNH", precision="double")
    assert e.value.status == _lib.ERR_GROUP_MISMATCH


# ---------------------------------------------------------------------------
# A3/A4, A7 single kernels
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["dualNH", "TGNH"])
@pytest.mark.parametrize("precision,tol", [("double", 1e-12), ("mixed", 1e-12), ("single", 1e-6)])
@pytest.mark.parametrize("sysname", ["pair+normal+massless", "il40", "mixed", "groups12", "groups32", "polymer"])
def test_kinetic_energies(sysname, mode, precision, tol):
    s, g, ng, it, ctx = make(sysname, mode, precision)
    o = make_oracle(s, g, ng, mode, it)
    ke = ctx.compute_kinetic_energies()
    ke_o = to_internal(o.kinetic_energies(s.velocities), mode)
    assert np.allclose(ke, ke_o, rtol=tol, atol=tol * np.abs(ke_o).max())
    ctx.close()


@pytest.mark.parametrize("precision,tol", [("double", 1e-13), ("mixed", 1e-13), ("single", 1e-6)])
def test_half_kick(precision, tol):
    s, g, ng, it, ctx = make("mixed", "TGNH", precision)
    o = make_oracle(s, g, ng, "TGNH", it)
    rng = np.random.default_rng(5)
    f = rng.normal(0, 300.0, s.positions.shape)
    ctx.setForces(f)
    fq = ctx.getForces()                       # what the fixed-point layout holds
    assert rel_err(fq, f) < 1e-9
    ctx.half_kick()
    v = s.velocities.copy()
    o.half_kick(v, fq)
    assert rel_err(ctx.getVelocities(), v) < tol
    ctx.close()


# ---------------------------------------------------------------------------
# A11: 100-step parity
# ---------------------------------------------------------------------------
CASES = [
    # sysname, mode, chains, drude_chains, com, hardwall
    ("pair+normal+massless", "dualNH", 1, False, True, 0.0),
    ("pair+normal+massless", "TGNH", 3, True, True, 0.0),
    ("water27", "dualNH", 3, True, True, 0.0),
    ("water27", "dualNH", 3, False, True, 0.0),          # bug-compatible coupled chain
    ("water27", "TGNH", 1, True, False, 0.0),            # bridge configuration A9
    ("water1000", "TGNH", 3, True, True, 0.0),
    ("nacl", "TGNH", 1, True, True, 0.0),
    ("nacl", "dualNH", 1, False, True, 0.0),
    ("il40", "TGNH", 3, True, True, 0.0),
    ("il40", "TGNH", 3, False, True, 0.0),
    ("mixed", "TGNH", 3, True, True, 0.0),
    ("mixed", "TGNH", 2, True, False, 0.0),
    ("mixed", "dualNH", 2, True, True, 0.0),
    ("groups12", "TGNH", 1, True, True, 0.0),             # 12 and 32 temperature groups
    ("groups32", "TGNH", 2, True, True, 0.02),
    ("polymer", "TGNH", 1, True, True, 0.0),              # molecule longer than a tile: COM from big_com_kernel
    ("polymer", "TGNH", 3, True, True, 0.0),
    ("polymer", "dualNH", 1, True, True, 0.0),
    ("mixed", "TGNH", 10, True, True, 0.0),               # chains of 5-16 links: a link per lane (chain_lanes_run); 10 = the reference test's value
    ("mixed", "TGNH", 6, False, True, 0.02),              # ... the Drude thermostat's higher links frozen
    ("groups12", "TGNH", 7, True, True, 0.0),             # ... 14 thermostats: one batch of 16 rows
    ("groups32", "TGNH", 5, True, False, 0.0),            # ... 34 thermostats: three batches; COM thermostat inert (Q = 0)
    ("water27", "TGNH", 16, True, True, 0.0),             # ... a full row
    ("water27", "TGNH", 17, True, True, 0.0),             # longer still: the LDS-resident chain
    ("water27", "dualNH", 10, False, True, 0.0),          # the Reference platform's own test setting: ten links, the coupled chain (chain_dualnh_long_kernel)
    ("mixed", "dualNH", 5, False, True, 0.02),            # dualNH chains of 5-16 links in registers: the coupled form ...
    ("water27", "dualNH", 16, False, True, 0.0),
    ("mixed", "dualNH", 7, True, True, 0.0),              # ... and the two independent chains (useDrudeNHChains), a lane each
    ("water27", "dualNH", 16, True, True, 0.02),
    ("water27", "dualNH", 17, False, True, 0.0),          # longer still: the transcription on LDS-resident vectors
    ("water27", "dualNH", 1, True, True, 0.02),           # dualNH one-link chains: the in-kernel chain's code path
    ("mixed", "dualNH", 1, True, True, 0.0),
]


@pytest.mark.parametrize("tiles", ["wave", "lds"])
@pytest.mark.parametrize("precision", ["mixed", "double"])
@pytest.mark.parametrize("sysname,mode,chains,drude_chains,com,hardwall", CASES)
def test_100_step_parity(sysname, mode, chains, drude_chains, com, hardwall, precision, tiles):
    s, g, ng, it, ctx = make(sysname, mode, precision, chains=chains, drude_chains=drude_chains, com=com,
                             hardwall=hardwall, tiles=tiles)
    o = make_oracle(s, g, ng, mode, it)
    pos_o, vel_o, kes, scs = oracle_run(o, s, 100, record=True, x0=ctx.sites())
    kes, scs = to_internal(kes, mode), to_internal(scs, mode)
    worst_ke = worst_sc = 0.0
    for i in range(100):
        ctx.step_begin()
        ke, sc = ctx.last_kinetic_energies(), ctx.last_scale_factors()
        worst_ke = max(worst_ke, np.abs(ke - kes[2 * i]).max() / np.abs(kes[2 * i]).max())
        m = np.ones(len(sc), bool)
        if mode == "dualNH":
            m[1] = False
        worst_sc = max(worst_sc, np.abs(sc[m] - scs[2 * i][m]).max())
        ctx.compute_forces()
        ctx.step_end()
        ke, sc = ctx.last_kinetic_energies(), ctx.last_scale_factors()
        worst_ke = max(worst_ke, np.abs(ke - kes[2 * i + 1]).max() / np.abs(kes[2 * i + 1]).max())
        worst_sc = max(worst_sc, np.abs(sc[m] - scs[2 * i + 1][m]).max())
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"{sysname} {mode} {precision}: pos {ep:.2e} vel {ev:.2e} KE {worst_ke:.2e} scale {worst_sc:.2e}")
    assert ep <= TOL and ev <= TOL
    assert worst_ke <= TOL_KE and worst_sc <= TOL_KE
    t, k = ctx.time()
    assert k == 100 and t == pytest.approx(0.1, rel=1e-12)
    # thermostat variables end in the same place
    for which in (0, 1):
        a, b = ctx.thermostat_state(which), o.chain(which)
        assert np.allclose(a, b, rtol=1e-6, atol=1e-9 * max(1.0, np.abs(b).max()))
    ctx.close()


TRUST = FLAG_TRUST_STATE_CHANGED
TRUST_CASES = [
    # sysname, mode, chains, drude_chains, flags
    ("mixed", "TGNH", 1, True, TRUST),                        # the chain inside the rescale launch, reading the staged block where it lies
    ("mixed", "TGNH", 3, True, TRUST),                        # 2-4 links inside the rescale launch
    ("mixed", "TGNH", 10, True, TRUST),                       # chain_kernel (a link per lane) started from the carried sums
    ("water1000", "TGNH", 1, True, TRUST | FLAG_RESIDENT_STEP),   # the end half as one step_kernel launch, the begin half the tile launch
    ("il40", "dualNH", 1, False, TRUST),                      # the Reference platform's coupled one-link chain
    ("mixed", "dualNH", 3, True, TRUST),
    ("polymer", "TGNH", 1, True, TRUST),                      # a molecule longer than a tile: its COM table is refreshed without the KE pass
    ("groups12", "TGNH", 2, True, TRUST),                     # more than 8 groups: LDS bins, chain_kernel
]


@pytest.mark.parametrize("precision", ["mixed", "double"])
@pytest.mark.parametrize("sysname,mode,chains,drude_chains,flags", TRUST_CASES)
def test_100_step_parity_trust_state_changed(sysname, mode, chains, drude_chains, flags, precision):
    """TGNH_FLAG_TRUST_STATE_CHANGED: the reference's pass structure (velocities never lag) whose begin half starts its chain from
    the kinetic energies the last end half's chain left (s^2 KE, Cu :574) instead of summing them again (Cu :474-488).  Against the
    oracle, which does sum them: (a) 100 undisturbed steps -- one KE pass in all, the first step's; (b) the same with every half step
    queried (queries settle what is pending: the other launch sequence), per-half-step kinetic energies and scale factors
    included; (c) velocities set in mid-run (stateChanged, DrudeTGNHIntegrator.cpp:166-170): the next half sums again."""
    s, g, ng, it, ctx = make(sysname, mode, precision, flags=flags, chains=chains, drude_chains=drude_chains, hardwall=0.02, tiles="lds")
    o = make_oracle(s, g, ng, mode, it)
    pos_o, vel_o, kes, scs = oracle_run(o, s, 100, record=True, x0=ctx.sites())
    kes, scs = to_internal(kes, mode), to_internal(scs, mode)
    ctx.timing(True)
    ctx.step(100)
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    ctx.timing(False)
    # (with RESIDENT_STEP the half steps that do sum are step_kernel launches: 100 end halves + the first step's begin half)
    resident = bool(flags & FLAG_RESIDENT_STEP) and ctx.resident_work_groups() > 0
    ke_passes = ctx.timing_read(_lib.KID_KE)[1] + (ctx.timing_read(_lib.KID_STEP)[1] - 100 if resident else 0)
    print(f"{sysname} {mode} {precision} chains {chains}: undisturbed pos {ep:.2e} vel {ev:.2e}, {ke_passes} begin halves summed in 100 steps")
    assert ep <= TOL and ev <= TOL
    assert ke_passes == 1 and ctx.pending_state() & 512
    for which in (0, 1):
        a, b = ctx.thermostat_state(which), o.chain(which)
        assert np.allclose(a, b, rtol=1e-6, atol=1e-9 * max(1.0, np.abs(b).max()))
    ctx.close()
    # (b) every half step queried
    s, g, ng, it, ctx = make(sysname, mode, precision, flags=flags, chains=chains, drude_chains=drude_chains, hardwall=0.02, tiles="lds")
    worst_ke = worst_sc = 0.0
    m = np.ones(kes.shape[1], bool)
    if mode == "dualNH":
        m[1] = False
    for i in range(100):
        ctx.step_begin()
        ke, sc = ctx.last_kinetic_energies(), ctx.last_scale_factors()
        worst_ke = max(worst_ke, np.abs(ke - kes[2 * i]).max() / np.abs(kes[2 * i]).max())
        worst_sc = max(worst_sc, np.abs(sc[m] - scs[2 * i][m]).max())
        ctx.compute_forces()
        ctx.step_end()
        ke, sc = ctx.last_kinetic_energies(), ctx.last_scale_factors()
        worst_ke = max(worst_ke, np.abs(ke - kes[2 * i + 1]).max() / np.abs(kes[2 * i + 1]).max())
        worst_sc = max(worst_sc, np.abs(sc[m] - scs[2 * i + 1][m]).max())
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"    queried: pos {ep:.2e} vel {ev:.2e} KE {worst_ke:.2e} scale {worst_sc:.2e}")
    assert ep <= TOL and ev <= TOL and worst_ke <= TOL_KE and worst_sc <= TOL_KE
    ctx.close()
    # (c) stateChanged in mid-run
    s, g, ng, it, ctx = make(sysname, mode, precision, flags=flags, chains=chains, drude_chains=drude_chains, hardwall=0.02, tiles="lds")
    o = make_oracle(s, g, ng, mode, it)
    x0 = ctx.sites()
    po, vo = s.positions.copy(), s.velocities.copy()
    f = o.harness_force(po, x0, synth.K_DRUDE, synth.K_TETHER)
    o.run_harness(po, vo, f, x0, synth.K_DRUDE, synth.K_TETHER, 10)
    vo *= 0.9
    o.run_harness(po, vo, f, x0, synth.K_DRUDE, synth.K_TETHER, 10)
    ctx.timing(True)
    ctx.step(10)
    assert ctx.pending_state() & 512
    ctx.setVelocities(ctx.getVelocities() * 0.9)
    assert not ctx.pending_state() & 512
    ctx.step(10)
    ep, ev = rel_err(ctx.getPositions(), po), rel_err(ctx.getVelocities(), vo)
    ctx.timing(False)
    print(f"    velocities set after 10 steps: pos {ep:.2e} vel {ev:.2e}")
    summed = ctx.timing_read(_lib.KID_KE)[1] + (ctx.timing_read(_lib.KID_STEP)[1] - 20 if resident else 0)
    assert ep <= TOL and ev <= TOL and summed == 2
    ctx.close()


def test_trust_state_changed_is_ignored_where_it_cannot_hold():
    """A molecule that spans two temperature groups (s^2 KE is then not the rescaled velocities' bin), DEFER_SCALE, a sharded
    handle: the flag changes nothing -- every begin half sums its kinetic energies, the trajectory is the plain one."""
    s, g, ng = synth.water_box(300)
    g = g.copy(); g[2::5] = 1                                   # every H1 in a group of its own
    runs = []
    for flags in (0, TRUST):
        it = integ(chains=1, hardwall=0.02)
        bind_groups(it, g, 2)
        ctx = HipContext(s, it, mode="TGNH", precision="double", flags=flags)
        ctx.timing(True)
        ctx.step(20)
        runs.append((ctx.getPositions(), ctx.getVelocities()))
        ctx.timing(False)
        assert ctx.timing_read(_lib.KID_KE)[1] == 20 and not ctx.pending_state() & 512
        ctx.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    s, g, ng, it, ctx = make("mixed", "TGNH", "double", flags=TRUST, chains=1)
    ctx.set_allreduce(lambda t: None)                            # a collective hook: sharded as far as the handle knows
    ctx.timing(True)
    ctx.step(5)
    ctx.timing(False)
    assert ctx.timing_read(_lib.KID_KE)[1] == 5 and not ctx.pending_state() & 512
    ctx.close()


FAR_CASES = [   # (mode, links, flags): every form of the chain -- chain_kernel's, the in-kernel ones, a link per lane, LDS-resident links
    ("TGNH", 1, 0), ("TGNH", 1, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP), ("TGNH", 1, FLAG_DEFER_SCALE),
    ("TGNH", 3, 0), ("TGNH", 3, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP), ("TGNH", 3, FLAG_DEFER_SCALE), ("TGNH", 2, 0), ("TGNH", 4, FLAG_DEFER_SCALE),
    ("TGNH", 10, 0), ("TGNH", 17, 0),
    ("dualNH", 1, 0), ("dualNH", 1, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP), ("dualNH", 3, 0), ("dualNH", 3, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP),
]


@pytest.mark.parametrize("amp", [800.0, 8000.0, 60000.0])
@pytest.mark.parametrize("mode,chains,flags", FAR_CASES)
def test_chain_far_from_equilibrium(mode, chains, flags, amp):
    """Thermostats far from equilibrium (Cu :566-592: exp(-dtc/8 etaDot[i+1]), exp(-dtc/2 etaDot[0])): with etaDot set to +-amp
    ps^-1 the exponents per sub-step are 0.02 (outside the short polynomials: chain_exp_wide), 0.2 and 1.5 (beyond it: the
    ln 2 reduction), on every form of the chain.  An equilibrating box of a million pairs sits in the first two regimes
    (profiles/r03_chain_cost.md); the parity cases elsewhere stay inside the short polynomial."""
    s, g, ng, it, ctx = make("mixed", mode, "double", chains=chains, flags=flags, tiles="lds")
    o = make_oracle(s, g, ng, mode, it)
    ed = ctx.thermostat_state(1)
    assert np.all(ed == 0) and np.array_equal(o.chain(1), ed)
    if mode == "TGNH":
        live = np.ones(len(ed), bool)
        live[chains::chains + 1] = False                     # every thermostat's dummy link stays at 0 (Cu :252)
    else:
        live = np.arange(len(ed)) < len(ed) - 2              # the interleaved vectors' two dummies (Ref :215)
    rng = np.random.default_rng(int(amp) + chains)
    ed[live] = amp * rng.choice([-1.0, 1.0], live.sum()) * rng.uniform(0.5, 1.0, live.sum())
    ctx.set_thermostat_state(1, ed)
    o.set_chain(1, ed)
    pos_o, vel_o = oracle_run(o, s, 3, x0=ctx.sites())
    if not (np.isfinite(vel_o).all() and np.isfinite(o.chain(1)).all()):
        ctx.close()
        pytest.skip("the reference arithmetic itself overflows here (long chain, links driven by amp^2)")
    ctx.step(3)
    vscale = max(np.abs(vel_o).max(), 1e-3 * np.abs(s.velocities).max())     # (a thermostat this far out freezes the box within a step: compare against what is left, not against 1e-12 nm/ps)
    ep, ev = rel_err(ctx.getPositions(), pos_o), np.abs(ctx.getVelocities() - vel_o).max() / vscale
    if flags & FLAG_DEFER_SCALE:                             # (deferred: the library's chain has run the coming step's first half already)
        o.propagate_nhc(vel_o.copy())
    worst = 0.0
    for which in (0, 1, 2):
        a, b = ctx.thermostat_state(which), o.chain(which)
        worst = max(worst, np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    print(f"{mode} {chains} links flags {flags} amp {amp:g}: pos {ep:.2e} vel {ev:.2e} thermostat {worst:.2e} |v|max {np.abs(vel_o).max():.3g}")
    assert ep <= 1e-10 and ev <= 1e-10 and worst <= 1e-10 and ctx.check() == 0
    ctx.close()


@pytest.mark.parametrize("precision", ["mixed", "double"])
@pytest.mark.parametrize("mode", ["dualNH", "TGNH"])
def test_100_step_parity_hardwall(mode, precision):
    """Hard wall on and actually hit: Drudes start hot (5 K spread) against a 0.0006 nm wall... the wall
    is placed so that a fraction of the pairs bounce every few steps."""
    old = synth.DRUDE_SIGMA
    synth.DRUDE_SIGMA = 0.0004
    try:
        s, g, ng = synth.mixed(200, 10)
    finally:
        synth.DRUDE_SIGMA = old
    it = integ(chains=2, hardwall=0.0012)
    if mode == "TGNH":
        bind_groups(it, g, ng)
    else:
        g, ng = np.zeros_like(g), 1
    ctx = HipContext(s, it, mode=mode, precision=precision)
    o = make_oracle(s, g, ng, mode, it)
    # count bounces in the oracle run to make sure the branch is exercised
    pos, vel, x0 = s.positions.copy(), s.velocities.copy(), ctx.sites()
    f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
    bounces = 0
    for _ in range(100):
        o.propagate_nhc(vel); o.half_kick(vel, f); o.drift(pos, vel)
        r = np.linalg.norm(pos[s.pair_drude] - pos[s.pair_parent], axis=1)
        bounces += int((r > 0.0012).sum())
        o.hardwall(pos, vel)
        f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
        o.half_kick(vel, f); o.propagate_nhc(vel)
    assert bounces > 20
    ctx.step(100)
    assert ctx.check() == 0
    ep, ev = rel_err(ctx.getPositions(), pos), rel_err(ctx.getVelocities(), vel)
    print(f"hardwall {mode} {precision}: bounces {bounces} pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL
    r = np.linalg.norm(ctx.getPositions()[s.pair_drude] - ctx.getPositions()[s.pair_parent], axis=1)
    assert r.max() <= 0.0012 * (1 + 1e-6)              # the reference's own assertion (TestReference...:105-108)
    ctx.close()


def test_system_without_drude_pairs_and_single_particle():
    """Edge cases: no pairs at all (the Drude thermostat has zero degrees of freedom, as in the reference its
    variables go NaN but touch nothing), and a one-particle system."""
    rng = np.random.default_rng(2)
    n = 700
    s = synth.DrudeSystem(mass=rng.uniform(1.0, 30.0, n), pair_drude=np.zeros(0, np.int32), pair_parent=np.zeros(0, np.int32),
                          resid=np.repeat(np.arange(n // 7), 7), positions=rng.uniform(0, 3, (n, 3)),
                          velocities=rng.normal(0, 1.0, (n, 3)))
    it = integ(chains=1)
    ctx = HipContext(s, it, mode="TGNH", precision="double")
    o = make_oracle(s, np.zeros(n, np.int32), 1, "TGNH", it)
    pos_o, vel_o = oracle_run(o, s, 30, x0=ctx.sites())
    ctx.step(30)
    assert rel_err(ctx.getPositions(), pos_o) < 1e-12 and rel_err(ctx.getVelocities(), vel_o) < 1e-10
    ctx.close()
    one = synth.DrudeSystem(mass=np.array([5.0]), pair_drude=np.zeros(0, np.int32), pair_parent=np.zeros(0, np.int32),
                            resid=np.array([0]), positions=np.array([[0.1, 0.2, 0.3]]), velocities=np.array([[1.0, -2.0, 0.5]]))
    it = integ(chains=1, com=False)
    ctx = HipContext(one, it, mode="TGNH", precision="mixed")
    o = make_oracle(one, np.zeros(1, np.int32), 1, "TGNH", it)
    pos_o, vel_o = oracle_run(o, one, 10, x0=ctx.sites())
    ctx.step(10)
    assert rel_err(ctx.getPositions(), pos_o) < 1e-9 and rel_err(ctx.getVelocities(), vel_o) < 1e-9
    ctx.close()


def test_hardwall_too_far_flag():
    s, g, ng = synth.water_box(8)
    it = integ(chains=1, hardwall=0.01)
    ctx = HipContext(s, it, mode="dualNH", precision="double")
    p = s.positions.copy()
    p[s.pair_drude[3]] = p[s.pair_parent[3]] + np.array([0.0, 0.03, 0.0])     # > 2 x wall after the drift
    ctx.setPositions(p)
    ctx.step_begin()
    with pytest.raises(TgnhError, match="too far beyond hard wall") as e:      # Ref :311-312
        ctx.check()
    assert e.value.status == _lib.ERR_HARDWALL
    ctx.close()


@pytest.mark.parametrize("flags", [0, FLAG_RESIDENT_STEP, FLAG_DEFER_SCALE, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP])
def test_single_precision_deviation(flags):
    """float4 state: measured, not gated at 1e-6 (see module docstring).  Bound: 2e-3 on velocities,
    5e-6 on positions after 100 steps of a 1000-water box (measured on MI355X: 1.0e-6 and 3.8e-4), in every pass
    structure (the float instantiations of the deferred launches and of step_kernel are what `bench.py`'s single legs run)."""
    s, g, ng, it, ctx = make("water1000", "TGNH", "single", flags=flags, chains=1, hardwall=0.02)
    o = make_oracle(s, g, ng, "TGNH", it)
    pos_o, vel_o = oracle_run(o, s, 100, x0=ctx.sites())
    ctx.step(100)
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"single precision deviation after 100 steps (flags {flags}): pos {ep:.2e} vel {ev:.2e}")
    assert ep <= 5e-6 and ev <= 2e-3 and ctx.check() == 0
    ctx.close()


# ---------------------------------------------------------------------------
# fused / split / lazy variants are the same integrator
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("sysname", ["mixed", "polymer", "groups12"])
@pytest.mark.parametrize("mode", ["dualNH", "TGNH"])
@pytest.mark.parametrize("flags", [FLAG_DEFER_SCALE])
def test_deferred_rescale_and_kick_match_plain(mode, flags, sysname):
    ref = make(sysname, mode, "double")
    alt = make(sysname, mode, "double", flags=flags)
    ref[4].step(60)
    alt[4].step(60)
    assert rel_err(alt[4].getPositions(), ref[4].getPositions()) < 1e-11
    assert rel_err(alt[4].getVelocities(), ref[4].getVelocities()) < 1e-10     # getVelocities flushes
    # and stepping on after a flush continues identically
    ref[4].step(15)
    alt[4].step(15)
    assert rel_err(alt[4].getVelocities(), ref[4].getVelocities()) < 1e-10
    for which in (0, 1):
        assert np.allclose(alt[4].thermostat_state(which), ref[4].thermostat_state(which), rtol=1e-9, atol=1e-12) \
            or flags == FLAG_DEFER_SCALE        # deferred: the chain already holds the next half step
    ref[4].close(); alt[4].close()


@pytest.mark.parametrize("mode,flags,chains,sysname", [
    ("TGNH", 0, 3, "mixed"), ("TGNH", 0, 1, "mixed"), ("TGNH", 0, 3, "il40"),
    ("TGNH", 0, 3, "polymer"), ("dualNH", 0, 3, "water1000"), ("dualNH", 0, 1, "mixed"),
    # ... and with the begin half's kinetic energies carried over from the last end half (TRUST_STATE_CHANGED).  (Not the deferred
    # structures: between two steps their thermostat variables are half a step ahead of the velocities a query returns, so H read
    # there is off by O(dt) -- 4.6e-3 / 2.3e-3 -- without anything being wrong; they are checked against the plain run instead.)
    ("TGNH", FLAG_TRUST_STATE_CHANGED, 3, "mixed"), ("TGNH", FLAG_TRUST_STATE_CHANGED, 1, "mixed"), ("dualNH", FLAG_TRUST_STATE_CHANGED, 1, "mixed")])
def test_extended_energy_is_conserved(mode, flags, chains, sysname):
    """SURVEY 8c(3), on the HIP path: the Nose-Hoover-chain invariant H (tests/helpers.py) computed from what the
    C ABI hands back (positions, velocities, thermostat state, dof).  The thermostats move > 10 % of the initial
    energy while H holds to O(dt^2); chains = 1 in TGNH mode is the in-kernel chain."""
    from helpers import extended_energy
    worst = []
    for dt in (0.0005, 0.00025):
        s, g, ng, it, ctx = make(sysname, mode, "double", flags=flags, chains=chains, dt=dt)
        x0, normal = ctx.sites(), ctx.topology(0)
        nkt = ctx.dof()[1]
        nkt = nkt[[0, 2]] if mode == "dualNH" else nkt

        def energy():
            return extended_energy(s, normal, ctx.getPositions(), ctx.getVelocities(), x0, nkt,
                                   ctx.thermostat_state(0), ctx.thermostat_state(1), ctx.thermostat_state(3), chains,
                                   synth.KB * 300.0, synth.KB * 1.0, mode)
        h0, ke0, _ = energy()
        dev, th_min = 0.0, 0.0
        for _ in range(10):
            ctx.step(int(round(0.02 / dt)))
            h, _, th = energy()
            dev, th_min = max(dev, abs(h - h0)), min(th_min, th)
        assert th_min < -0.1 * ke0
        worst.append(dev / h0)
        ctx.close()
    print(f"extended energy {mode} flags={flags} C={chains} {sysname}: max |dH|/H0 = {worst[0]:.2e} (dt 0.5 fs), {worst[1]:.2e} (0.25 fs)")
    assert worst[0] < 2e-4 and worst[1] < 5e-5
    assert 2.5 < worst[0] / worst[1] < 7.0


@pytest.mark.parametrize("flags", [FLAG_DEFER_SCALE])
def test_one_link_dualnh_without_drude_chains_all_variants(flags):
    """The C++ default (useDrudeNHChains = false) with one link: the real chain is damped by the Drude thermostat's
    etaDot (Ref :476-481 with numTempGroup = 1).  That coupling runs inside the rescale launch (chain1q_run, one
    shuffle per sub-step); the pass-structure variants and a hipGraph replay must agree with the plain, eager run,
    which test_100_step_parity checks against the oracle."""
    ref = make("mixed", "dualNH", "double", chains=1, drude_chains=False, hardwall=0.02)
    alt = make("mixed", "dualNH", "double", flags=flags, chains=1, drude_chains=False, hardwall=0.02)
    ref[4].step(40)
    alt[4].step(20)
    replay = alt[4].capture_steps(5)
    for _ in range(4):
        replay()
    assert rel_err(alt[4].getPositions(), ref[4].getPositions()) < 1e-11
    assert rel_err(alt[4].getVelocities(), ref[4].getVelocities()) < 1e-10
    ref[4].close(); alt[4].close()


@pytest.mark.parametrize("mode", ["dualNH", "TGNH"])
@pytest.mark.parametrize("precision", ["mixed", "double"])
def test_100_step_parity_deferred_rescale(mode, precision):
    """The bench default (TGNH_FLAG_DEFER_SCALE: the end-of-step rescale is applied by the next step's first
    pass and the chain runs both thermostat half steps back to back on s^2 KE) against the oracle directly."""
    s, g, ng, it, ctx = make("mixed", mode, precision, flags=FLAG_DEFER_SCALE, chains=3, hardwall=0.02)
    o = make_oracle(s, g, ng, mode, it)
    pos_o, vel_o = oracle_run(o, s, 100, x0=ctx.sites())
    ctx.step(100)
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"deferred {mode} {precision}: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL
    ctx.close()


RESIDENT = FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP


@pytest.mark.parametrize("precision", ["mixed", "double"])
@pytest.mark.parametrize("sysname,mode,drude_chains,com,chains", [
    ("mixed", "TGNH", True, True, 1), ("mixed", "dualNH", True, True, 1), ("mixed", "dualNH", False, True, 1),
    ("polymer", "TGNH", True, True, 1), ("water1000", "TGNH", True, False, 1), ("il40", "TGNH", True, True, 1),
    ("ragged6", "TGNH", True, True, 1), ("groups6", "TGNH", True, True, 1),
    # chains of 2-4 links: the whole step still one launch (wstep_kernel with the links in registers)
    ("mixed", "TGNH", True, True, 3), ("mixed", "TGNH", False, True, 2), ("water1000", "TGNH", True, True, 4),
    ("mixed", "dualNH", True, True, 3), ("mixed", "dualNH", False, True, 4), ("groups6", "TGNH", True, True, 2)])
def test_100_step_parity_resident_step(sysname, mode, drude_chains, com, chains, precision):
    """TGNH_FLAG_RESIDENT_STEP: one launch per time step (wstep_kernel / step_kernel: kick + KE sums, the work-groups meet on
    the device, row sum, both chain halves, kick + rescale + kick + drift + hard wall) against the oracle directly."""
    s, g, ng, it, ctx = make(sysname, mode, precision, flags=RESIDENT, chains=chains, drude_chains=drude_chains, com=com, hardwall=0.02)
    o = make_oracle(s, g, ng, mode, it)
    pos_o, vel_o = oracle_run(o, s, 100, x0=ctx.sites())
    # dualNH's COUPLED chain (useDrudeNHChains = false) of 2-4 links is the one exception: wstep_kernel has no room for its fast form,
    # and every kernel of every rank must run the same chain arithmetic -- such a handle quietly steps the DEFER_SCALE way (round 5)
    coupled = mode == "dualNH" and not drude_chains and chains > 1
    assert (ctx.resident_kernel() is None) if coupled else (ctx.resident_work_groups() >= 1)
    ctx.timing(True)
    ctx.step(100)
    ctx.torch.cuda.synchronize()
    ctx.timing(False)
    from openmm_drudenose_amd import _lib
    if coupled:
        assert ctx.timing_read(_lib.KID_STEP)[1] == 0 and ctx.timing_read(_lib.KID_SKD)[1] >= 99       # the deferred launches, chain inside
    else:
        assert ctx.timing_read(_lib.KID_STEP)[1] >= 99 and ctx.timing_read(_lib.KID_CHAIN)[1] == 0    # really one launch per step
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"resident step {sysname} {mode} {precision}: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL and ctx.check() == 0
    ctx.close()


@pytest.mark.parametrize("precision", ["mixed", "double"])
@pytest.mark.parametrize("sysname,mode,drude_chains,com", [
    ("mixed", "TGNH", True, True), ("mixed", "dualNH", True, True), ("mixed", "dualNH", False, True),
    ("polymer", "TGNH", True, True), ("water1000", "TGNH", True, False), ("groups6", "TGNH", True, True)])
def test_100_step_parity_resident_halves_of_the_reference_structure(sysname, mode, drude_chains, com, precision):
    """TGNH_FLAG_RESIDENT_STEP alone: the reference's own pass structure (velocities never lag -- what an OpenMM context
    needs), each thermostat half one launch of step_kernel: KE | chain | rescale+kick+drift+hard wall, and
    kick+KE | chain | rescale.  Against the oracle directly; two launches per step, none of the tile launches."""
    s, g, ng, it, ctx = make(sysname, mode, precision, flags=FLAG_RESIDENT_STEP, chains=1, drude_chains=drude_chains, com=com, hardwall=0.02)
    assert ctx.resident_work_groups() >= 1
    o = make_oracle(s, g, ng, mode, it)
    pos_o, vel_o = oracle_run(o, s, 100, x0=ctx.sites(), record=True)[:2]
    ctx.timing(True)
    for i in range(100):
        ctx.step_begin(); ctx.compute_forces(); ctx.step_end()
    ctx.torch.cuda.synchronize()
    ctx.timing(False)
    nbig = 1 if sysname == "polymer" else 0
    assert ctx.timing_read(_lib.KID_STEP)[1] == 200
    assert all(ctx.timing_read(k)[1] == 0 for k in (_lib.KID_SKD, _lib.KID_KICK_KE, _lib.KID_SCALE, _lib.KID_KE, _lib.KID_CHAIN))
    assert ctx.timing_read(_lib.KID_OTHER)[1] == 200 * nbig            # (the COM of a molecule longer than a tile)
    # velm is the reference's end-of-step state as it stands: no flush involved
    vel = ctx.velm[:, :3].to(ctx.torch.float64).cpu().numpy()
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(vel, vel_o)
    print(f"resident halves {sysname} {mode} {precision}: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL and ctx.check() == 0
    ctx.close()


def test_resident_step_is_the_deferred_integrator():
    """Same trajectory as the multi-launch deferred structure (row sums in a different order: 1e-11), per-step KE and
    scale factors equal to the oracle's, the step kernel really is what ran, queries between steps settle the pending
    half the classic way and stepping goes on, and a hipGraph replay is bitwise the eager run."""
    ref = make("mixed", "TGNH", "double", flags=FLAG_DEFER_SCALE, chains=1, hardwall=0.02)
    alt = make("mixed", "TGNH", "double", flags=RESIDENT, chains=1, hardwall=0.02)
    gra = make("mixed", "TGNH", "double", flags=RESIDENT, chains=1, hardwall=0.02)
    assert 1 <= alt[4].resident_work_groups() <= 8 and ref[4].resident_work_groups() == 0     # the census at create passed
    alt[4].timing(True)
    ref[4].step(30); alt[4].step(30); gra[4].step(30)
    alt[4].torch.cuda.synchronize()
    alt[4].timing(False)
    assert alt[4].timing_read(_lib.KID_STEP)[1] == 29 and alt[4].timing_read(_lib.KID_KICK_KE)[1] == 0
    # a query between steps (thermostat state, velocities) ...
    e_ref, e_alt = ref[4].thermostat_state(1), alt[4].thermostat_state(1)
    assert np.allclose(e_alt, e_ref, rtol=1e-9, atol=1e-13)
    assert rel_err(alt[4].getVelocities(), ref[4].getVelocities()) < 1e-10
    ke_alt = alt[4].last_kinetic_energies()
    assert np.allclose(ke_alt, ref[4].last_kinetic_energies(), rtol=1e-11)
    # ... and on it goes
    ref[4].step(20); alt[4].step(20)
    assert rel_err(alt[4].getPositions(), ref[4].getPositions()) < 1e-11
    assert rel_err(alt[4].getVelocities(), ref[4].getVelocities()) < 1e-10
    assert alt[4].time() == ref[4].time()
    # graph replay: 4 x 5 steps; the eager twin never had a query in between, so compare with a fresh eager run
    eag = make("mixed", "TGNH", "double", flags=RESIDENT, chains=1, hardwall=0.02)
    eag[4].step(50)
    replay = gra[4].capture_steps(5)
    for _ in range(4):
        replay()
    assert np.array_equal(gra[4].getPositions(), eag[4].getPositions())
    assert np.array_equal(gra[4].getVelocities(), eag[4].getVelocities())
    assert np.array_equal(gra[4].thermostat_state(1), eag[4].thermostat_state(1))
    for c in (ref, alt, gra, eag):
        c[4].close()


def test_resident_step_falls_back_where_it_cannot_run():
    """Chains longer than four links (the chain kernel's) and more than 8 temperature groups are not the step kernels': the
    flag is accepted and the handle steps the deferred way."""
    for sysname, chains in (("mixed", 6), ("groups12", 1)):
        ref = make(sysname, "TGNH", "double", flags=FLAG_DEFER_SCALE, chains=chains)
        alt = make(sysname, "TGNH", "double", flags=RESIDENT, chains=chains)
        alt[4].timing(True)
        ref[4].step(20); alt[4].step(20)
        alt[4].torch.cuda.synchronize(); alt[4].timing(False)
        assert alt[4].timing_read(_lib.KID_STEP)[1] == 0
        assert np.array_equal(alt[4].getVelocities(), ref[4].getVelocities())
        ref[4].close(); alt[4].close()


@pytest.mark.parametrize("nranks", [1, 2])
def test_resident_step_with_mailbox_exchange(nranks):
    """step_kernel's sharded form: work-group 0 sums the rows and sends them to every rank's mailbox, every work-group
    waits for all ranks' sums there.  One rank (its own mailbox), and two ranks as two handles on this GPU that share
    its resident work-group slots (tgnh_set_resident_share) on two hardware queues."""
    from openmm_drudenose_amd.system import shard_bounds
    s, g, ng = synth.mixed(400, 30)
    it = integ(chains=1, hardwall=0.02)
    bind_groups(it, g, ng)
    ref = HipContext(s, it, mode="TGNH", precision="double", flags=FLAG_DEFER_SCALE)
    torch = ref.torch
    b = shard_bounds(s, nranks)
    parts, terms, streams = [], [], []
    for r in range(nranks):
        loc, lg = s.slice_molecules(b[r], b[r + 1]), g[b[r]:b[r + 1]]
        itr = integ(chains=1, hardwall=0.02)
        bind_groups(itr, lg, ng)
        parts.append(HipContext(loc, itr, mode="TGNH", precision="double", flags=RESIDENT))
        parts[-1].set_resident_share(nranks)
        terms.append(parts[-1].local_dof_terms())
        streams.append(torch.cuda.Stream(priority=-r))
    total = sum(terms)
    boxes = [c.exchange_create(nranks, r)[1] for r, c in enumerate(parts)]
    for c in parts:
        c.set_global_dof_terms(total)
        c.exchange_attach_pointers(boxes)
        c.timing(True)
    torch.cuda.synchronize()
    for _ in range(40):
        ref.step_begin(); ref.compute_forces(); ref.step_end()
        for c, st in zip(parts, streams):
            with torch.cuda.stream(st):
                c.step_begin(); c.compute_forces(); c.step_end()
    torch.cuda.synchronize()
    for c in parts:
        c.timing(False)
        assert c.timing_read(_lib.KID_STEP)[1] == 39
    # settle every rank's pending half before any rank waits on a query (queries are collective here)
    for c, st in zip(parts, streams):
        with torch.cuda.stream(st):
            _check_ok = c.lib.tgnh_flush(c.h, c._stream())
            assert _check_ok == 0
    torch.cuda.synchronize()
    for c in parts:
        assert c.check() == 0
    pos = np.concatenate([c.getPositions() for c in parts])
    vel = np.concatenate([c.getVelocities() for c in parts])
    assert rel_err(pos, ref.getPositions()) < 1e-12 and rel_err(vel, ref.getVelocities()) < 1e-10
    for c in parts[1:]:
        for which in (0, 1):
            assert np.array_equal(parts[0].thermostat_state(which), c.thermostat_state(which))   # rank-order sums: bitwise
    assert np.allclose(parts[0].thermostat_state(1), ref.thermostat_state(1), rtol=1e-9, atol=1e-13)
    for c in parts:
        c.exchange_detach()
    for c in parts + [ref]:
        c.close()


@pytest.mark.parametrize("mode", ["TGNH", "dualNH"])
@pytest.mark.parametrize("chains", [1, 3])          # 1: the chain runs inside the rescale launch (staged block + commit)
@pytest.mark.parametrize("flags", [0, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP])
def test_graph_replay_matches_eager(flags, chains, mode):
    """hipGraph capture of the step loop (HipContext.capture_steps) replays the very same launches: bitwise equal.
    Five steps per graph on purpose: nothing in the captured launch arguments may alternate between replays."""
    ref = make("mixed", mode, "mixed", flags=flags, hardwall=0.02, chains=chains)
    alt = make("mixed", mode, "mixed", flags=flags, hardwall=0.02, chains=chains)
    ref[4].step(3 + 4 * 5)
    alt[4].step(3)
    replay = alt[4].capture_steps(5)
    for _ in range(4):
        replay()
    assert np.array_equal(alt[4].getPositions(), ref[4].getPositions())
    assert np.array_equal(alt[4].getVelocities(), ref[4].getVelocities())
    assert alt[4].time()[1] == ref[4].time()[1] == 23
    assert np.array_equal(alt[4].thermostat_state(1), ref[4].thermostat_state(1))
    ref[4].close(); alt[4].close()


@pytest.mark.parametrize("flags", [0, FLAG_RESIDENT_STEP, FLAG_TRUST_STATE_CHANGED])
def test_graph_replay_of_the_constrained_step_matches_eager(flags):
    """capture_steps on a system WITH constraints records the loop step() runs for it -- the split entry points around the harness
    SHAKE / velocity stage / virtual sites (Cu :336-406) -- not the unconstrained one (until round 4 it recorded tgnh_run_harness
    whatever the system held: a replay integrated rigid water without its constraints).  Rigid SWM4 water: bitwise equal to eager
    steps, and the bonds are at their lengths."""
    out = []
    for graphed in (False, True):
        s, g, ng = synth.water_box(216, rigid=True)
        it = integ(chains=1, hardwall=0.02, dt=0.0005)
        bind_groups(it, g, ng)
        ctx = HipContext(s, it, mode="TGNH", precision="double", flags=flags)
        assert ctx.constrained
        ctx.step(3)
        if graphed:
            replay = ctx.capture_steps(5)
            for _ in range(4):
                replay()
        else:
            ctx.step(20)
        assert ctx.time()[1] == 23 and ctx.check() == 0
        out.append((ctx.getPositions(), ctx.getVelocities(), ctx.thermostat_state(1)))
        ctx.close()
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    pos = out[1][0].reshape(-1, 5, 3)                         # O, D, H1, H2, M
    assert np.allclose(np.linalg.norm(pos[:, 2] - pos[:, 0], axis=1), 0.09572, rtol=2e-5)
    assert np.allclose(np.linalg.norm(pos[:, 3] - pos[:, 2], axis=1), 0.15139, rtol=2e-5)


def test_split_constraint_path_matches_fused():
    """begin_kick / begin_move / end_kick / end_thermo (the posDelta path around OpenMM's constraint
    call-outs, Cu :356-369, :384-402) with no constraints applied equals the fused step."""
    s, g, ng, it, ref = make("mixed", "TGNH", "double", hardwall=0.0)
    _, _, _, _, alt = make("mixed", "TGNH", "double", hardwall=0.0)
    lib = alt.lib
    for _ in range(30):
        ref.step_begin(); ref.compute_forces(); ref.step_end()
        for fn in (lib.tgnh_step_begin_kick, lib.tgnh_step_begin_move):
            assert fn(alt.h, alt._stream()) == 0
        alt.compute_forces()
        for fn in (lib.tgnh_step_end_kick, lib.tgnh_step_end_thermo):
            assert fn(alt.h, alt._stream()) == 0
    assert rel_err(alt.getPositions(), ref.getPositions()) < 1e-11
    assert rel_err(alt.getVelocities(), ref.getVelocities()) < 1e-9           # v = (dt v)/dt round trip
    ref.close(); alt.close()


CONSTRAINED = {
    "rigid SWM4 water (3 constraints + M virtual site per molecule)": lambda: synth.water_box(343, rigid=True),
    "ionic liquid, X-H bonds constrained": lambda: synth.ionic_liquid(40, constrained=True),
}


@pytest.mark.parametrize("precision", ["mixed", "double"])
@pytest.mark.parametrize("mode", ["dualNH", "TGNH"])
@pytest.mark.parametrize("name", list(CONSTRAINED))
def test_constrained_path_parity(name, mode, precision):
    """The split step (begin_kick / SHAKE on posDelta / begin_move / virtual sites / force / end_kick /
    [TGNH: velocity stage] / end_thermo; Cu :336-406, Ref :253-284) against the oracle's constrained loop.
    The constraint solver is a harness call-out on both sides; its tolerance is set to 1e-10 so that the two
    converged solutions agree far inside the 1e-6 gate."""
    s, g, ng = CONSTRAINED[name]()
    it = integ(chains=2, hardwall=0.02)
    it.setConstraintTolerance(1e-10)
    if mode == "TGNH":
        bind_groups(it, g, ng)
    else:
        g, ng = np.zeros_like(g), 1
    ctx = HipContext(s, it, mode=mode, precision=precision)
    assert ctx.constrained
    o = make_oracle(s, g, ng, mode, it)
    assert np.allclose(ctx.dof()[0], to_internal(o.dof()[0], mode), rtol=1e-14)      # constraints reduce the dof
    pos, vel, x0 = s.positions.copy(), s.velocities.copy(), ctx.sites()
    f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
    o.run_harness_constrained(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, 1e-10, 50)
    ctx.step(50)
    assert ctx.check() == 0                                                          # bit1 = SHAKE not converged
    gp, gv = ctx.getPositions(), ctx.getVelocities()
    ep, ev = rel_err(gp, pos), rel_err(gv, vel)
    print(f"constrained {name} {mode} {precision}: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL
    a, d = s.cluster_atoms, s.cluster_dist
    for k, (i, j) in enumerate(s.PAIRS):
        m = d[:, k] > 0
        if m.any():
            r = np.linalg.norm(gp[a[m, i]] - gp[a[m, j]], axis=1)
            assert np.abs(r - d[m, k]).max() <= 1e-7 * d[m, k].max()                 # fp32 posq + correction holds ~1e-9 nm
    if s.site_atoms is not None:
        sa, w = s.site_atoms, s.site_weights
        expect = sum(w[:, [m]] * gp[sa[:, m + 1]] for m in range(3))
        assert np.abs(gp[sa[:, 0]] - expect).max() <= 1e-7
    ctx.close()


@pytest.mark.parametrize("mode", ["dualNH", "TGNH"])
@pytest.mark.parametrize("name", list(CONSTRAINED))
def test_constrained_path_parity_with_resident_halves(name, mode):
    """The split step around the constraint call-outs with TGNH_FLAG_RESIDENT_STEP: begin_kick = KE | chain |
    rescale+kick+posDelta and end_thermo = KE | chain | rescale, each one launch (Cu :336-360, :394-402)."""
    s, g, ng = CONSTRAINED[name]()
    it = integ(chains=1, hardwall=0.02)
    it.setConstraintTolerance(1e-10)
    if mode == "TGNH":
        bind_groups(it, g, ng)
    else:
        g, ng = np.zeros_like(g), 1
    ctx = HipContext(s, it, mode=mode, precision="mixed", flags=FLAG_RESIDENT_STEP)
    assert ctx.constrained and ctx.resident_work_groups() >= 1
    o = make_oracle(s, g, ng, mode, it)
    pos, vel, x0 = s.positions.copy(), s.velocities.copy(), ctx.sites()
    f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
    o.run_harness_constrained(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, 1e-10, 50)
    ctx.timing(True)
    ctx.step(50)
    ctx.torch.cuda.synchronize()
    ctx.timing(False)
    assert ctx.timing_read(_lib.KID_STEP)[1] == 100 and ctx.timing_read(_lib.KID_CHAIN)[1] == 0
    ep, ev = rel_err(ctx.getPositions(), pos), rel_err(ctx.getVelocities(), vel)
    print(f"constrained, resident halves, {name} {mode}: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL and ctx.check() == 0
    ctx.close()


def test_shake_call_outs_alone():
    """The two constraint stages by themselves against the oracle's (same sweeps, same order)."""
    s, g, ng = synth.ionic_liquid(25, constrained=True)
    it = integ(chains=1)
    bind_groups(it, g, ng)
    ctx = HipContext(s, it, mode="TGNH", precision="double")
    o = make_oracle(s, g, ng, "TGNH", it)
    rng = np.random.default_rng(3)
    delta = rng.normal(0, 2e-3, s.positions.shape)
    ctx.pos_delta[:, :3] = ctx.torch.from_numpy(delta).to(ctx.dev)
    assert ctx.lib.tgnh_harness_shake_positions(ctx.h, 1e-11, ctx._stream()) == 0
    o.shake_positions(s.positions, delta, 1e-11)
    assert rel_err(ctx.pos_delta[:, :3].cpu().numpy(), delta) < 1e-9
    vel = s.velocities.copy()
    assert ctx.lib.tgnh_harness_shake_velocities(ctx.h, 1e-11, ctx._stream()) == 0
    o.shake_velocities(s.positions, vel, 1e-11)
    assert rel_err(ctx.getVelocities(), vel) < 1e-9
    # along every constrained bond the relative velocity is gone
    a, d = s.cluster_atoms, s.cluster_dist
    gv = ctx.getVelocities()
    for k, (i, j) in enumerate(s.PAIRS):
        m = d[:, k] > 0
        if m.any():
            r = s.positions[a[m, i]] - s.positions[a[m, j]]
            assert np.abs(((gv[a[m, i]] - gv[a[m, j]]) * r).sum(1)).max() < 1e-9
    ctx.close()


def test_particle_sharded_hip_path_on_one_gpu():
    """The sharded schedule on the real kernels: two handles on this one GPU own the two halves of the molecules
    (system.shard_bounds), degrees of freedom are summed over the 'ranks', and the all-reduce hook of each handle
    adds the other handle's kinetic-energy sums (what RCCL does across GPUs).  The trajectory must be the
    unsharded one.  (Real multi-GPU runs need the driver's 8-GPU node; the collective itself is torch.distributed's.)"""
    from openmm_drudenose_amd.system import shard_bounds
    s, g, ng = synth.mixed(400, 30)
    it = integ(chains=1, hardwall=0.02)
    bind_groups(it, g, ng)
    ref = HipContext(s, it, mode="TGNH", precision="double")
    b = shard_bounds(s, 2)
    parts, terms = [], []
    for r in range(2):
        loc, lg = s.slice_molecules(b[r], b[r + 1]), g[b[r]:b[r + 1]]
        itr = integ(chains=1, hardwall=0.02)
        bind_groups(itr, lg, ng)
        ctx = HipContext(loc, itr, mode="TGNH", precision="double")
        parts.append(ctx)
        terms.append(ctx.local_dof_terms())
    total = terms[0] + terms[1]
    assert np.allclose(total, ref.local_dof_terms(), rtol=1e-13)
    peer = [None, None]                      # peer[r]: the other shard's KE sums for the phase in flight (device tensor)
    for r, ctx in enumerate(parts):
        ctx.set_global_dof_terms(total)
        ctx.set_allreduce(lambda t, r=r: t.add_(peer[r]) if peer[r] is not None else None)
    assert np.allclose(parts[0].dof()[0], ref.dof()[0], rtol=1e-13)
    torch = ref.torch

    def exchange():
        """both shards reduce their own rows at the current phase; each then knows what the other would contribute"""
        peer[0] = peer[1] = None
        ke = [torch.from_numpy(c.compute_kinetic_energies()).to(c.dev) for c in parts]
        peer[0], peer[1] = ke[1], ke[0]

    lib = ref.lib
    # (the shards step through the split entry points, the glue's sequence around constraint call-outs: the second exchange
    # falls between the half kick and the thermostat half.  That sequence has an odd number of sweeps, like every pass
    # structure, so a step starts in the direction of its number and the query before it sums in the order the step will.)
    for _ in range(40):
        ref.step_begin(); ref.compute_forces(); ref.step_end()
        exchange()
        for c in parts:
            assert lib.tgnh_step_begin_kick(c.h, c._stream()) == 0     # KE (own rows) -> hook adds the peer's -> chain -> rescale, kick
            assert lib.tgnh_step_begin_move(c.h, c._stream()) == 0     # drift, hard wall
        for c in parts:
            c.compute_forces()
        for c in parts:
            assert lib.tgnh_step_end_kick(c.h, c._stream()) == 0
        exchange()
        for c in parts:
            assert lib.tgnh_step_end_thermo(c.h, c._stream()) == 0
    pos = np.concatenate([c.getPositions() for c in parts])
    vel = np.concatenate([c.getVelocities() for c in parts])
    assert rel_err(pos, ref.getPositions()) < 1e-12 and rel_err(vel, ref.getVelocities()) < 1e-10
    assert np.allclose(parts[0].thermostat_state(1), parts[1].thermostat_state(1), rtol=0, atol=0)   # replicated chain: bitwise
    assert np.allclose(parts[0].thermostat_state(1), ref.thermostat_state(1), rtol=1e-9, atol=1e-13)
    for c in parts + [ref]:
        c.close()


@pytest.mark.parametrize("variant,chains,nranks,sysname,mode", [
    (FLAG_DEFER_SCALE, 1, 2, "mixed", "TGNH"), (0, 1, 2, "mixed", "TGNH"), (FLAG_DEFER_SCALE, 3, 2, "mixed", "TGNH"),
    (0, 3, 2, "mixed", "TGNH"),
    (FLAG_DEFER_SCALE, 1, 2, "groups32", "TGNH"),   # 2 ranks x 34 thermostats = 68 cells: more than one wavefront's worth
    (FLAG_DEFER_SCALE, 1, 2, "mixed", "dualNH"), (0, 3, 2, "mixed", "dualNH"),
    (FLAG_DEFER_SCALE, 1, 2, "ragged6", "TGNH")])    # shards cut at ragged molecule boundaries
def test_particle_sharded_mailbox_exchange_on_one_gpu(variant, chains, nranks, sysname, mode):
    """The mailbox exchange (tgnh_exchange_*: the KE all-reduce done by the integrator's own kernels with stores into
    every peer's mailbox) with the 'ranks' as handles on this one GPU, each on its own stream, mailboxes attached by
    pointer: every rank's rescale launch spins until all ranks' sum launches have delivered.  Trajectory = the
    unsharded one; thermostats bit-identical on all ranks.  chains = 1 waits inside the rescale launch's prologue,
    chains = 3 inside the chain launch.  (Two ranks, on a normal- and a high-priority stream: streams of one process
    share a few hardware queues, and a rank spinning in front of its peer's sum launch in the same queue could only
    time out; one process per GPU has no such coupling.)"""
    from openmm_drudenose_amd.system import shard_bounds
    s, g, ng = synth.mixed(400, 30) if sysname == "mixed" else SYSTEMS[sysname]()
    if mode == "dualNH":
        g, ng = np.zeros_like(g), 1
    it = integ(chains=chains, hardwall=0.02)
    if mode == "TGNH":
        bind_groups(it, g, ng)
    ref = HipContext(s, it, mode=mode, precision="double", flags=variant)
    torch = ref.torch
    b = shard_bounds(s, nranks)
    parts, terms, streams = [], [], []
    for r in range(nranks):
        loc, lg = s.slice_molecules(b[r], b[r + 1]), g[b[r]:b[r + 1]]
        itr = integ(chains=chains, hardwall=0.02)
        if mode == "TGNH":
            bind_groups(itr, lg, ng)
        parts.append(HipContext(loc, itr, mode=mode, precision="double", flags=variant))
        terms.append(parts[-1].local_dof_terms())
        streams.append(torch.cuda.Stream(priority=-r))       # normal / high priority: two different hardware queues
    total = sum(terms)
    boxes = [c.exchange_create(nranks, r)[1] for r, c in enumerate(parts)]
    for c in parts:
        c.set_global_dof_terms(total)
        c.exchange_attach_pointers(boxes)
    torch.cuda.synchronize()
    for _ in range(40):
        ref.step_begin(); ref.compute_forces(); ref.step_end()
        for c, st in zip(parts, streams):
            with torch.cuda.stream(st):
                c.step_begin(); c.compute_forces(); c.step_end()
    torch.cuda.synchronize()
    for c in parts:
        assert c.check() & 4 == 0                  # no exchange time-out
    pos = np.concatenate([c.getPositions() for c in parts])
    vel = np.concatenate([c.getVelocities() for c in parts])
    assert rel_err(pos, ref.getPositions()) < 1e-12 and rel_err(vel, ref.getVelocities()) < 1e-10
    for c in parts[1:]:
        for which in (0, 1):
            assert np.array_equal(parts[0].thermostat_state(which), c.thermostat_state(which))   # rank-order sums: bitwise
    assert np.allclose(parts[0].thermostat_state(1), ref.thermostat_state(1), rtol=1e-9, atol=1e-13)
    for c in parts + [ref]:
        c.close()


@pytest.mark.parametrize("variant", [FLAG_DEFER_SCALE, 0])
def test_mailbox_ranks_stay_bit_identical_when_one_rank_is_queried(variant):
    """The determinism contract of include/drude_tgnh.h: the ranks of one run hold bit-identical thermostats whatever each of
    them is asked in between.  Two ranks (handles on this GPU, mailboxes by pointer); rank 0 alone serves a query between
    steps every few steps -- the thermostat state (which makes it run the chain in the standalone kernel where rank 1 runs it
    inside its next rescale launch), the cached and the plain kinetic energy, the velocities (a flush: extra sweeps on rank 0
    only).  Every rank adds the same world x NT numbers in rank order and runs the same chain arithmetic, so nothing of that
    may show: thermostats equal bit for bit over the ranks at the end, trajectory the unsharded one."""
    from openmm_drudenose_amd.system import shard_bounds
    s, g, ng = synth.mixed(400, 30)
    it = integ(chains=1, hardwall=0.02)
    bind_groups(it, g, ng)
    ref = HipContext(s, it, mode="TGNH", precision="double", flags=variant)
    torch = ref.torch
    b = shard_bounds(s, 2)
    parts, terms, streams = [], [], []
    for r in range(2):
        loc, lg = s.slice_molecules(b[r], b[r + 1]), g[b[r]:b[r + 1]]
        itr = integ(chains=1, hardwall=0.02)
        bind_groups(itr, lg, ng)
        parts.append(HipContext(loc, itr, mode="TGNH", precision="double", flags=variant))
        terms.append(parts[-1].local_dof_terms())
        streams.append(torch.cuda.Stream(priority=-r))
    total = sum(terms)
    boxes = [c.exchange_create(2, r)[1] for r, c in enumerate(parts)]
    for c in parts:
        c.set_global_dof_terms(total)
        c.exchange_attach_pointers(boxes)
    torch.cuda.synchronize()
    asked = 0
    for step in range(60):
        ref.step_begin(); ref.compute_forces(); ref.step_end()
        for c, st in zip(parts, streams):
            with torch.cuda.stream(st):
                c.step_begin(); c.compute_forces(); c.step_end()
        if step % 4 == 3:
            with torch.cuda.stream(streams[0]):
                c0 = parts[0]
                which = (step // 4) % 4
                if which == 0:
                    c0.thermostat_state(1)
                elif which == 1:
                    c0.last_kinetic_energies(); c0.kinetic_energy()
                elif which == 2:
                    c0.getVelocities()
                else:
                    c0.ke_sum_valid = False; c0.kinetic_energy(); c0.ke_sum_valid = True
                asked += 1
    torch.cuda.synchronize()
    assert asked >= 12
    for c in parts:
        assert c.check() & 4 == 0
    pos = np.concatenate([c.getPositions() for c in parts])
    vel = np.concatenate([c.getVelocities() for c in parts])
    assert rel_err(pos, ref.getPositions()) < 1e-12 and rel_err(vel, ref.getVelocities()) < 1e-10
    for which in (0, 1, 2):
        assert np.array_equal(parts[0].thermostat_state(which), parts[1].thermostat_state(which))
    for c in parts + [ref]:
        c.close()


def test_mailbox_exchange_times_out_instead_of_hanging():
    """A peer that never sends: the wait is bounded, status bit 2 is raised, later waits return at once."""
    import time
    s, g, ng = synth.water_box(27)
    it = integ(chains=1)
    bind_groups(it, g, ng)
    a, b = HipContext(s, it, mode="TGNH", precision="double"), HipContext(s, integ(chains=1), mode="TGNH", precision="double")
    boxes = [a.exchange_create(2, 0)[1], b.exchange_create(2, 1)[1]]
    a.exchange_attach_pointers(boxes)
    t0 = time.time()
    a.step_begin()                                   # rank 1 never steps
    a.torch.cuda.synchronize()
    first = time.time() - t0
    t0 = time.time()
    a.compute_forces(); a.step_end(); a.step_begin()      # the host has not looked at the status word yet
    a.torch.cuda.synchronize()
    assert time.time() - t0 < 0.5 * first + 0.05     # the latch: no second long wait
    assert first < 30.0
    assert a.status_flags() & 4
    # ... and once the host has seen it, the failure is sticky: nothing integrates on a partial kinetic-energy sum
    for call in (a.check, a.step_end, a.step_begin, lambda: a.thermostat_state(1), a.last_scale_factors, a.kinetic_energy):
        with pytest.raises(TgnhError) as e:
            call()
        assert e.value.status == _lib.ERR_STATE and "timed out" in str(e.value)
    a.exchange_detach()
    a.close(); b.close()


def test_exchange_failure_is_noticed_without_any_query():
    """A caller that only ever steps (an OpenMM run between reporter intervals): the library reads the status word
    back behind every 64th step, so the failure surfaces as TGNH_ERR_STATE from a later step call by itself."""
    s, g, ng = synth.water_box(27)
    it = integ(chains=1)
    bind_groups(it, g, ng)
    a, b = HipContext(s, it, mode="TGNH", precision="double"), HipContext(s, integ(chains=1), mode="TGNH", precision="double")
    boxes = [a.exchange_create(2, 0)[1], b.exchange_create(2, 1)[1]]
    a.exchange_attach_pointers(boxes)
    raised_at = None
    for i in range(200):
        try:
            a.step_begin(); a.compute_forces(); a.step_end()         # rank 1 never steps
        except TgnhError as e:
            assert e.status == _lib.ERR_STATE and "timed out" in str(e)
            raised_at = i
            break
        if i % 50 == 49:
            a.torch.cuda.synchronize()               # let the read-back land (the host runs ahead of the device)
    assert raised_at is not None and 64 <= raised_at <= 150, raised_at
    a.close(); b.close()


def test_against_committed_regression_vectors():
    """The HIP path against tests/golden/oracle_regression.npz (the oracle's frozen outputs; data only, so this check
    does not need the oracle at run time)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_regression_vectors", os.path.join(path, "make_regression_vectors.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    frozen = np.load(os.path.join(path, "oracle_regression.npz"))
    for name, (build, omode, chains, drude_chains, com, hw, steps) in gen.CASES.items():
        mode = "TGNH" if omode == 1 else "dualNH"
        s, g, ng = build()
        it = integ(chains=chains, drude_chains=drude_chains, com=com, hardwall=hw)
        if mode == "TGNH":
            bind_groups(it, g, ng)
        ctx = HipContext(s, it, mode=mode, precision="double")     # double: tether sites stored exactly, as the oracle had them
        kes, scs = to_internal(frozen[f"{name}/ke"], mode), to_internal(frozen[f"{name}/scale"], mode)
        for i in range(steps):
            ctx.step_begin()
            assert np.allclose(ctx.last_kinetic_energies(), kes[2 * i], rtol=TOL_KE, atol=TOL_KE * np.abs(kes[2 * i]).max())
            ctx.compute_forces()
            ctx.step_end()
            assert np.allclose(ctx.last_kinetic_energies(), kes[2 * i + 1], rtol=TOL_KE, atol=TOL_KE * np.abs(kes[2 * i + 1]).max())
            m = np.ones(kes.shape[1], bool)
            if mode == "dualNH":
                m[1] = False
            assert np.abs(ctx.last_scale_factors()[m] - scs[2 * i + 1][m]).max() <= TOL_KE
        assert rel_err(ctx.getPositions()[:64], frozen[f"{name}/pos64"]) <= TOL
        assert rel_err(ctx.getVelocities()[:64], frozen[f"{name}/vel64"]) <= TOL
        assert np.allclose(ctx.thermostat_state(1), frozen[f"{name}/etaDot"], rtol=1e-6, atol=1e-9)
        ctx.close()


@pytest.mark.parametrize("mode,chains,flags", [("TGNH", 3, 0), ("TGNH", 1, 0), ("dualNH", 1, 0), ("dualNH", 3, 0)])
def test_thermostat_state_checkpoint_roundtrip(mode, chains, flags):
    """Save positions, velocities and the thermostat variables, restore them into a fresh context, continue: bitwise the
    same trajectory.  chains = 1: the saved state has to come out of (and go back into) the in-kernel chain's staged
    block."""
    s, g, ng, it, a = make("il40", mode, "double", chains=chains, flags=flags)
    a.step(20)
    saved = [a.thermostat_state(w) for w in range(3)]
    pos, vel = a.getPositions(), a.getVelocities()
    a.step(20)
    _, _, _, _, b = make("il40", mode, "double", chains=chains, flags=flags)
    b.setPositions(pos); b.setVelocities(vel); b.compute_forces()
    for w in range(3):
        b.set_thermostat_state(w, saved[w])
    b.step(20)
    assert np.array_equal(b.getVelocities(), a.getVelocities())
    assert np.array_equal(b.getPositions(), a.getPositions())
    a.close(); b.close()


@pytest.mark.parametrize("mode,chains", [("TGNH", 1), ("TGNH", 3), ("dualNH", 1), ("dualNH", 3)])
def test_checkpoint_through_serialization_restores_masses_and_clock(mode, chains):
    """save_thermostat / load_thermostat: eta, etaDot, etaDotDot, etaMass, time and stepCount.  The thermostat masses
    are changed by hand before the checkpoint (which = 3): with the in-kernel chain (chains = 1) a value written to the
    live block alone would be overwritten by the next commit of the staged block one step later."""
    from openmm_drudenose_amd.serialization import save_thermostat, load_thermostat
    s, g, ng, it, a = make("il40", mode, "double", chains=chains)
    a.step(5)
    q = a.thermostat_state(3)
    q2 = q * np.linspace(1.3, 1.7, len(q))
    a.set_thermostat_state(3, q2)
    a.step(7)
    assert np.array_equal(a.thermostat_state(3), q2)                 # not reverted by a later commit
    state = save_thermostat(a)
    pos, vel = a.getPositions(), a.getVelocities()
    assert state["stepCount"] == 12 and state["time"] == pytest.approx(12 * 0.001)
    a.step(9)
    _, _, _, _, b = make("il40", mode, "double", chains=chains)
    b.setPositions(pos); b.setVelocities(vel); b.compute_forces()
    load_thermostat(b, state)
    assert b.time() == (state["time"], 12)
    b.step(9)
    assert b.time()[1] == a.time()[1] == 21 and b.time()[0] == pytest.approx(a.time()[0], rel=1e-12)
    assert np.array_equal(b.getVelocities(), a.getVelocities())
    assert np.array_equal(b.getPositions(), a.getPositions())
    assert np.array_equal(b.thermostat_state(3), q2)
    # the changed masses really took part: a context with the original masses ends elsewhere
    _, _, _, _, c = make("il40", mode, "double", chains=chains)
    c.step(21)
    assert not np.array_equal(c.getVelocities(), a.getVelocities())
    a.close(); b.close(); c.close()


def test_bridge_identity_on_the_gpu():
    """with G = 1, no COM group, useDrudeNHChains = true, no CMMotionRemover, no constraints the two
    semantic modes are the same integrator.  TGNH-HIP vs dualNH-HIP vs the dualNH oracle -- the one cross-check
    between the two independently restated modes, here on the device."""
    for chains in (1, 3):
        s, g, ng = synth.water_box(27)
        g = np.zeros_like(g)
        it_t, it_d = integ(chains=chains, com=False, hardwall=0.02), integ(chains=chains, com=False, hardwall=0.02)
        bind_groups(it_t, g, 1)
        t = HipContext(s, it_t, mode="TGNH", precision="double")
        d = HipContext(s, it_d, mode="dualNH", precision="double")
        o = make_oracle(s, g, 1, "dualNH", it_d)
        pos_o, vel_o = oracle_run(o, s, 100, x0=d.sites())
        t.step(100); d.step(100)
        for name, c in (("TGNH-HIP", t), ("dualNH-HIP", d)):
            ep, ev = rel_err(c.getPositions(), pos_o), rel_err(c.getVelocities(), vel_o)
            print(f"bridge chains={chains} {name} vs dualNH oracle: pos {ep:.2e} vel {ev:.2e}")
            assert ep < 1e-9 and ev < 1e-9
        assert rel_err(t.getVelocities(), d.getVelocities()) < 1e-9
        # thermostats: TGNH keeps [group 0, COM (inert), Drude], dualNH [real, -, Drude]
        kt, kd = t.last_kinetic_energies(), d.last_kinetic_energies()
        assert kt[0] == pytest.approx(kd[0], rel=1e-9) and kt[2] == pytest.approx(kd[2], rel=1e-9) and kt[1] == 0.0
        st, sd = t.last_scale_factors(), d.last_scale_factors()
        assert st[0] == pytest.approx(sd[0], rel=1e-11) and st[2] == pytest.approx(sd[2], rel=1e-11)
        t.close(); d.close()


@pytest.mark.parametrize("tiles", ["wave", "lds"])
@pytest.mark.parametrize("flags", [0, RESIDENT])
def test_dualnh_ignores_temperature_groups_on_the_gpu(flags, tiles):
    """dualNH with the integrator's four temperature groups handed over = dualNH without any, bit for bit, and both on the
    dualNH oracle (which has no groups to read): the Reference platform's two thermostats do not know about groups."""
    s, g, ng = SYSTEMS["mixed"]()
    wf = FLAG_WAVE_TILES if tiles == "wave" else 0
    it_g, it_0 = integ(chains=3, hardwall=0.02), integ(chains=3, hardwall=0.02)
    bind_groups(it_g, g, ng)
    a = HipContext(s, it_g, mode="dualNH", precision="double", flags=flags | wf)
    b = HipContext(s, it_0, mode="dualNH", precision="double", flags=flags | wf)
    o = make_oracle(s, np.zeros_like(g), 1, "dualNH", it_0)
    pos_o, vel_o = oracle_run(o, s, 50, x0=b.sites())
    a.step(50); b.step(50)
    assert np.array_equal(a.getVelocities(), b.getVelocities()) and np.array_equal(a.getPositions(), b.getPositions())
    assert np.array_equal(a.last_kinetic_energies(), b.last_kinetic_energies())
    assert rel_err(a.getPositions(), pos_o) <= TOL and rel_err(a.getVelocities(), vel_o) <= TOL
    a.close(); b.close()


def test_10000_steps_all_pass_structures_track_the_oracle():
    """Ten picoseconds, 1000 waters, double precision: the reference's pass structure, the deferred one and the one-launch
    step all stay on the oracle's trajectory (the thermostats swing the group temperature between 30 and 1200 K on this
    harmonic workload -- every branch of the on-device chain arithmetic gets exercised, not only its first 100 steps)."""
    s, g, ng = SYSTEMS["water1000"]()
    ctxs = {}
    for name, flags in (("plain", 0), ("plain, resident halves", FLAG_RESIDENT_STEP), ("deferred", FLAG_DEFER_SCALE), ("resident", RESIDENT)):
        it = integ(chains=1, hardwall=0.02)
        bind_groups(it, g, ng)
        ctxs[name] = HipContext(s, it, mode="TGNH", precision="double", flags=flags)
    it = integ(chains=1, hardwall=0.02)
    o = make_oracle(s, g, ng, "TGNH", it)
    pos, vel, x0 = s.positions.copy(), s.velocities.copy(), ctxs["plain"].sites()
    f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
    worst = {n: 0.0 for n in ctxs}
    for _ in range(10):
        o.run_harness(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, 1000)
        for n, c in ctxs.items():
            c.step(1000)
            worst[n] = max(worst[n], rel_err(c.getVelocities(), vel), rel_err(c.getPositions(), pos))
    print("10000 steps, max rel err over the run:", {n: f"{w:.1e}" for n, w in worst.items()})
    for n, c in ctxs.items():
        assert worst[n] <= 1e-7 and c.check() == 0, (n, worst[n])
        c.close()


@pytest.mark.parametrize("flags", [0, FLAG_DEFER_SCALE])
def test_1000_step_mixed_precision_gate(flags):
    """north_star's tolerance is stated over 100 steps; mixed precision keeps positions as float + float correction,
    whose round-off accumulates with the step count.  1 000 steps against the oracle, same 1e-6 gate; the figures are
    printed for DESIGN.md."""
    s, g, ng, it, ctx = make("mixed", "TGNH", "mixed", flags=flags, chains=1, hardwall=0.02)
    o = make_oracle(s, g, ng, "TGNH", it)
    pos_o, vel_o = oracle_run(o, s, 1000, x0=ctx.sites())
    ctx.step(1000)
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"1000 steps mixed flags={flags}: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL and ctx.check() == 0
    ctx.close()


@pytest.mark.parametrize("precision", ["single", "mixed", "double"])
@pytest.mark.parametrize("sysname", ["water27", "water1000", "mixed", "il40", "polymer", "ragged6", "pair+normal+massless"])
def test_packed_sites_give_the_same_forces(sysname, precision):
    """The harness force reads the sites in packed form (tgnh_harness_pack_sites: one byte per slot, the tethered slots' sites
    only) -- bit for bit the forces of the kernel that reads x0 and the meta word, partners near and far."""
    s, g, ng, it, ctx = make(sysname, "TGNH", precision)
    rng = np.random.default_rng(5)
    ctx.setPositions(s.positions + rng.normal(0, 0.01, s.positions.shape))
    assert ctx.sites_packed
    ctx.compute_forces()
    packed = ctx.force.clone()
    ctx.force.zero_()
    ctx.unpacked_sites = True
    ctx.compute_forces()
    assert ctx.torch.equal(packed, ctx.force) and int(packed.abs().max()) > 0
    # other sites: packed again, the forces follow (and differ from the first set's)
    ctx.unpacked_sites = False
    ctx.set_sites(s.positions + rng.normal(0, 0.02, s.positions.shape))
    ctx.compute_forces()
    repacked = ctx.force.clone()
    ctx.unpacked_sites = True
    ctx.compute_forces()
    assert ctx.torch.equal(repacked, ctx.force)
    if (s.mass > 0).sum() > len(s.pair_drude):               # (some slot is tethered)
        assert not ctx.torch.equal(repacked, packed)
    meta = ctx.topology(8).view(np.uint32)
    off = ((meta >> 10) & 2047).astype(np.int64) - 1024
    print(sysname, precision, "partners more than 15 slots away:", int(np.sum((np.abs(off) > 15) & ((meta & 3) != 0))))
    ctx.close()


@pytest.mark.parametrize("precision", ["single", "mixed", "double"])
@pytest.mark.parametrize("n_mol", [27, 1000, 1500])
def test_lattice_sites_give_the_same_forces(n_mol, precision):
    """A box that repeats one molecule on a lattice (synth.water_box says so: DrudeSystem.lattice) needs no per-slot site data at
    all: tgnh_harness_pack_sites checks the hint slot by slot and the force kernel then forms flag byte and site from the slot
    index (force_lattice_kernel: positions in, forces out, nothing else).  Bit for bit the forces of the packed form -- a cube of
    molecules and a box that does not fill its cube (1 500 of 12^3), a shard that starts in the middle of the box -- and a box
    whose sites do not obey the formula falls back to the packed form by itself."""
    s, g, ng = synth.water_box(n_mol)
    rng = np.random.default_rng(9)
    moved = s.positions + rng.normal(0, 0.01, s.positions.shape)
    forces = {}
    for lat in (True, False):
        it = integ(chains=1)
        ctx = HipContext(s, it, mode="TGNH", precision=precision, lattice_sites=lat)
        assert ctx.sites_kind() == ("lattice" if lat else "packed")
        ctx.setPositions(moved)
        ctx.compute_forces()
        forces[lat] = ctx.force.clone()
        if lat:                                              # other sites (not the lattice's): the hint no longer holds
            ctx.set_sites(moved)
            assert ctx.sites_kind() == "packed"
        ctx.close()
    assert forces[True].abs().max() > 0 and ctx.torch.equal(forces[True], forces[False])
    if n_mol >= 1000:                                        # a shard: molecules 300 .. 800 of the box
        sh = s.slice_molecules(1500, 4000)
        assert sh.lattice is not None and sh.lattice[4] == 300
        out = {}
        for lat in (True, False):
            ctx = HipContext(sh, integ(chains=1), mode="TGNH", precision=precision, lattice_sites=lat)
            assert ctx.sites_kind() == ("lattice" if lat else "packed")
            ctx.setPositions(moved[1500:4000])
            ctx.compute_forces()
            out[lat] = ctx.force.clone()
            ctx.close()
        assert ctx.torch.equal(out[True], out[False])
        n = sh.num_particles
        pad = forces[True].numel() // 3
        whole = forces[True].view(3, pad)[:, 1500:4000]
        assert ctx.torch.equal(out[True].view(3, -1)[:, :n], whole)      # ... the same forces as inside the whole box


@pytest.mark.parametrize("flags", [0, FLAG_RESIDENT_STEP])
@pytest.mark.parametrize("mode", ["dualNH", "TGNH"])
def test_kinetic_energy_query(mode, flags):
    s, g, ng, it, ctx = make("mixed", mode, "double", chains=3 if flags == 0 else 1, flags=flags)
    o = make_oracle(s, g, ng, mode, it)
    f = o.harness_force(s.positions, ctx.sites(), synth.K_DRUDE, synth.K_TETHER)
    assert ctx.kinetic_energy() == pytest.approx(o.kinetic_energy_query(s.velocities, f, False), rel=1e-9)
    pos_o, vel_o = oracle_run(o, s, 3, x0=ctx.sites())
    ctx.step(3)
    fo = o.harness_force(pos_o, ctx.sites(), synth.K_DRUDE, synth.K_TETHER)
    assert ctx.kinetic_energy() == pytest.approx(o.kinetic_energy_query(vel_o, fo, True), rel=1e-7)
    ctx.close()


@pytest.mark.parametrize("chains,flags", [(1, 0), (3, 0), (1, FLAG_RESIDENT_STEP)])
def test_setters_take_effect_mid_run(chains, flags):
    """Step size and drudeStepsPerRealStep are re-read every step (Cu :292, :437)."""
    s, g, ng, it, ctx = make("water27", "TGNH", "double", chains=chains, flags=flags)
    o = make_oracle(s, g, ng, "TGNH", it)
    pos, vel, x0 = s.positions.copy(), s.velocities.copy(), ctx.sites()
    f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
    o.run_harness(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, 10)
    ctx.step(10)
    it.setStepSize(0.0005); it.setDrudeStepsPerRealStep(5)
    o.set_step_size(0.0005); o.set_drude_steps(5)
    o.run_harness(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, 10)
    ctx.step(10)
    assert rel_err(ctx.getVelocities(), vel) < 1e-9 and rel_err(ctx.getPositions(), pos) < 1e-11
    ctx.close()


# ---------------------------------------------------------------------------
# BASELINE.json configs at their stated sizes (parity-test cases)
# ---------------------------------------------------------------------------
CONFIGS = {
    # name: (builder, mode, hardwall, steps)   -- steps are fewer where the single-thread oracle is slow
    "C2 SWM4 32k atoms, 1 group": (lambda: synth.water_box(6400), 0.0, 100),
    "C3 ionic liquid 100k atoms, 2 groups": (lambda: synth.ionic_liquid(2222), 0.0, 100),
    "C4 mixed 500k atoms, 4 groups, hard wall": (lambda: synth.mixed(60000, 4444), 0.02, 50),
    "C5 SWM4 2M atoms": (lambda: synth.water_box(400000), 0.0, 20),
}


def test_config3_with_shake_at_its_stated_size():
    """BASELINE config 3 as worded: [BMIM][BF4]-like ionic liquid, 100 k atoms, 2 temperature groups **+ SHAKE** (the 15 X-H
    bonds per cation): the split path with the harness SHAKE / velocity stage as call-outs, against the oracle's constrained
    loop (solver tolerance 1e-10 on both sides)."""
    s, g, ng = synth.ionic_liquid(2222, constrained=True)
    it = integ(chains=1, hardwall=0.0)
    it.setConstraintTolerance(1e-10)
    bind_groups(it, g, ng)
    ctx = HipContext(s, it, mode="TGNH", precision="mixed")
    assert ctx.constrained and len(s.constraints) == 15 * 2222
    o = make_oracle(s, g, ng, "TGNH", it)
    pos, vel, x0 = s.positions.copy(), s.velocities.copy(), ctx.sites()
    f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
    o.run_harness_constrained(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, 1e-10, 30)
    ctx.step(30)
    ep, ev = rel_err(ctx.getPositions(), pos), rel_err(ctx.getVelocities(), vel)
    print(f"C3 with SHAKE: N={s.num_particles} constraints={len(s.constraints)} pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL and ctx.check() == 0
    ctx.close()


@pytest.mark.parametrize("name,nranks,nsteps", [("C4 mixed 500k atoms, 4 groups, hard wall", 4, 10), ("C5 SWM4 2M atoms", 8, 6)])
def test_configs_4_and_5_particle_sharded_at_their_stated_size(name, nranks, nsteps):
    """BASELINE configs 4 and 5 as worded -- particle-sharded 4x and 8x -- as far as one GPU goes: the shards are handles on
    this GPU, each handle's all-reduce hook adds the other shards' kinetic-energy sums (what RCCL does across GPUs), dof
    terms summed over the shards.  Against the unsharded HIP run of the same system (which test_config_sizes_parity holds
    against the oracle): positions 1e-12, velocities 1e-10, thermostats bitwise over the shards."""
    from openmm_drudenose_amd.system import shard_bounds
    build, hardwall, _ = CONFIGS[name]
    s, g, ng = build()
    it = integ(chains=1, hardwall=hardwall)
    bind_groups(it, g, ng)
    ref = HipContext(s, it, mode="TGNH", precision="mixed")
    torch = ref.torch
    b = shard_bounds(s, nranks)
    parts, terms = [], []
    for r in range(nranks):
        loc, lg = s.slice_molecules(b[r], b[r + 1]), g[b[r]:b[r + 1]]
        itr = integ(chains=1, hardwall=hardwall)
        bind_groups(itr, lg, ng)
        parts.append(HipContext(loc, itr, mode="TGNH", precision="mixed"))
        terms.append(parts[-1].local_dof_terms())
    total = sum(terms)
    assert np.allclose(total, ref.local_dof_terms(), rtol=1e-11)       # red_g is a sum over 10^5..10^6 particles
    ref.set_global_dof_terms(total)                                    # the very same N kT and Q on both sides
    reduced = [None]                        # what the all-reduce delivers: the shards' sums added in rank order, same bits everywhere
    for r, ctx in enumerate(parts):
        ctx.set_global_dof_terms(total)
        ctx.set_allreduce(lambda t: t.copy_(reduced[0]) if reduced[0] is not None else None)

    def exchange():
        reduced[0] = None
        ke = [torch.from_numpy(c.compute_kinetic_energies()).to(c.dev) for c in parts]
        tot = ke[0].clone()
        for k in ke[1:]:
            tot += k
        reduced[0] = tot
    lib = ref.lib
    for _ in range(nsteps):
        ref.step_begin(); ref.compute_forces(); ref.step_end()
        exchange()
        for c in parts:
            c.step_begin()
        for c in parts:
            c.compute_forces()
        for c in parts:
            assert lib.tgnh_step_end_kick(c.h, c._stream()) == 0
        exchange()
        for c in parts:
            assert lib.tgnh_step_end_thermo(c.h, c._stream()) == 0
    pos = np.concatenate([c.getPositions() for c in parts])
    vel = np.concatenate([c.getVelocities() for c in parts])
    ep, ev = rel_err(pos, ref.getPositions()), rel_err(vel, ref.getVelocities())
    print(f"{name}, {nranks} shards of {[c.n for c in parts]} slots: pos {ep:.2e} vel {ev:.2e}")
    assert ep < 1e-12 and ev < 1e-10
    for c in parts[1:]:
        assert np.array_equal(parts[0].thermostat_state(1), c.thermostat_state(1))
    assert np.allclose(parts[0].thermostat_state(1), ref.thermostat_state(1), rtol=1e-9, atol=1e-13)
    for c in parts + [ref]:
        c.close()


@pytest.mark.parametrize("precision", ["mixed", "double"])
@pytest.mark.parametrize("sysname,mode,drude_chains", [("mixed", "TGNH", True), ("groups6", "TGNH", True), ("mixed", "dualNH", False),
                                                       ("polymer", "TGNH", True), ("water1000", "TGNH", True)])
def test_collective_hook_path_against_the_oracle(sysname, mode, drude_chains, precision):
    """DEFER_SCALE | RESIDENT_STEP with an all-reduce hook (the RCCL path of a sharded run; step_kernel cannot hold a
    collective, so the handle steps with the deferred launches: rescale+kick+drift with the chain inside, kick+KE, row sum,
    hook).  One rank (its all-reduce is the identity): 100 steps against the oracle, and the hook is really called once per
    step."""
    calls = [0]
    s, g, ng, it, ctx = make(sysname, mode, precision, flags=RESIDENT, chains=1, drude_chains=drude_chains, hardwall=0.02)
    ctx.set_allreduce(lambda t: calls.__setitem__(0, calls[0] + 1))
    o = make_oracle(s, g, ng, mode, it)
    pos_o, vel_o = oracle_run(o, s, 100, x0=ctx.sites())
    ctx.timing(True)
    ctx.step(100)
    ctx.torch.cuda.synchronize()
    ctx.timing(False)
    from openmm_drudenose_amd import _lib
    launches = {_lib.KERNEL_NAMES[k]: ctx.timing_read(k)[1] for k in _lib.KERNEL_NAMES}
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"hook path {sysname} {mode} {precision}: pos {ep:.2e} vel {ev:.2e}, launches {launches}, hook calls {calls[0]}")
    assert ep <= TOL and ev <= TOL and ctx.check() == 0
    assert calls[0] >= 100 and launches["kick+KE"] == 100 and launches["resident step"] == 0
    ctx.close()


@pytest.mark.parametrize("flags,chains", [(FLAG_DEFER_SCALE, 1), (0, 1), (FLAG_DEFER_SCALE, 3)])
def test_native_rccl_site_equals_the_hook_path(flags, chains):
    """tgnh_rccl_init: the library enqueues ncclAllReduce(sum, ncclDouble, NT values) itself, between the row sum and the
    chain.  A one-rank communicator (all a one-GPU box can hold; its all-reduce is the identity) against the same run with an
    identity hook: the same launches around the collective, so positions, velocities and thermostats agree bit for bit --
    eagerly, and with the collective captured into a hipGraph with the steps.  Across ranks the path is the driver's to run
    (bench.py --gpus N uses it for the headline)."""
    runs = {}
    for which in ("hook", "rccl", "rccl-graph"):
        s, g, ng, it, ctx = make("mixed", "TGNH", "mixed", flags=flags, chains=chains, hardwall=0.02)
        if which == "hook":
            ctx.set_allreduce(lambda t: None)
        else:
            ctx.rccl_init(1, 0, ctx.rccl_unique_id())
        if which == "rccl-graph":
            ctx.step(4)
            replay = ctx.capture_steps(4)
            for _ in range(9):
                replay()
        else:
            ctx.step(40)
        ctx.torch.cuda.synchronize()
        assert ctx.check() == 0
        runs[which] = (ctx.getPositions(), ctx.getVelocities(), ctx.thermostat_state(0), ctx.thermostat_state(1))
        if which == "rccl":
            ctx.rccl_shutdown()              # back to no exchange: the handle keeps stepping
            ctx.step(2)
        ctx.close()
    for which in ("rccl", "rccl-graph"):
        for a, b in zip(runs["hook"], runs[which]):
            assert np.array_equal(a, b), which


@pytest.mark.parametrize("flags", [FLAG_DEFER_SCALE, RESIDENT])
def test_collective_hook_path_two_shards_on_one_gpu(flags):
    """The collective-hook path with two shards (two handles on this GPU, one stream), with the deferred launches and with
    RESIDENT_STEP asked for on top (which a handle with a hook cannot use): the hook of the first shard only notes
    its buffer, the hook of the second -- called after both passes are enqueued -- leaves the sum in both, as an all-reduce
    would.  The trajectory is the unsharded one, the thermostats agree bit for bit."""
    from openmm_drudenose_amd.system import shard_bounds
    s, g, ng = synth.mixed(400, 30)
    it = integ(chains=1, hardwall=0.02)
    bind_groups(it, g, ng)
    ref = HipContext(s, it, mode="TGNH", precision="double", flags=FLAG_DEFER_SCALE)
    b = shard_bounds(s, 2)
    parts, terms = [], []
    for r in range(2):
        loc, lg = s.slice_molecules(b[r], b[r + 1]), g[b[r]:b[r + 1]]
        itr = integ(chains=1, hardwall=0.02)
        bind_groups(itr, lg, ng)
        parts.append(HipContext(loc, itr, mode="TGNH", precision="double", flags=flags))
        terms.append(parts[-1].local_dof_terms())
    total = terms[0] + terms[1]
    ref.set_global_dof_terms(total)
    first, known, quiet = [None], [None], [False]

    def hook0(t):
        if quiet[0]:
            return
        if known[0] is not None:
            t.copy_(known[0])
        first[0] = t

    def hook1(t):
        if quiet[0]:
            return
        if known[0] is not None:
            t.copy_(known[0])
            return
        t.add_(first[0])           # rank order: shard 0 + shard 1 (t1 + t0 has the same bits)
        first[0].copy_(t)
    for ctx, hook in zip(parts, (hook0, hook1)):
        ctx.set_global_dof_terms(total)
        ctx.set_allreduce(hook)
    torch = ref.torch
    for step in range(60):
        ref.step_begin(); ref.compute_forces(); ref.step_end()
        if step == 0:
            # the very first half step is a KE pass + hook + rescale inside ONE call per shard (an all-reduce would hold the
            # stream until the other rank has contributed; two handles on one stream cannot): its total is formed up front
            quiet[0] = True                  # (the query goes through the hook too: each shard's own sums are wanted here)
            ke = [torch.from_numpy(c.compute_kinetic_energies()).to(c.dev) for c in parts]
            quiet[0] = False
            known[0] = ke[0] + ke[1]
        for c in parts:
            c.step_begin()
        known[0] = None
        for c in parts:
            c.compute_forces()
        for c in parts:
            c.step_end()
    pos = np.concatenate([c.getPositions() for c in parts])
    vel = np.concatenate([c.getVelocities() for c in parts])
    ep, ev = rel_err(pos, ref.getPositions()), rel_err(vel, ref.getVelocities())
    print(f"hook path, two shards: pos {ep:.2e} vel {ev:.2e}")
    assert ep < 1e-12 and ev < 1e-10
    assert np.array_equal(parts[0].thermostat_state(1), parts[1].thermostat_state(1))
    assert np.allclose(parts[0].thermostat_state(1), ref.thermostat_state(1), rtol=1e-9, atol=1e-13)
    for c in parts + [ref]:
        c.close()


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_sizes_parity(name):
    build, hardwall, nsteps = CONFIGS[name]
    s, g, ng = build()
    it = integ(chains=1, hardwall=hardwall)
    bind_groups(it, g, ng)
    ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE)
    o = make_oracle(s, g, ng, "TGNH", it)
    pos_o, vel_o = oracle_run(o, s, nsteps, x0=ctx.sites())
    ctx.step(nsteps)
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"{name}: N={s.num_particles} P={s.num_pairs} G={ng} steps={nsteps} pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL and ctx.check() == 0
    ctx.close()
    # the same with one launch per step (step_kernel): at these sizes the collecting work-group reads hundreds of rows, in
    # several blocks of 64, and with four groups a row's thermostats in two batches
    it2 = integ(chains=1, hardwall=hardwall)
    bind_groups(it2, g, ng)
    ctx = HipContext(s, it2, mode="TGNH", precision="mixed", flags=RESIDENT)
    assert ctx.resident_work_groups() >= 1
    ctx.step(nsteps)
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"{name}, resident step ({ctx.resident_work_groups()} work-groups per CU): pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL and ctx.check() == 0
    ctx.close()


@pytest.mark.parametrize("flags", [FLAG_DEFER_SCALE, RESIDENT])
@pytest.mark.parametrize("chains", [3, 4])
def test_multi_link_chains_inside_the_launches_at_two_million_slots(chains, flags):
    """Chains of 2-4 links stay inside the streaming launches up to 2.5 M slots (round 4; 1 M before): config 5's box, where the
    instantiations with the links' registers now run with hundreds of work-groups -- against the oracle, and no chain launch."""
    s, g, ng = synth.water_box(400000)
    it = integ(chains=chains, hardwall=0.02)
    bind_groups_array(it, g, ng)
    ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=flags)
    o = make_oracle(s, g, ng, "TGNH", it)
    pos_o, vel_o = oracle_run(o, s, 4, x0=ctx.sites())
    ctx.timing(True)
    ctx.step(4)
    ctx.torch.cuda.synchronize(); ctx.timing(False)
    assert ctx.timing_read(_lib.KID_CHAIN)[1] == 0                   # the chain ran inside the rescale launches / the one-launch step
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"2 M slots, {chains} links, flags {flags}: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL and ctx.check() == 0
    o.propagate_nhc(vel_o.copy())                                    # (deferred: the next step's first thermostat half has run)
    for which in (0, 1):
        a, b = ctx.thermostat_state(which), o.chain(which)
        assert np.allclose(a, b, rtol=1e-6, atol=1e-9 * max(1.0, np.abs(b).max()))
    ctx.close()


# (The sharded counterpart -- two shards of a million slots with the mailboxes -- cannot be rehearsed on ONE GPU: a launch of that
# size fills the device, and a rank whose work-groups all wait in it for the peer's sums keeps the peer's launch from starting; the
# wait is bounded and both handles report "mailbox exchange timed out", with one link as with three (tried, round 4).  Shards
# that small that two launches fit the device together are in tests/oracle_soak.py --sharded and test_random_configurations_gpu.py.)


# ---------------------------------------------------------------------------
# full size (BASELINE.json metric: 1 M Drude pairs): size-independent properties
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["TGNH", "dualNH"])
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 6, 7])
def test_random_ragged_topologies_against_the_oracle(seed, mode):
    """Ragged inputs on the device (tests/helpers.py::random_topology: molecules of 1-40 slots, some longer than a
    tile, Drudes before or after their parents and up to 30 slots away, massless sites, up to 6 groups): 40 steps
    against the oracle, hard wall on; pass structure, precision and chain length vary with the seed."""
    from helpers import random_topology
    mass, pd, pp, resid, group, ngroups, cons, sizes, first, rng = random_topology(seed)
    n = len(mass)
    pos = rng.uniform(0.0, 3.0, (n, 3))
    s, g, ng = synth._finish(mass, np.array(pd, np.int32), np.array(pp, np.int32), resid, pos, group, ngroups, rng, 300.0, 1.0,
                             f"ragged{seed}")
    it = integ(chains=1 + seed % 3, hardwall=0.02)
    if mode == "TGNH":
        bind_groups(it, g, ng)
    else:
        g, ng = np.zeros_like(g), 1
    flags = (FLAG_DEFER_SCALE, 0)[seed % 2]          # both pass structures, both precisions that are gated
    ctx = HipContext(s, it, mode=mode, precision=("double", "mixed")[seed % 2], flags=flags)
    o = make_oracle(s, g, ng, mode, it)
    pos_o, vel_o = oracle_run(o, s, 40, x0=ctx.sites())
    ctx.step(40)
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"ragged seed {seed} {mode}: {n} slots, {len(pd)} pairs, {len(sizes)} molecules, {ng} groups: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL
    ctx.close()


@pytest.mark.parametrize("mode", ["TGNH", "dualNH"])
@pytest.mark.parametrize("seed", [1, 2, 4, 7])
def test_random_constraint_clusters_against_the_oracle(seed, mode):
    """The split (constrained) step on ragged topologies with random constraint clusters of 2-4 atoms (rigid or
    bonds from one atom; tests/helpers.py::random_clusters): SHAKE on posDelta, the velocity stage (TGNH), dof
    reduced per group -- 40 steps against the oracle's constrained loop."""
    from helpers import random_topology, random_clusters
    mass, pd, pp, resid, group, ngroups, cons, sizes, first, rng = random_topology(seed)
    pos = rng.uniform(0.0, 3.0, (len(mass), 3))
    ca, cd = random_clusters(rng, mass, pd, pp, sizes, first, pos)
    s, g, ng = synth._finish(mass, np.array(pd, np.int32), np.array(pp, np.int32), resid, pos, group, ngroups, rng, 300.0, 1.0,
                             f"clusters{seed}")
    s.set_clusters(ca, cd)
    it = integ(chains=1 + seed % 2, hardwall=0.02)
    it.setConstraintTolerance(1e-10)
    if mode == "TGNH":
        bind_groups(it, g, ng)
    else:
        g, ng = np.zeros_like(g), 1
    ctx = HipContext(s, it, mode=mode, precision="double")
    assert ctx.constrained and len(ca) > 10
    o = make_oracle(s, g, ng, mode, it)
    assert np.allclose(ctx.dof()[0], to_internal(o.dof()[0], mode), rtol=1e-14)
    pos_o, vel_o, x0 = s.positions.copy(), s.velocities.copy(), ctx.sites()
    f = o.harness_force(pos_o, x0, synth.K_DRUDE, synth.K_TETHER)
    o.run_harness_constrained(pos_o, vel_o, f, x0, synth.K_DRUDE, synth.K_TETHER, 1e-10, 40)
    ctx.step(40)
    assert ctx.check() == 0
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    print(f"clusters seed {seed} {mode}: {len(ca)} clusters, {len(s.constraints)} constraints: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= TOL and ev <= TOL
    ctx.close()


_FULL = {}


def full_size_case(mode):
    """The metric system (1 M Drude pairs = 5 M slots, the bench's integrator settings) and three oracle steps of it, once per
    mode: the oracle manages 5-10 steps/s at this size.  Tether sites as every float-position context stores them."""
    if "system" not in _FULL:
        _FULL["system"] = synth.water_box(1_000_000)
    s, g, ng = _FULL["system"]
    if mode not in _FULL:
        it = integ(chains=1, hardwall=0.02)
        gm, ngm = (g, ng) if mode == "TGNH" else (np.zeros_like(g), 1)
        o = make_oracle(s, gm, ngm, mode, it)
        x0 = s.positions.astype(np.float32).astype(np.float64)
        _FULL[mode] = oracle_run(o, s, 3, x0=x0) + (x0,)
    return (s, g, ng) + _FULL[mode]


RESIDENT_DEFER = FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP


@pytest.mark.parametrize("mode,flags,precision", [
    ("TGNH", RESIDENT_DEFER, "mixed"),       # bench.py's headline leg: wstep_kernel
    ("dualNH", RESIDENT_DEFER, "mixed"),     # ... its dualNH/mixed/resident leg
    ("TGNH", RESIDENT_DEFER, "single"),      # ... single precision through wstep_kernel (reported; the gate is the single-precision one)
    ("TGNH", FLAG_DEFER_SCALE, "mixed"),     # tile_kernel + wke_kernel, three launches per step
    ("dualNH", FLAG_DEFER_SCALE, "mixed"),
    ("TGNH", 0, "mixed"),                    # the reference's pass structure (what the OpenMM glue runs)
    ("TGNH", FLAG_RESIDENT_STEP, "mixed"),   # ... with each thermostat half one step_kernel launch
    ("TGNH", FLAG_TRUST_STATE_CHANGED, "mixed"),                        # ... without the begin half's KE pass (bench.py's plain-trust leg)
    ("TGNH", FLAG_TRUST_STATE_CHANGED | FLAG_RESIDENT_STEP, "mixed"),   # ... and the end half as one launch (plain-resident-trust)
    ("TGNH", FLAG_GATHER, "mixed"),          # the gather path (tgnh_gather.hip) at the metric size: bench.py's plain-gather variant
    ("dualNH", FLAG_GATHER, "mixed"),
])
def test_full_size_steps_against_the_oracle(mode, flags, precision):
    """The metric configuration itself against the oracle directly, through every launch structure bench.py times: a few steps
    are affordable -- enough to pass through every launch of the step, the in-kernel chain, the row collection over all the
    work-groups of the resident grid (every wavefront of wstep_kernel walks ~20 tiles forward and back here, and work-group 0
    collects the largest number of rows: the regime the 2 M-slot cases do not reach)."""
    s, g, ng, pos_o, vel_o, x0 = full_size_case(mode)
    it = integ(chains=1, hardwall=0.02)
    if mode == "TGNH":
        bind_groups_array(it, g, ng)
    ctx = HipContext(s, it, mode=mode, precision=precision, flags=flags)
    assert np.array_equal(ctx.sites(), x0)
    assert ctx.step_path()[0] == ("gather" if flags & FLAG_GATHER else "tiled")
    ctx.timing(True)
    ctx.step(3)
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    ctx.timing(False)
    nstep = ctx.timing_read(_lib.KID_STEP)[1]
    print(f"5 M slots, 3 steps, {mode} {precision} flags {flags}: pos {ep:.2e} vel {ev:.2e} ({nstep} one-launch steps, "
          f"{ctx.resident_work_groups()} work-groups per CU resident)")
    if flags == RESIDENT_DEFER:
        assert nstep >= 2                   # it really was wstep_kernel: steps 2 and 3 (the first step has no end half to fold in)
    elif flags == FLAG_RESIDENT_STEP:
        assert nstep >= 6                   # both halves of every step
    elif flags == FLAG_TRUST_STATE_CHANGED | FLAG_RESIDENT_STEP:
        assert nstep == 4                   # the first step's begin half and every end half; steps 2 and 3 begin from the carried sums
        assert ctx.timing_read(_lib.KID_KE)[1] == 0
    elif flags == FLAG_TRUST_STATE_CHANGED:
        assert nstep == 0 and ctx.timing_read(_lib.KID_KE)[1] == 1      # one KE pass in all: the first step's
    else:
        assert nstep == 0
    if precision == "single":               # float4 state against the fp64 oracle: test_single_precision_deviation's figures
        assert ep <= 1e-6 and ev <= 1e-3
    else:
        assert ep <= TOL and ev <= TOL      # (thermostat variables are not compared: deferred, the chain is half a step ahead)
    assert ctx.check() == 0
    ctx.close()


@pytest.mark.parametrize("flags", [0, FLAG_DEFER_SCALE, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP, FLAG_TRUST_STATE_CHANGED])
@pytest.mark.parametrize("mode", ["TGNH", "dualNH"])
def test_system_without_any_drude_pair(mode, flags):
    """A DrudeForce without particles (the API only asks for exactly one DrudeForce, DrudeTGNHIntegrator.cpp:110-124): 3 P = 0 Drude
    degrees of freedom, so the Drude thermostat's mass is 0 and its chain is 0/0 in the reference's own arithmetic (Cu :227-235, :605;
    Ref :168-171, :491-492) -- a NaN scale factor that multiplies nothing, since no pair exists.  The oracle shows exactly that; the
    HIP path must too: finite positions and velocities equal to the oracle's, the real thermostats' factors equal, NaN where the
    oracle has NaN, no status bit.  (Massless sites and molecules of one to three atoms in the box.)"""
    rng = np.random.default_rng(5)
    n = 900
    mass = np.full(n, 39.9)
    mass[::7] = 0.0
    resid = np.repeat(np.arange(n // 3), 3)[:n]
    pos = rng.uniform(0, 4, (n, 3))
    vel = rng.normal(0, 0.3, (n, 3))
    vel[mass == 0] = 0
    s = synth.DrudeSystem(mass=mass, pair_drude=np.zeros(0, np.int32), pair_parent=np.zeros(0, np.int32), resid=resid, positions=pos, velocities=vel)
    g = np.zeros(n, np.int32)
    it = integ(chains=2, hardwall=0.02)
    if mode == "TGNH":
        bind_groups(it, g, 1)
    else:
        # the Reference platform reads pairParticles[0] at initialize (Ref :181): undefined behaviour without a pair; the library says so
        with pytest.raises(TgnhError) as e:
            HipContext(s, it, mode=mode, precision="double", flags=flags)
        assert e.value.status == _lib.ERR_UNSUPPORTED and "Drude pair" in str(e.value)
        return
    ctx = HipContext(s, it, mode=mode, precision="double", flags=flags)
    o = make_oracle(s, g, 1, mode, it)
    pos_o, vel_o, kes, scs = oracle_run(o, s, 40, record=True, x0=ctx.sites())
    ctx.step(40)
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    sc, sco = ctx.last_scale_factors(), to_internal(scs, mode)[-1]
    print(f"no Drude pair, {mode}, flags {flags}: pos {ep:.2e} vel {ev:.2e}; scale factors {sc} (oracle {sco})")
    assert np.isfinite(ctx.getPositions()).all() and np.isfinite(ctx.getVelocities()).all()
    assert ep <= TOL and ev <= TOL and ctx.check() == 0
    if not flags & FLAG_DEFER_SCALE:          # (deferred: the factors in hand belong to the next step's first half as well)
        assert np.isnan(sc[-1]) and np.isnan(sco[-1])
        real = ~np.isnan(sco)
        assert np.allclose(sc[real], sco[real], rtol=0, atol=TOL_KE)
    ctx.close()


def test_full_size_properties():
    import torch
    s, g, ng = synth.water_box(1_000_000)
    it = integ(chains=1)
    ctx = HipContext(s, it, mode="TGNH", precision="mixed")
    assert ctx.topology(7).shape[0] - 1 >= 5_000_000 // 512
    # (1) partition identity: sum of all KE bins == sum m v^2 (every molecule inside one group)
    ke = ctx.compute_kinetic_energies()
    w = ctx.velm[:, 3]
    m = torch.where(w > 0, 1.0 / torch.where(w > 0, w, torch.ones_like(w)), torch.zeros_like(w))
    total = float((m[:, None] * ctx.velm[:, :3] ** 2).sum())
    assert ke.sum() == pytest.approx(total, rel=1e-10)
    # (2) determinism: the reduction order is fixed (no atomics), and a query leaves the sweep direction -- which orders the
    #     additions -- as it found it: asked again it returns the same bits, and so does the step's own KE launch (3)
    for _ in range(3):
        assert np.array_equal(ctx.compute_kinetic_energies(), ke)
    #     ... and the plain query (A12: OpenMM's computeKineticEnergy(0), Cu :656) sums work-group partials in index order
    assert not ctx.ke_sum_valid
    k1 = ctx.kinetic_energy()
    assert k1 == ctx.kinetic_energy() == ctx.kinetic_energy() and k1 == pytest.approx(0.5 * total, rel=1e-12)
    # (3) rescale: every bin is s^2 times its value before
    ctx.step_begin()
    ke0, sc = ctx.last_kinetic_energies(), ctx.last_scale_factors()
    assert np.allclose(ke0, ke, rtol=1e-13)          # (the step sums its partial rows in the rescale launch's prologue, the query in rowsum_kernel)
    ctx.compute_forces(); ctx.step_end()
    ke2, sc2 = ctx.last_kinetic_energies(), ctx.last_scale_factors()
    ke3 = ctx.compute_kinetic_energies()
    assert np.allclose(ke3, ke2 * sc2 ** 2, rtol=1e-9)
    # (4) massless sites never move, Drude pairs stay bound, everything finite after 20 more steps
    ctx.step(20)
    pos, vel = ctx.getPositions(), ctx.getVelocities()
    assert np.isfinite(pos).all() and np.isfinite(vel).all()
    msk = s.mass == 0
    assert np.array_equal(pos[msk].astype(np.float32), s.positions[msk].astype(np.float32))
    r = np.linalg.norm(pos[s.pair_drude] - pos[s.pair_parent], axis=1)
    assert r.max() < 0.01
    # (5) thermostats keep every kinetic-energy bin in a physical range (the harness oscillators start at
    #     their potential minimum, so KE first drops towards its equipartition share)
    dof, nkt = ctx.dof()
    ke_end = ctx.compute_kinetic_energies()
    assert np.all(ke_end / nkt > 0.3) and np.all(ke_end / nkt < 1.5)
    ctx.close()
