"""GPU parity tests of the GATHER path (tgnh_gather.hip): what the tiled kernels cannot hold -- a Drude particle more than a tile
from its parent, pairs overlapping so densely that no tile cut exists, more than 32 temperature groups, residues in several
runs -- is not refused (the reference gathers by arbitrary index, K :171-186, and sizes its bins by G + 2, K :138-200) but steps
through the reference's own un-fused kernels by global index.  Same gates as tests/test_gpu_parity.py: topology arrays bit-exact,
positions and velocities within 1e-6 relative over 100 steps (mixed, double), KE[] and scale[] per step within 1e-6, against
the CPU oracle through the C ABI."""
import numpy as np
import pytest

from openmm_drudenose_amd import synth, _lib
from openmm_drudenose_amd.drudetgnhplugin import (DrudeTGNHIntegrator, HipContext, TgnhError, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP,
                                                   FLAG_TRUST_STATE_CHANGED, FLAG_GATHER)
from helpers import make_oracle, oracle_run, rel_err, to_internal, drudes_at_the_end, onion, far_pairs, interleaved

pytestmark = pytest.mark.gpu
TOL = 1e-6


def integ(chains=3, drude_chains=True, com=True, hardwall=0.0, dt=0.001):
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, dt, 20, chains, drude_chains, com)
    it.setMaxDrudeDistance(hardwall)
    return it


def bind_groups(it, group, ngroups):
    for _ in range(ngroups):
        it.addTempGroup()
    it._particleTempGroup = np.ascontiguousarray(group, np.int32)


CASES = {
    "groups33": lambda: synth.many_groups(300, 20, 33),          # one more than the tiled kernels' bins hold
    "groups40": lambda: synth.many_groups(300, 20, 40),          # more than chain_kernel's 34 thermostats: gather_rowsum / gather_chain
    "groups300": lambda: synth.many_groups(400, 20, 300),
    "groups2046": lambda: synth.many_groups(2100, 60, 2046),      # the most the kinetic-energy kernel's per-wavefront bins hold (64 KiB of LDS); 2047: refused
    "drudes-at-the-end": lambda: drudes_at_the_end(300),          # (COM group off in TGNH mode: with it the reference's walk of `count`
                                                                  # particles from a residue's last run leaves the array for the last residues)
    "interleaved": interleaved,
    "onion": onion,
    "far-pairs": far_pairs,
}


def run_case(name, mode, precision, nsteps=100, flags=0, **kw):
    s, g, ng = CASES[name]()
    it = integ(**kw)
    if mode == "TGNH":
        bind_groups(it, g, ng)
    else:
        g, ng = np.zeros_like(g), 1
    ctx = HipContext(s, it, mode=mode, precision=precision, flags=flags)
    path, why = ctx.step_path()
    assert path == "gather" and why, (path, why)
    o = make_oracle(s, g, ng, mode, it)
    # A1: the reference's own index lists, bit-exact
    assert np.array_equal(ctx.topology(0), o.normal_particles())
    assert np.array_equal(ctx.topology(1), s.pair_drude) and np.array_equal(ctx.topology(2), s.pair_parent)
    assert np.allclose(ctx.dof()[0], to_internal(o.dof()[0], mode), rtol=1e-14)
    pos_o, vel_o, kes, scs = oracle_run(o, s, nsteps, record=True, x0=ctx.sites())
    worst_ke = worst_sc = 0.0
    m = np.ones(ctx.num_thermostats(), bool)
    if mode == "dualNH":
        m[1] = False                                           # (the library's unused middle slot: [real, -, Drude])
    for k in range(nsteps):
        ctx.step_begin()
        worst_ke = max(worst_ke, rel_err(ctx.last_kinetic_energies(), to_internal(kes[2 * k], mode)))
        worst_sc = max(worst_sc, np.abs(ctx.last_scale_factors()[m] - to_internal(scs[2 * k], mode)[m]).max())
        ctx.compute_forces()
        ctx.step_end()
        worst_ke = max(worst_ke, rel_err(ctx.last_kinetic_energies(), to_internal(kes[2 * k + 1], mode)))
        worst_sc = max(worst_sc, np.abs(ctx.last_scale_factors()[m] - to_internal(scs[2 * k + 1], mode)[m]).max())
    ep, ev = rel_err(ctx.getPositions(), pos_o), rel_err(ctx.getVelocities(), vel_o)
    flags_seen = ctx.status_flags()
    ctx.close()
    print(f"gather {name} {mode} {precision} ({why}): pos {ep:.2e} vel {ev:.2e} KE {worst_ke:.2e} scale {worst_sc:.2e} status {flags_seen}")
    return ep, ev, worst_ke, worst_sc


@pytest.mark.parametrize("precision", ["mixed", "double"])
@pytest.mark.parametrize("name", list(CASES))
def test_100_step_parity_on_the_gather_path(name, precision):
    """TGNH mode (platforms/cuda's semantics), three links, COM group on, hard wall on"""
    ep, ev, eke, esc = run_case(name, "TGNH", precision, hardwall=0.02, com=name != "drudes-at-the-end")
    assert ep < TOL and ev < TOL and eke < TOL and esc < TOL, (ep, ev, eke, esc)


@pytest.mark.parametrize("name,kw", [("groups40", dict(chains=1)), ("groups40", dict(chains=6, drude_chains=False)),
                                     ("groups300", dict(chains=2, com=False)), ("far-pairs", dict(chains=1, com=False)),
                                     ("onion", dict(chains=5))])
def test_gather_path_chain_lengths_and_switches(name, kw):
    """one link, 5-6 links (the scratch-row form for more than 34 thermostats), the Drude thermostat's higher links frozen, COM group off"""
    ep, ev, eke, esc = run_case(name, "TGNH", "double", nsteps=60, **kw)
    assert ep < TOL and ev < TOL and eke < TOL and esc < TOL, (ep, ev, eke, esc)


@pytest.mark.parametrize("name", ["drudes-at-the-end", "onion", "far-pairs"])
@pytest.mark.parametrize("drude_chains", [True, False])
def test_dualnh_on_the_gather_path(name, drude_chains):
    """platforms/reference's algorithm (Ref :426-546) on the same topologies, the indexing quirk of useDrudeNHChains = false included"""
    ep, ev, eke, esc = run_case(name, "dualNH", "mixed", hardwall=0.02, drude_chains=drude_chains)
    assert ep < TOL and ev < TOL and eke < TOL and esc < TOL, (ep, ev, eke, esc)


def test_flags_that_change_the_pass_structure_are_ignored_on_the_gather_path():
    """DEFER_SCALE / RESIDENT_STEP / TRUST_STATE_CHANGED leave the trajectory what it is on the tiled path; the gather path steps in
    the reference's own structure whatever is asked: velocities never lag, setters between steps are allowed, same results"""
    ref = run_case("far-pairs", "TGNH", "double", nsteps=40, hardwall=0.02)
    for flags in (FLAG_DEFER_SCALE, FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP, FLAG_TRUST_STATE_CHANGED):
        got = run_case("far-pairs", "TGNH", "double", nsteps=40, flags=flags, hardwall=0.02)
        assert got == ref, (flags, got, ref)
    s, g, ng = CASES["far-pairs"]()
    it = integ(hardwall=0.02)
    bind_groups(it, g, ng)
    ctx = HipContext(s, it, mode="TGNH", precision="double", flags=FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)
    assert ctx.resident_kernel() is None and ctx.pending_state() & 0xff == 0
    ctx.step(3)
    assert ctx.pending_state() & 0xff == 0                     # nothing owed between steps
    ctx.setVelocities(ctx.getVelocities())                     # (refused between the steps of a deferred sequence on the tiled path)
    ctx.close()


def test_kinetic_energy_queries_on_the_gather_path():
    """tgnh_compute_kinetic_energies (the A3/A4 bins of the current velocities) and the A12 query, 40 groups"""
    s, g, ng = CASES["groups40"]()
    it = integ(chains=2)
    bind_groups(it, g, ng)
    ctx = HipContext(s, it, mode="TGNH", precision="double")
    o = make_oracle(s, g, ng, "TGNH", it)
    ctx.step(5)
    vel = ctx.getVelocities()
    assert rel_err(ctx.compute_kinetic_energies(), o.kinetic_energies(vel)) < 1e-12
    ke_sum = 0.5 * float((s.mass[:, None] * vel ** 2).sum())
    ctx.ke_sum_valid = False
    assert abs(ctx.kinetic_energy() - ke_sum) < 1e-10 * ke_sum
    ctx.close()


def test_velocities_written_between_a_kinetic_energy_query_and_a_step():
    """The gather path's rescale launches reuse the centre-of-mass table of the kinetic-energy pass before them (run_gather,
    tgnh_host.cpp) -- inside one entry point only: a query's table must not outlive velocities the caller writes after it.
    Today every rescale of this path follows a kinetic-energy pass inside the same entry point (the flags that would part them
    are ignored), so no sequence of calls reaches a stale table even without entry()'s clearing of the mark (checked once with a
    build that lacked it: this test passed there too); it pins the sequence that a change to that rule would break first.  The
    same calls on a tiled handle of the same system (TGNH mode, COM group on, a different drift added to every molecule)."""
    out = []
    for flags in (0, FLAG_GATHER):
        s, g, ng = synth.mixed(200, 15)
        it = integ(chains=1)
        bind_groups(it, g, ng)
        ctx = HipContext(s, it, mode="TGNH", precision="double", flags=flags)
        ctx.step(3)
        ctx.compute_kinetic_energies()                          # leaves a table of THESE velocities
        rng = np.random.default_rng(5)
        v = ctx.getVelocities()
        per_mol = rng.normal(0.0, 0.5, (int(s.resid.max()) + 1, 3))          # a different drift for every molecule: every COM velocity changes
        ctx.setVelocities(v + per_mol[s.resid])
        ctx.step(2)
        out.append((ctx.getPositions(), ctx.getVelocities()))
        ctx.close()
    (p0, v0), (p1, v1) = out
    assert rel_err(p1, p0) < 1e-12 and rel_err(v1, v0) < 1e-11, (rel_err(p1, p0), rel_err(v1, v0))


def test_residues_scattered_all_over_the_array_stay_inside_it():
    """A residue array in random order (every residue in many runs): the reference's table then says (count, start of the LAST run)
    and its COM kernel walks `count` particles from there (K :90-91) -- behind the end of the arrays for the residues whose last
    run lies near it: undefined in the reference and in the oracle alike, so there is no trajectory to compare.  The library's walk
    stops at the array's end (gather_com_kernel), tgnh_create's tile bookkeeping too (found by tests/test_desc_fuzz.py): the handle
    steps, everything stays finite, the same launches give the same bits twice."""
    out = []
    for rep in range(2):
        s, g, ng = synth.mixed(120, 10)
        rng = np.random.default_rng(11)
        resid = s.resid.copy()
        rng.shuffle(resid)
        s2 = type(s)(mass=s.mass, pair_drude=s.pair_drude, pair_parent=s.pair_parent, resid=resid, positions=s.positions,
                     velocities=s.velocities)
        it = integ(chains=2, hardwall=0.02)
        bind_groups(it, np.zeros_like(g), 1)                    # (one group: a Drude particle and its parent must share theirs, Ref :128-131)
        ctx = HipContext(s2, it, mode="TGNH", precision="mixed")
        assert ctx.step_path() == ("gather", "particles of a residue are not contiguous")
        ctx.step(10)
        pos, vel = ctx.getPositions(), ctx.getVelocities()
        assert np.isfinite(pos).all() and np.isfinite(vel).all() and np.isfinite(ctx.compute_kinetic_energies()).all()
        assert ctx.status_flags() & ~1 == 0
        out.append((pos, vel))
        ctx.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_more_groups_than_the_bins_hold_are_refused():
    s, g, ng = synth.many_groups(2100, 60, 2047)
    it = integ()
    bind_groups(it, g, ng)
    with pytest.raises(TgnhError) as e:
        HipContext(s, it, mode="TGNH", precision="double")
    assert e.value.status == _lib.ERR_UNSUPPORTED and "2046 temperature groups" in str(e.value)


def test_sharded_gather_path_with_an_allreduce_hook():
    """Two handles own the two halves of the molecules of a 40-group box (more than 34 thermostats: the gather path's own row sum
    and chain), each hook adds the other's kinetic-energy sums (what RCCL does across GPUs; the pattern of
    test_particle_sharded_hip_path_on_one_gpu): the trajectory of the unsharded run, thermostats bit-identical over the shards.
    The mailbox exchange holds <= 34 thermostats and says so."""
    from openmm_drudenose_amd.system import shard_bounds
    s, g, ng = CASES["groups40"]()
    it = integ(chains=2, hardwall=0.02)
    bind_groups(it, g, ng)
    ref = HipContext(s, it, mode="TGNH", precision="double")
    assert ref.step_path()[0] == "gather"
    b = shard_bounds(s, 2)
    parts, terms = [], []
    for r in range(2):
        loc, lg = s.slice_molecules(b[r], b[r + 1]), g[b[r]:b[r + 1]]
        itr = integ(chains=2, hardwall=0.02)
        bind_groups(itr, lg, ng)
        ctx = HipContext(loc, itr, mode="TGNH", precision="double")
        parts.append(ctx)
        terms.append(ctx.local_dof_terms())
    with pytest.raises(TgnhError, match="too many thermostats"):
        parts[0].exchange_create(2, 0)
    total = terms[0] + terms[1]
    assert np.allclose(total, ref.local_dof_terms(), rtol=1e-13)
    peer = [None, None]
    for r, ctx in enumerate(parts):
        ctx.set_global_dof_terms(total)
        ctx.set_allreduce(lambda t, r=r: t.add_(peer[r]) if peer[r] is not None else None)
    torch = ref.torch

    def exchange():
        peer[0] = peer[1] = None
        ke = [torch.from_numpy(c.compute_kinetic_energies()).to(c.dev) for c in parts]
        peer[0], peer[1] = ke[1], ke[0]

    lib = ref.lib
    for _ in range(30):
        ref.step_begin(); ref.compute_forces(); ref.step_end()
        exchange()
        for c in parts:
            assert lib.tgnh_step_begin_kick(c.h, c._stream()) == 0
            assert lib.tgnh_step_begin_move(c.h, c._stream()) == 0
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
    assert np.array_equal(parts[0].thermostat_state(1), parts[1].thermostat_state(1))     # replicated chain: bitwise
    assert np.allclose(parts[0].thermostat_state(1), ref.thermostat_state(1), rtol=1e-9, atol=1e-13)
    for c in parts + [ref]:
        c.close()


@pytest.mark.parametrize("sysname,mode", [("mixed", "TGNH"), ("polymer", "TGNH"), ("water-rigid", "TGNH"), ("il-shake", "dualNH")])
def test_the_gather_flag_gives_the_tiled_path_s_trajectory(sysname, mode):
    """TGNH_FLAG_GATHER on topologies the tiles CAN hold: the two implementations against each other, 60 steps -- unconstrained
    boxes and, through the split entry points with the harness' SHAKE / virtual-site call-outs, the constrained ones"""
    build = {"mixed": lambda: synth.mixed(300, 20), "polymer": lambda: synth.polymer_in_water(700, 100),
             "water-rigid": lambda: synth.water_box(216, rigid=True), "il-shake": lambda: synth.ionic_liquid(40, constrained=True)}[sysname]
    out = []
    for flags in (0, FLAG_GATHER):
        s, g, ng = build()
        it = integ(chains=3, hardwall=0.02)
        it.setConstraintTolerance(1e-10)
        if mode == "TGNH":
            bind_groups(it, g, ng)
        ctx = HipContext(s, it, mode=mode, precision="double", flags=flags)
        assert ctx.step_path()[0] == ("gather" if flags else "tiled")
        ctx.step(60)
        out.append((ctx.getPositions(), ctx.getVelocities(), ctx.thermostat_state(1), ctx.status_flags() & ~1))
        ctx.close()
    (p0, v0, t0, f0), (p1, v1, t1, f1) = out
    print(f"gather flag vs tiled, {sysname} {mode}: pos {rel_err(p1, p0):.2e} vel {rel_err(v1, v0):.2e}")
    assert f0 == 0 and f1 == 0
    assert rel_err(p1, p0) < 1e-11 and rel_err(v1, v0) < 1e-9 and np.allclose(t1, t0, rtol=1e-8, atol=1e-10)
