"""CPU tests of the oracle (oracle/tgnh_oracle.c): internal consistency, the bridge
identity between the two semantic modes, and the reference's own statistical
known-answer test.  No GPU."""
import numpy as np
import pytest

from openmm_drudenose_amd import synth
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator
from oracle import Oracle, MODE_DUALNH, MODE_TGNH
from helpers import ONE_4PI_EPS0, oracle_run, rel_err

KB = synth.KB


def _integ(**kw):
    a = dict(temperature=300.0, couplingTime=0.1, drudeTemperature=1.0, drudeCouplingTime=0.005, stepSize=0.001,
             drudeStepsPerRealStep=20, numNHChains=3, useDrudeNHChains=True, useCOMTempGroup=True)
    a.update(kw)
    return DrudeTGNHIntegrator(**a)


def _oracle(system, group, ng, mode, integ):
    return Oracle.from_integrator(system, integ, group, ng, mode)


def test_normal_particles_and_pairs_match_set_semantics():
    # Ref :113-137: normalParticles = ascending members of {0..N-1} minus every p and p1
    s, g, ng = synth.mixed(20, 3)
    o = _oracle(s, g, ng, MODE_TGNH, _integ())
    expect = np.setdiff1d(np.arange(s.num_particles), np.r_[s.pair_drude, s.pair_parent])
    assert np.array_equal(o.normal_particles(), expect.astype(np.int32))


def test_dof_dualnh_water():
    # Ref :119,:133,:157-165: realDof = 3*#massive - 3P - constraints - 3[CMM]
    s, g, ng = synth.water_box(8)
    s.has_cm_motion_remover = True
    o = _oracle(s, g, ng, MODE_DUALNH, _integ())
    dof, nkt = o.dof()
    assert dof[0] == 3 * (4 * 8) - 3 * 8 - 3 and dof[1] == 3 * 8
    assert nkt[0] == pytest.approx(dof[0] * KB * 300.0, rel=1e-15)
    assert nkt[1] == pytest.approx(dof[1] * KB * 1.0, rel=1e-15)


def test_dof_tgnh_groups_and_com():
    # Cu :126-134, :195-212, :219: dof_g - red_g with red_g = sum 3 m_i / M_res ; COM = 3R - 3[CMM]
    s, g, ng = synth.ionic_liquid(4)
    s.has_cm_motion_remover = True
    o = _oracle(s, g, ng, MODE_TGNH, _integ())
    dof, _ = o.dof()
    # every molecule is inside one group, so red_g = 3 * (#molecules of g)
    assert dof[0] == pytest.approx(3 * 35 * 4 - 3 * 10 * 4 - 3 * 4, rel=1e-14)     # cations: 35 massive sites, 10 pairs
    assert dof[1] == pytest.approx(3 * 10 * 4 - 3 * 5 * 4 - 3 * 4, rel=1e-14)
    assert dof[2] == 3 * 8 - 3 and dof[3] == 3 * 15 * 4
    o2 = _oracle(s, g, ng, MODE_TGNH, _integ(useCOMTempGroup=False))
    dof2, _ = o2.dof()
    assert dof2[0] == 3 * 35 * 4 - 3 * 10 * 4 and dof2[2] == 0


def test_group_mismatch_raises():
    s, g, ng = synth.water_box(2)
    g = g.copy(); g[1] = 1      # Drude in another group than its parent: Cu :145-146
    with pytest.raises(Exception, match="Temperature group for drude particle"):
        _oracle(s, g, 2, MODE_TGNH, _integ())


def test_pair_kick_identity():
    """Ref :565-583 writes the pair half kick in COM/relative coordinates; it equals v += dt/2 F/m."""
    s, g, ng = synth.mixed(30, 4)
    rng = np.random.default_rng(1)
    f = rng.normal(0, 500.0, s.positions.shape)
    for mode in (MODE_DUALNH, MODE_TGNH):
        o = _oracle(s, g, ng, mode, _integ())
        v = s.velocities.copy()
        o.half_kick(v, f)
        inv = np.where(s.mass > 0, 1.0 / np.where(s.mass > 0, s.mass, 1), 0.0)
        expect = s.velocities + 0.5 * 0.001 * inv[:, None] * f
        assert rel_err(v, expect) < 1e-14


def test_ke_partition_identity():
    """sum_g KE_g + KE_COM + KE_Drude == sum m v^2 when no molecule spans groups."""
    s, g, ng = synth.mixed(40, 5)
    o = _oracle(s, g, ng, MODE_TGNH, _integ())
    ke = o.kinetic_energies(s.velocities)
    total = float((s.mass[:, None] * s.velocities ** 2).sum())
    assert ke.sum() == pytest.approx(total, rel=1e-12)
    od = _oracle(s, g, ng, MODE_DUALNH, _integ())
    assert od.kinetic_energies(s.velocities).sum() == pytest.approx(total, rel=1e-12)


def test_two_body_analytic_split():
    """G = 1 with the COM group, one molecule that is one Drude pair.  Everything is analytic: the
    group bin holds the pair's centre of mass *relative to the molecule's* (zero), the COM bin M |v_cm|^2, the Drude
    bin mu |v2 - v1|^2 (K :154, :184-185); rescaling by (s_g, s_COM, s_D) gives v_i = s_COM v_cm -/+ s_D (m_j/M) v_rel
    (K :272-300)."""
    s, g, ng = synth.single_pair()
    o = _oracle(s, g, ng, MODE_TGNH, _integ())
    m1, m2 = s.mass
    v1, v2 = s.velocities
    vcm, vrel = (m1 * v1 + m2 * v2) / (m1 + m2), v2 - v1
    ke = o.kinetic_energies(s.velocities)
    assert ke[0] == pytest.approx(0.0, abs=1e-28)
    assert ke[1] == pytest.approx((m1 + m2) * vcm @ vcm, rel=1e-14)
    assert ke[2] == pytest.approx(m1 * m2 / (m1 + m2) * vrel @ vrel, rel=1e-14)
    v = s.velocities.copy()
    o.scale_velocities(v, [0.7, 1.1, 0.9])
    assert np.allclose(v[0], 1.1 * vcm - 0.9 * m2 / (m1 + m2) * vrel, rtol=1e-14)
    assert np.allclose(v[1], 1.1 * vcm + 0.9 * m1 / (m1 + m2) * vrel, rtol=1e-14)


def test_rescale_scales_ke_by_square():
    """After the rescale each KE bin is s^2 times its old value (single-group molecules)."""
    s, g, ng = synth.mixed(40, 5)
    for mode in (MODE_DUALNH, MODE_TGNH):
        o = _oracle(s, g, ng, mode, _integ())
        v = s.velocities.copy()
        ke0, sc = o.propagate_nhc(v)
        ke1 = o.kinetic_energies(v)
        assert np.allclose(ke1, ke0 * sc ** 2, rtol=1e-12, atol=0)


@pytest.mark.parametrize("chains", [1, 3])
@pytest.mark.parametrize("hardwall", [0.0, 0.02])
def test_bridge_identity_tgnh_equals_dualnh(chains, hardwall):
    """G=1, no COM group, useDrudeNHChains, no CMMotionRemover, no constraints
    => the CUDA-platform algorithm is the Reference-platform algorithm."""
    s, g, ng = synth.water_box(27)
    it = _integ(numNHChains=chains, useCOMTempGroup=False, useDrudeNHChains=True)
    it.setMaxDrudeDistance(hardwall)
    od, ot = _oracle(s, g, ng, MODE_DUALNH, it), _oracle(s, g, ng, MODE_TGNH, it)
    pd_, vd = oracle_run(od, s, 100)
    pt, vt = oracle_run(ot, s, 100)
    assert rel_err(pt, pd_) < 1e-12 and rel_err(vt, vd) < 1e-11
    # thermostat variables too: TGNH rows [0]=group, [1]=COM(inert), [2]=Drude vs interleaved [r0,d0,r1,d1..]
    C = chains
    ed, et = od.chain(1), ot.chain(1).reshape(3, C + 1)
    assert np.allclose(et[0, :C], ed[0:2 * C:2], rtol=1e-10, atol=1e-14)
    assert np.allclose(et[2, :C], ed[1:2 * C:2], rtol=1e-10, atol=1e-14)
    assert np.all(et[1, 0] == 0)


def test_dualnh_without_drude_chains_is_coupled():
    """Ref :476-481 with numTempGroup=1: real link 0 is damped with the Drude thermostat's
    etaDot (bug-compatible indexing) -- so the result differs from the consistent layout."""
    s, g, ng = synth.water_box(8)
    a = _oracle(s, g, ng, MODE_DUALNH, _integ(numNHChains=1, useDrudeNHChains=False))
    b = _oracle(s, g, ng, MODE_DUALNH, _integ(numNHChains=1, useDrudeNHChains=True))
    assert len(a.chain(1)) == 1 + 3 and len(b.chain(1)) == 2 + 2
    pa, va = oracle_run(a, s, 50)
    pb, vb = oracle_run(b, s, 50)
    assert np.isfinite(va).all() and np.isfinite(vb).all()
    assert rel_err(va, vb) > 1e-9


def test_hardwall_pulls_pair_back_inside():
    s, g, ng = synth.water_box(8)
    it = _integ()
    it.setMaxDrudeDistance(0.02)
    for mode in (MODE_DUALNH, MODE_TGNH):
        o = _oracle(s, g, ng, mode, it)
        pos, vel = s.positions.copy(), s.velocities.copy()
        pos[s.pair_drude[0]] = pos[s.pair_parent[0]] + np.array([0.0, 0.0, 0.03])
        vel[s.pair_drude[0]] = vel[s.pair_parent[0]] + np.array([0.3, 0.0, 2.0])   # moving outward, as after a drift
        o.hardwall(pos, vel)
        r = np.linalg.norm(pos[s.pair_drude] - pos[s.pair_parent], axis=1)
        assert r.max() <= 0.02 * (1 + 1e-9)
    o = _oracle(s, g, ng, MODE_DUALNH, it)
    pos = s.positions.copy()
    pos[s.pair_drude[0]] = pos[s.pair_parent[0]] + np.array([0.0, 0.0, 0.05])      # > 2 x wall: Ref :311-312
    with pytest.raises(Exception, match="too far beyond hard wall"):
        o.hardwall(pos, s.velocities.copy())


def test_reference_testSinglePair_statistics():
    """The reference's testSinglePair (TestReferenceDrudeTGNHIntegrator.cpp:54-109; commented out in its
    main() :257-259): one pair on a harmonic Drude spring k = ONE_4PI_EPS0*q^2/alpha = ONE_4PI_EPS0*1.5, hard
    wall 0.05 nm, DrudeTGNHIntegrator(300, 0.1, 10, 0.005, 0.003, 20, 2, false).  Its assertions:
      (a) r <= max(1+1e-6) at every sample                      -> holds for the oracle
      (b) mean KE_internal = 3/2 kT(10 K) within 1 %            -> holds for the oracle (0.998)
      (c) mean KE_cm = 3/2 kT(300 K) within 10 %                -> does NOT hold (0.76): with
          useDrudeNHChains=false the real thermostat is damped by the Drude thermostat's etaDot
          (Ref :476-481), which is presumably why the reference keeps this test disabled.
    (c) is therefore recorded, not asserted; this test is a weak statistical pin only."""
    s, g, ng = synth.single_pair()
    it = DrudeTGNHIntegrator(300.0, 0.1, 10.0, 0.005, 0.003, 20, 2, False)
    it.setMaxDrudeDistance(0.05)
    o = _oracle(s, g, ng, MODE_DUALNH, it)
    k = ONE_4PI_EPS0 * 1.5
    pos, vel, x0 = s.positions.copy(), s.velocities.copy(), s.positions.copy()
    f = o.harness_force(pos, x0, k, 0.0)
    o.run_harness(pos, vel, f, x0, k, 0.0, 1000)
    m1, m2 = 1.0, 0.1
    tot, red = m1 + m2, m1 * m2 / (m1 + m2)
    ke_cm = ke_int = 0.0
    nsamp = 10000
    for _ in range(nsamp):
        o.run_harness(pos, vel, f, x0, k, 0.0, 10)
        vcm = vel[0] * (m1 / tot) + vel[1] * (m2 / tot)
        ke_cm += 0.5 * tot * vcm.dot(vcm)
        vi = vel[0] - vel[1]
        ke_int += 0.5 * red * vi.dot(vi)
        assert np.linalg.norm(pos[0] - pos[1]) <= 0.05 * (1 + 1e-6)                 # (a)
    assert ke_int / nsamp == pytest.approx(1.5 * KB * 10.0, rel=0.01)              # (b)
    ratio_c = ke_cm / nsamp / (1.5 * KB * 300.0)                                   # (c) recorded
    assert 0.5 < ratio_c < 1.5


def test_oracle_regression_vectors():
    """tests/golden/oracle_regression.npz: frozen outputs of THIS oracle (not reference outputs -- the reference cannot
    run here); any edit that changes what the oracle, the synthetic builders or the harness force compute shows up."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_regression_vectors", os.path.join(path, "make_regression_vectors.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    frozen = np.load(os.path.join(path, "oracle_regression.npz"))
    for name in gen.CASES:
        now = gen.run(name)
        for k, v in now.items():
            ref = frozen[f"{name}/{k}"]
            assert ref.shape == v.shape and np.allclose(v, ref, rtol=1e-12, atol=1e-14 * max(1.0, np.abs(ref).max())), (name, k)


@pytest.mark.parametrize("mode,system", [("TGNH", "ionic"), ("TGNH", "water"), ("dualNH", "water")])
def test_extended_energy_is_conserved(mode, system):
    """the Nose-Hoover-chain invariant.  While the thermostats take > 25 % of the initial energy out of
    the particles, H stays put, and its fluctuation falls ~4x when dt is halved (second-order splitting)."""
    from helpers import extended_energy
    s, g, ng = synth.ionic_liquid(4) if system == "ionic" else synth.water_box(27)
    C, T, TD = 3, 300.0, 1.0
    worst = []
    for dt in (0.0005, 0.00025):
        it = _integ(stepSize=dt, numNHChains=C)
        o = _oracle(s, g, ng, MODE_TGNH if mode == "TGNH" else MODE_DUALNH, it)
        pos, vel, x0 = s.positions.copy(), s.velocities.copy(), s.positions.copy()
        f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
        normal = o.normal_particles()

        def energy():
            return extended_energy(s, normal, pos, vel, x0, o.dof()[1], o.chain(0), o.chain(1), o.chain(3), C,
                                   KB * T, KB * TD, mode)
        h0, ke0, _ = energy()
        dev, th_min = 0.0, 0.0
        for _ in range(10):
            o.run_harness(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, int(round(0.02 / dt)))
            h, _, th = energy()
            dev, th_min = max(dev, abs(h - h0)), min(th_min, th)
        assert th_min < -0.25 * ke0, "the thermostats must have done real work for this test to mean anything"
        worst.append(dev / h0)
    print("dH/H0:", worst)
    assert worst[0] < 2e-4 and worst[1] < 5e-5
    assert 2.5 < worst[0] / worst[1] < 7.0
