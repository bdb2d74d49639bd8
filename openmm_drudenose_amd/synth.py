"""Synthetic Drude systems: the configs of BASELINE.json / SURVEY.md 8(d).

No OpenMM, no force field files: topology, masses and geometry are generated
(seed 20191024).  Site masses follow the reference's own test
(platforms/reference/tests/TestReferenceDrudeTGNHIntegrator.cpp:132-136) and example
(example/nacl_tg.py:49-53).  Every builder returns (DrudeSystem, group[N], num_groups).
"""
import numpy as np

from .system import DrudeSystem

SEED = 20191024
KB = 8.31446261815324e-3        # kJ/mol/K (OpenMM BOLTZ, CODATA 2018)
K_DRUDE = 4184.0 * 100.0        # kJ/mol/nm^2, SWM4-NDP Drude spring (TestReference...:149: 100000*4.184)
K_TETHER = 1000.0               # kJ/mol/nm^2, harness tether of every non-Drude massive site
DRUDE_SIGMA = None              # override of the initial Drude displacement spread (nm); None = thermal

# SWM4-NDP site tables: (mass, is_drude, parent offset, geometry offset)
_W_TEST = dict(mass=[15.6, 0.4, 1.0, 1.0, 0.0], drude=1, parent=0,      # O, D, H1, H2, M  (TestReference...:132-136)
               geom=[(0, 0, 0), (0, 0, 0), (0.09572, 0, 0), (-0.023999, 0.092663, 0), (0.0, 0.0, 0.0)])
_W_NACL = dict(mass=[15.5994, 1.008, 1.008, 0.0, 0.4], drude=4, parent=0,   # O, H1, H2, M, D  (nacl_1m_pos.pdb order)
               geom=[(0, 0, 0), (0.09572, 0, 0), (-0.023999, 0.092663, 0), (0, 0, 0), (0, 0, 0)])


def _lattice(n, spacing):
    side = int(np.ceil(n ** (1.0 / 3.0) - 1e-9))
    idx = np.arange(n)
    return np.stack([idx // (side * side), (idx // side) % side, idx % side], 1).astype(np.float64) * spacing


def _finish(mass, pair_drude, pair_parent, resid, pos, group, num_groups, rng, temperature, drude_temperature,
            name, constraints=None, cmm=False):
    n = mass.shape[0]
    # Drude displaced from its parent by the thermal spread of the spring at the Drude temperature,
    # N(0, sqrt(kB T_D / k)) per axis, so the cold thermostat starts near equilibrium
    sigma = DRUDE_SIGMA if DRUDE_SIGMA is not None else np.sqrt(KB * drude_temperature / K_DRUDE)
    pos[pair_drude] = pos[pair_parent] + rng.normal(0.0, sigma, (pair_drude.shape[0], 3))
    vel = np.zeros((n, 3))
    massive = mass > 0
    in_pair = np.zeros(n, bool)
    in_pair[pair_drude] = True
    in_pair[pair_parent] = True
    sel = massive & ~in_pair
    vel[sel] = rng.normal(0.0, 1.0, (int(sel.sum()), 3)) * np.sqrt(KB * temperature / mass[sel])[:, None]
    m1, m2 = mass[pair_drude], mass[pair_parent]
    mt, mu = m1 + m2, m1 * m2 / (m1 + m2)
    vcm = rng.normal(0.0, 1.0, (m1.shape[0], 3)) * np.sqrt(KB * temperature / mt)[:, None]
    vrel = rng.normal(0.0, 1.0, (m1.shape[0], 3)) * np.sqrt(KB * drude_temperature / mu)[:, None]
    vel[pair_drude] = vcm - vrel * (m2 / mt)[:, None]      # rel = v_parent - v_drude
    vel[pair_parent] = vcm + vrel * (m1 / mt)[:, None]
    sys_ = DrudeSystem(mass=mass, pair_drude=pair_drude, pair_parent=pair_parent, resid=resid,
                       constraints=constraints if constraints is not None else np.zeros((0, 2), np.int32),
                       has_cm_motion_remover=cmm, positions=pos, velocities=vel, name=name)
    return sys_, np.ascontiguousarray(group, np.int32), int(num_groups)


def _molecules(n_mol, table, spacing, origin=0.0):
    k = len(table["mass"])
    mass = np.tile(np.asarray(table["mass"], np.float64), n_mol)
    base = np.arange(n_mol, dtype=np.int64) * k
    centers = _lattice(n_mol, spacing) + origin
    pos = (centers[:, None, :] + np.asarray(table["geom"], np.float64)[None, :, :]).reshape(-1, 3)
    return mass, base + table["drude"], base + table["parent"], np.repeat(np.arange(n_mol), k), pos


def water_box(n_mol, order="test", temperature=300.0, drude_temperature=1.0, seed=SEED, spacing=0.31, rigid=False):
    """SWM4-NDP water: 5 slots per molecule (one massless M site), one Drude pair per molecule.
    rigid=True adds the reference test's three distance constraints per molecule and its M virtual site
    (TestReferenceDrudeTGNHIntegrator.cpp:145-148): O-H 0.09572, H-H 0.15139 nm, ThreeParticleAverageSite weights."""
    rng = np.random.default_rng(seed)
    table = _W_TEST if order == "test" else _W_NACL
    mass, pd, pp, resid, pos = _molecules(n_mol, table, spacing)
    out = _finish(mass, pd, pp, resid, pos, np.zeros(mass.shape[0], np.int32), 1, rng, temperature, drude_temperature,
                  f"swm4-{n_mol}" + ("-rigid" if rigid else ""))
    out[0].lattice = (5, int(np.ceil(n_mol ** (1.0 / 3.0) - 1e-9)), float(spacing), np.asarray(table["geom"], np.float64), 0)   # (_lattice, _molecules)
    if rigid:
        o, h1, h2, m = (0, 2, 3, 4) if order == "test" else (0, 1, 2, 3)
        base = np.arange(n_mol) * 5
        atoms = np.stack([base + o, base + h1, base + h2, -np.ones(n_mol, np.int64)], 1)
        dist = np.tile(np.array([0.09572, 0.09572, 0.0, 0.15139, 0.0, 0.0]), (n_mol, 1))
        out[0].set_clusters(atoms, dist)
        out[0].set_virtual_sites(np.stack([base + m, base + o, base + h1, base + h2], 1),
                                 np.tile(np.array([0.786646558, 0.106676721, 0.106676721]), (n_mol, 1)))
        out[0].positions[base + m] = (0.786646558 * out[0].positions[base + o] + 0.106676721 * out[0].positions[base + h1]
                                      + 0.106676721 * out[0].positions[base + h2])
    return out


def nacl(temperature=300.0, drude_temperature=1.0, seed=SEED):
    """Topology of example/nacl_tg.py: 492 SWM4 waters + 10 Na+ + 10 Cl- = 2500 slots, 512 pairs."""
    rng = np.random.default_rng(seed)
    mass, pd, pp, resid, pos = _molecules(492, _W_NACL, 0.31)
    n0, r0 = mass.shape[0], 492
    ion_mass = np.array([22.98977 - 0.4, 0.4] * 10 + [35.453 - 0.4, 0.4] * 10)      # nacl_tg.py:49-53
    ion_base = n0 + 2 * np.arange(20)
    ion_pos = np.repeat(_lattice(20, 0.6) + np.array([0.0, 0.0, 3.0]), 2, axis=0)
    mass = np.r_[mass, ion_mass]
    pd = np.r_[pd, ion_base + 1]
    pp = np.r_[pp, ion_base]
    resid = np.r_[resid, r0 + np.repeat(np.arange(20), 2)]
    pos = np.r_[pos, ion_pos]
    return _finish(mass, pd, pp, resid, pos, np.zeros(mass.shape[0], np.int32), 1, rng, temperature, drude_temperature,
                   "nacl-1m")


def _ion_tables(rng):
    # cation: 10 x (heavy, Drude) then 15 H; anion: 5 x (heavy, Drude)
    cat_mass = [12.011 - 0.4, 0.4] * 10 + [1.008] * 15
    ani_mass = [18.998 - 0.4, 0.4] * 5
    cat_geom = rng.uniform(-0.15, 0.15, (35, 3))
    ani_geom = rng.uniform(-0.08, 0.08, (10, 3))
    return np.asarray(cat_mass), cat_geom, np.asarray(ani_mass), ani_geom


def _ion_pairs(n_pairs, rng, origin, res0, slot0):
    cat_mass, cat_geom, ani_mass, ani_geom = _ion_tables(rng)
    centers = _lattice(2 * n_pairs, 0.6) + origin
    masses, poss, pds, pps, resids, kinds = [], [], [], [], [], []
    # interleave cation, anion, cation, ... (alternating residues)
    per = 45
    base = slot0 + np.arange(n_pairs, dtype=np.int64) * per
    mass = np.tile(np.r_[cat_mass, ani_mass], n_pairs)
    pos = np.empty((n_pairs * per, 3))
    pos.reshape(n_pairs, per, 3)[:, :35, :] = centers[0::2][:, None, :] + cat_geom[None]
    pos.reshape(n_pairs, per, 3)[:, 35:, :] = centers[1::2][:, None, :] + ani_geom[None]
    heavy_c = 2 * np.arange(10)
    heavy_a = 35 + 2 * np.arange(5)
    heavy = np.r_[heavy_c, heavy_a]
    pp = (base[:, None] + heavy[None, :]).reshape(-1)
    pd = pp + 1
    resid = res0 + (2 * np.arange(n_pairs)[:, None] + np.r_[np.zeros(35, np.int64), np.ones(10, np.int64)][None, :]).reshape(-1)
    kind = np.tile(np.r_[np.zeros(35, np.int32), np.ones(10, np.int32)], n_pairs)     # 0 cation, 1 anion
    return mass, pd, pp, resid, pos, kind


def ionic_liquid(n_pairs, temperature=300.0, drude_temperature=1.0, seed=SEED, constrained=False):
    """[BMIM][BF4]-like: cation 35 sites (10 heavy + 10 Drude + 15 H), anion 10 (5 heavy + 5 Drude).
    Two temperature groups: cations 0, anions 1.  Every ion is its own molecule.
    constrained=True: the 15 X-H bonds of every cation are distance constraints (H k bonded to heavy atom k mod 10),
    i.e. BASELINE.json's "[BMIM][BF4] ... + SHAKE constraints"."""
    rng = np.random.default_rng(seed)
    mass, pd, pp, resid, pos, kind = _ion_pairs(n_pairs, rng, 0.0, 0, 0)
    out = _finish(mass, pd, pp, resid, pos, kind, 2, rng, temperature, drude_temperature,
                  f"il-{n_pairs}" + ("-shake" if constrained else ""))
    if constrained:
        s = out[0]
        base = np.arange(n_pairs) * 45
        atoms, dist = [], []
        for h in range(10):
            hs = [20 + h] + ([30 + h] if h < 5 else [])
            a = np.stack([base + 2 * h] + [base + x for x in hs] + [-np.ones(n_pairs, np.int64)] * (3 - len(hs)), 1)
            d = np.zeros((n_pairs, 6))
            for k, x in enumerate(hs):
                d[:, k] = np.linalg.norm(s.positions[base + x] - s.positions[base + 2 * h], axis=1)
            atoms.append(a); dist.append(d)
        s.set_clusters(np.concatenate(atoms), np.concatenate(dist))
    return out


def mixed(n_water, n_pairs, temperature=300.0, drude_temperature=1.0, seed=SEED):
    """Mixed solvent + ions, four temperature groups: water 0, cations 1, anions 2, a tagged 10 % of waters 3."""
    rng = np.random.default_rng(seed)
    wm, wpd, wpp, wres, wpos = _molecules(n_water, _W_TEST, 0.31)
    wgroup = np.where(np.repeat(np.arange(n_water) % 10 == 9, 5), 3, 0).astype(np.int32)
    zoff = wpos[:, 0].max() + 1.0 if n_water else 0.0
    im, ipd, ipp, ires, ipos, kind = _ion_pairs(n_pairs, rng, np.array([zoff, 0.0, 0.0]), n_water, wm.shape[0])
    mass = np.r_[wm, im]
    group = np.r_[wgroup, kind + 1].astype(np.int32)
    return _finish(mass, np.r_[wpd, ipd], np.r_[wpp, ipp], np.r_[wres, ires], np.r_[wpos, ipos], group, 4, rng,
                   temperature, drude_temperature, f"mixed-{n_water}w-{n_pairs}ip")


def polymer_in_water(n_units, n_water, temperature=300.0, drude_temperature=1.0, seed=SEED):
    """One long polarizable chain molecule (n_units x [heavy, Drude, H] = 3 n_units slots in ONE molecule, the case of a
    protein or polymer: longer than a 512-slot tile when n_units > 170) followed by SWM4 waters.
    Temperature groups: polymer 1, water 0."""
    rng = np.random.default_rng(seed)
    pm = np.tile(np.array([12.011 - 0.4, 0.4, 1.008]), n_units)
    pbase = np.arange(n_units, dtype=np.int64) * 3
    ppos = np.repeat(np.stack([0.15 * np.arange(n_units), np.zeros(n_units), -1.0 * np.ones(n_units)], 1), 3, axis=0)
    ppos += np.tile(np.array([[0, 0, 0], [0, 0, 0], [0.0, 0.1, 0.0]]), (n_units, 1))
    wm, wpd, wpp, wres, wpos = _molecules(n_water, _W_TEST, 0.31)
    n0 = pm.shape[0]
    mass = np.r_[pm, wm]
    group = np.r_[np.ones(n0, np.int32), np.zeros(wm.shape[0], np.int32)]
    resid = np.r_[np.zeros(n0, np.int64), 1 + wres]
    return _finish(mass, np.r_[pbase + 1, n0 + wpd], np.r_[pbase, n0 + wpp], resid, np.r_[ppos, wpos], group, 2, rng,
                   temperature, drude_temperature, f"polymer{n_units}-water{n_water}")


def many_groups(n_water, n_pairs, num_groups, **kw):
    """The mixed system with `num_groups` temperature groups dealt out per molecule (molecule index mod num_groups)."""
    s, _, _ = mixed(n_water, n_pairs, **kw)
    return s, np.ascontiguousarray(s.resid % num_groups, np.int32), int(num_groups)


def single_pair():
    """platforms/reference/tests/TestReferenceDrudeTGNHIntegrator.cpp:54-83 (testSinglePair)."""
    s = DrudeSystem(mass=np.array([1.0, 0.1]), pair_drude=np.array([1]), pair_parent=np.array([0]),
                    resid=np.array([0, 0]), positions=np.array([[0, 0, 0], [0, 0, 0.01]], np.float64),
                    velocities=np.array([[1, 0, 0], [1, 0, 0.01]], np.float64), name="single-pair")
    return s, np.zeros(2, np.int32), 1


def pair_normal_massless(seed=SEED):
    """One Drude pair, one ordinary particle and one massless site, in two molecules."""
    rng = np.random.default_rng(seed)
    mass = np.array([15.6, 0.4, 12.0, 0.0])
    pos = np.array([[0, 0, 0], [0, 0, 0], [0.5, 0, 0], [0.5, 0.1, 0]], np.float64)
    return _finish(mass, np.array([1]), np.array([0]), np.array([0, 0, 1, 1]), pos, np.zeros(4, np.int32), 1, rng,
                   300.0, 1.0, "pair+normal+massless")
