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


def extended_energy(system, normal, pos, vel, x0, nkt, eta, eta_dot, eta_mass, chains, kT, kT_drude, mode,
                    k_drude=synth.K_DRUDE, k_tether=synth.K_TETHER):
    """The quantity Nose-Hoover-chain dynamics conserves, for the harness force field:
        H = 1/2 sum m v^2 + U + sum_t [ sum_i 1/2 Q_ti etaDot_ti^2 + NkT_t eta_t0 + kT_t sum_{i>=1} eta_ti ].
    The reference never evaluates it (it is the invariant SURVEY 8c(3) asks the build to add); a restatement whose
    chain, KE partition or rescale were wrong in sign, factor or coupling would drift linearly instead of
    fluctuating at O(dt^2).  Thermostat arrays: TGNH [thermostat][link] (etaDot rows of C+1); dualNH the reference's
    interleaved [real0, drude0, real1, drude1, ...] (Ref :186-217), valid with useDrudeNHChains only."""
    m = system.mass
    ke = 0.5 * float((m[:, None] * vel ** 2).sum())
    tether = np.zeros(len(m), bool)
    tether[normal] = True
    tether[system.pair_parent] = True
    tether &= m > 0
    sep = pos[system.pair_drude] - pos[system.pair_parent]
    pe = 0.5 * k_tether * float(((pos - x0)[tether] ** 2).sum()) + 0.5 * k_drude * float((sep ** 2).sum())
    C = chains
    if mode == "dualNH":
        eta, eta_dot, eta_mass = (np.asarray(a)[:2 * C].reshape(C, 2).T for a in (eta, eta_dot, eta_mass))
    else:
        nt = len(nkt)
        eta, eta_dot, eta_mass = (np.asarray(a).reshape(nt, -1)[:, :C] for a in (eta, eta_dot, eta_mass))
    th = 0.0
    for t in range(len(nkt)):
        if eta_mass[t, 0] <= 0:                       # inert thermostat (Cu :561 etaMass > 0 guard)
            continue
        kt = kT_drude if t == len(nkt) - 1 else kT
        th += 0.5 * float((eta_mass[t] * eta_dot[t] ** 2).sum()) + nkt[t] * eta[t, 0] + kt * float(eta[t, 1:].sum())
    return ke + pe + th, ke, th


def random_topology(seed):
    """Ragged test input: molecules of 1-40 slots (every third seed: two longer than a tile), Drude pairs anywhere inside
    a molecule (Drude before or after its parent, up to 30 slots apart), massless sites, 1-6 temperature groups assigned
    per molecule, a few constraints inside molecules."""
    rng = np.random.default_rng(1000 + seed)
    sizes = rng.integers(1, 41, size=rng.integers(40, 400))
    if seed % 3 == 0:
        sizes[rng.integers(0, len(sizes), 2)] = rng.integers(600, 1500, 2)
    n = int(sizes.sum())
    resid = np.repeat(np.arange(len(sizes)), sizes).astype(np.int32)
    first = np.r_[0, np.cumsum(sizes)[:-1]]
    mass = rng.uniform(1.0, 40.0, n)
    ngroups = int(rng.integers(1, 7))
    group = np.repeat(rng.integers(0, ngroups, len(sizes)), sizes).astype(np.int32)
    pd, pp, used = [], [], np.zeros(n, bool)
    for f, sz in zip(first, sizes):
        for _ in range(int(rng.integers(0, max(1, sz // 2) + 1))):
            a = f + int(rng.integers(0, sz))
            b = a + int(rng.integers(-30, 31))
            if b == a or b < f or b >= f + sz or used[a] or used[b]:
                continue
            used[a] = used[b] = True
            pd.append(a); pp.append(b)
            mass[a] = 0.4
    free = np.flatnonzero(~used)
    mass[rng.choice(free, size=min(len(free), n // 20), replace=False)] = 0.0
    for f, sz in zip(first, sizes):                      # every molecule keeps a massive particle (else v_com is 0/0: rejected)
        if not mass[f:f + sz].any():
            mass[f] = 12.0
    cons = []
    for f, sz in zip(first, sizes):
        if sz >= 3 and rng.random() < 0.3:
            i, j = f + rng.choice(sz, 2, replace=False)
            if mass[i] > 0 and mass[j] > 0:
                cons.append((i, j))
    return mass, pd, pp, resid, group, ngroups, cons, sizes, first, rng


def random_clusters(rng, mass, pd, pp, sizes, first, pos):
    """Constraint clusters for a random_topology: in about half of the molecules, 2-4 massive atoms outside Drude pairs are
    pulled to within ~0.1 nm of one another and held at those distances -- all pairs (a rigid body, up to 3 atoms) or the bonds from
    the first atom only.  -> (cluster_atoms [K,4], cluster_dist [K,6]) in DrudeSystem's canonical pair order."""
    PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))
    in_pair = np.zeros(len(mass), bool)
    in_pair[pd] = True
    in_pair[pp] = True
    atoms, dist = [], []
    for f, sz in zip(first, sizes):
        cand = [i for i in range(f, f + sz) if mass[i] > 0 and not in_pair[i]]
        if len(cand) < 2 or rng.random() < 0.5:
            continue
        c = int(rng.integers(2, min(4, len(cand)) + 1))
        pick = rng.choice(cand, c, replace=False)
        for k in range(1, c):
            pos[pick[k]] = pos[pick[0]] + rng.normal(0.0, 0.06, 3) + np.array([0.08, 0.0, 0.0])
        rigid = c <= 3 and rng.random() < 0.5            # (a random rigid tetrahedron can take SHAKE > 500 sweeps at 1e-10)
        a = [-1] * 4
        a[:c] = [int(x) for x in pick]
        d = [0.0] * 6
        for k, (i, j) in enumerate(PAIRS):
            if j < c and (rigid or i == 0):
                d[k] = float(np.linalg.norm(pos[pick[i]] - pos[pick[j]]))
        atoms.append(a)
        dist.append(d)
    return np.array(atoms, np.int32).reshape(-1, 4), np.array(dist, np.float64).reshape(-1, 6)


# ---- topologies the tiled kernels cannot hold: they step on the gather path (tgnh_gather.hip; tests/test_gather_gpu.py) ----
def drudes_at_the_end(n_mol, seed=3):
    """A water box as some builders write it: all atoms first (O, H, H, M per molecule), all Drude particles appended behind them.
    Every Drude is far from its parent, and every residue comes in two runs."""
    s, _, _ = synth.water_box(n_mol)
    d = np.asarray(s.pair_drude)
    keep = np.setdiff1d(np.arange(s.num_particles), d)
    order = np.r_[keep, d]                                    # new -> old
    inv = np.empty_like(order); inv[order] = np.arange(len(order))
    out = synth.DrudeSystem(mass=s.mass[order], pair_drude=inv[s.pair_drude].astype(np.int32), pair_parent=inv[s.pair_parent].astype(np.int32),
                            resid=s.resid[order], positions=s.positions[order], velocities=s.velocities[order], name="drudes-at-the-end")
    return out, np.zeros(out.num_particles, np.int32), 1


def onion(n=700, seed=5):
    """One molecule whose pairs nest like onion skins (i, n - 1 - i): every cut between slot 1 and n - 1 goes through a pair."""
    rng = np.random.default_rng(seed)
    mass = rng.uniform(8.0, 20.0, n)
    pd, pp = np.arange(n // 2, dtype=np.int32), (n - 1 - np.arange(n // 2)).astype(np.int32)
    mass[pd] = 0.4
    pos = rng.uniform(0.0, 2.0, (n, 3))
    pos[pd] = pos[pp] + rng.normal(0.0, 0.002, (n // 2, 3))
    return synth._finish(mass, pd, pp, np.zeros(n, np.int32), pos, np.zeros(n, np.int32), 1, rng, 300.0, 1.0, "onion")


def far_pairs(seed=7):
    """Water-like molecules plus one 1400-slot molecule whose Drude particles sit 600-1200 slots from their parents."""
    rng = np.random.default_rng(seed)
    s, g, ng = synth.mixed(200, 10)
    n0, nb = s.num_particles, 1400
    mass = np.r_[s.mass, rng.uniform(6.0, 30.0, nb)]
    parents = n0 + np.arange(0, 100)
    drudes = n0 + nb - 1 - np.arange(0, 100) * 2
    mass[drudes] = 0.4
    resid = np.r_[s.resid, np.full(nb, s.resid.max() + 1)].astype(np.int32)
    group = np.r_[g, np.full(nb, 1)].astype(np.int32)
    pos = np.r_[s.positions, rng.uniform(0.0, 4.0, (nb, 3))]
    pos[drudes] = pos[parents] + rng.normal(0.0, 0.002, (100, 3))
    out = synth._finish(mass, np.r_[s.pair_drude, drudes].astype(np.int32), np.r_[s.pair_parent, parents].astype(np.int32), resid, pos,
                        group, ng, rng, 300.0, 1.0, "far-pairs")
    return out


def interleaved(n_mol=60):
    """Two particles of neighbouring molecules swapped: residues 0 and 1 each come in several runs.  The reference's table says
    (count, start of the LAST run) for them (Cu :121-124) and its COM kernel walks `count` particles from there (K :90-91),
    whoever they belong to -- reproduced as it is, here and in the oracle (well inside the array for these two)."""
    s, g, ng = synth.water_box(n_mol)
    resid = s.resid.copy()
    resid[[1, 6]] = resid[[6, 1]]
    out = synth.DrudeSystem(mass=s.mass, pair_drude=s.pair_drude, pair_parent=s.pair_parent, resid=resid, positions=s.positions,
                            velocities=s.velocities, name="interleaved")
    return out, g, ng


