"""The system of the reference's testWater (platforms/reference/tests/TestReferenceDrudeTGNHIntegrator.cpp:111-166):
6x6x6 SWM4-NDP waters on a 0.6 nm lattice in a 4.2 nm periodic box, three distance constraints and one
ThreeParticleAverageSite per molecule, CMMotionRemover, velocities zero."""
import numpy as np

from openmm_drudenose_amd import DrudeSystem, DrudeTGNHIntegrator

GRID, SPACING = 6, 0.6
BOX = SPACING * (GRID + 1)
W = np.array([0.786646558, 0.106676721, 0.106676721])


def build():
    n_mol = GRID ** 3
    pos = []
    for i in range(GRID):
        for j in range(GRID):
            for k in range(GRID):
                p = np.array([i * SPACING, j * SPACING, k * SPACING])
                pos += [p, p, p + [0.09572, 0, 0], p + [-0.023999, 0.092663, 0], p]       # test :152-160
    pos = np.array(pos)
    base = np.arange(n_mol) * 5
    s = DrudeSystem(mass=np.tile([15.6, 0.4, 1.0, 1.0, 0.0], n_mol), pair_drude=base + 1, pair_parent=base,
                    resid=np.repeat(np.arange(n_mol), 5), has_cm_motion_remover=True, positions=pos,
                    velocities=np.zeros_like(pos), name="reference testWater")
    s.set_clusters(np.stack([base, base + 2, base + 3, -np.ones(n_mol, np.int64)], 1),
                   np.tile([0.09572, 0.09572, 0.0, 0.15139, 0.0, 0.0], (n_mol, 1)))               # test :143-145
    s.set_virtual_sites(np.stack([base + 4, base, base + 2, base + 3], 1), np.tile(W, (n_mol, 1)))   # test :146
    s.positions[base + 4] = W[0] * pos[base] + W[1] * pos[base + 2] + W[2] * pos[base + 3]
    return s


def integrator(use_drude_nh_chains=False):
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.0005, 20, 10, use_drude_nh_chains)          # test :166
    it.setMaxDrudeDistance(0.05)                                                                   # test :167
    return it


def expected_temperature(s):
    """test :186-190"""
    n_mol = s.num_particles // 5
    std = 3 * 3 * n_mol - len(s.constraints) - 3
    drude = 3 * n_mol
    return (std * 300.0 + drude * 1.0) / (std + drude), std + drude
