"""Plain description of what the integrator needs from an OpenMM System + DrudeForce."""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class DrudeSystem:
    """The subset of OpenMM's System / DrudeForce / Context::getMolecules that
    DrudeTGNHIntegrator::initialize reads (openmmapi/src/DrudeTGNHIntegrator.cpp:103-160)."""
    mass: np.ndarray                       # [N] System::getParticleMass
    pair_drude: np.ndarray                 # [P] DrudeForce p   (Drude particle)
    pair_parent: np.ndarray                # [P] DrudeForce p1  (parent)
    resid: np.ndarray                      # [N] molecule index (Context::getMolecules)
    constraints: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.int32))
    has_cm_motion_remover: bool = False
    positions: np.ndarray = None           # [N,3] nm
    velocities: np.ndarray = None          # [N,3] nm/ps
    name: str = ""
    # harness call-out data (what OpenMM's constraint / virtual-site machinery would hold)
    cluster_atoms: np.ndarray = None       # [K,4] slot indices, -1 = unused
    cluster_dist: np.ndarray = None        # [K,6] distances for pairs (0,1)(0,2)(0,3)(1,2)(1,3)(2,3), 0 = none
    site_atoms: np.ndarray = None          # [S,4] (site, p1, p2, p3)
    site_weights: np.ndarray = None        # [S,3]
    # harness only: the box repeats ONE molecule on a simple cubic lattice (synth.water_box) -- (slots per molecule, side, spacing nm,
    # geometry [slots][3], index of this system's first molecule in the box); a hint for the harness force's tether sites, checked
    # slot by slot by the library before it is used
    lattice: tuple = None

    def __post_init__(self):
        self.mass = np.ascontiguousarray(self.mass, np.float64)
        self.pair_drude = np.ascontiguousarray(self.pair_drude, np.int32)
        self.pair_parent = np.ascontiguousarray(self.pair_parent, np.int32)
        self.resid = np.ascontiguousarray(self.resid, np.int32)
        self.constraints = np.ascontiguousarray(self.constraints, np.int32).reshape(-1, 2)
        if self.cluster_atoms is not None:
            self.set_clusters(self.cluster_atoms, self.cluster_dist)

    PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))

    def set_clusters(self, atoms, dist):
        """Constraint clusters; System::getConstraintParameters is derived from them (dof bookkeeping)."""
        self.cluster_atoms = np.ascontiguousarray(atoms, np.int32).reshape(-1, 4)
        self.cluster_dist = np.ascontiguousarray(dist, np.float64).reshape(-1, 6)
        cons = []
        for k, (a, b) in enumerate(self.PAIRS):
            m = self.cluster_dist[:, k] > 0
            cons.append(np.stack([self.cluster_atoms[m, a], self.cluster_atoms[m, b]], 1))
        self.constraints = np.ascontiguousarray(np.concatenate(cons), np.int32).reshape(-1, 2)

    def set_virtual_sites(self, atoms, weights):
        self.site_atoms = np.ascontiguousarray(atoms, np.int32).reshape(-1, 4)
        self.site_weights = np.ascontiguousarray(weights, np.float64).reshape(-1, 3)

    @property
    def num_particles(self):
        return int(self.mass.shape[0])

    @property
    def num_pairs(self):
        return int(self.pair_drude.shape[0])

    @property
    def num_residues(self):
        return int(self.resid.max()) + 1 if self.resid.size else 0

    def slice_molecules(self, lo, hi):
        """Sub-system of the particle slots [lo, hi) (whole molecules), for particle sharding."""
        sel = np.arange(lo, hi)
        inv = -np.ones(self.num_particles, np.int64)
        inv[sel] = np.arange(hi - lo)
        pm = (self.pair_drude >= lo) & (self.pair_drude < hi)
        if not np.all(((self.pair_parent >= lo) & (self.pair_parent < hi)) == pm):
            raise ValueError("shard boundary cuts a Drude pair")
        cm = np.zeros(len(self.constraints), bool)
        if len(self.constraints):
            cm = (self.constraints[:, 0] >= lo) & (self.constraints[:, 0] < hi)
        r = self.resid[lo:hi]
        lat = None
        if self.lattice is not None and lo % self.lattice[0] == 0:
            lat = self.lattice[:4] + (self.lattice[4] + lo // self.lattice[0],)
        return DrudeSystem(
            lattice=lat,
            mass=self.mass[lo:hi], pair_drude=inv[self.pair_drude[pm]], pair_parent=inv[self.pair_parent[pm]],
            resid=r - r.min() if r.size else r, constraints=inv[self.constraints[cm]] if cm.any() else np.zeros((0, 2), np.int32),
            has_cm_motion_remover=self.has_cm_motion_remover,
            positions=None if self.positions is None else self.positions[lo:hi].copy(),
            velocities=None if self.velocities is None else self.velocities[lo:hi].copy(),
            name=self.name + f"[{lo}:{hi}]")


def shard_bounds(system, world_size):
    """Contiguous slabs of whole molecules balanced by slot count."""
    n = system.num_particles
    # molecule starts: positions where resid changes
    starts = np.flatnonzero(np.r_[True, system.resid[1:] != system.resid[:-1]])
    bounds = [0]
    for r in range(1, world_size):
        target = n * r / world_size
        k = int(np.searchsorted(starts, target))
        cand = starts[min(k, len(starts) - 1)]
        if k > 0 and abs(starts[k - 1] - target) <= abs(cand - target):
            cand = starts[k - 1]
        bounds.append(int(max(cand, bounds[-1])))
    bounds.append(n)
    return bounds
