"""ctypes wrapper of oracle/libtgnh_oracle.so (oracle/tgnh_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libtgnh_oracle.so")
MODE_DUALNH, MODE_TGNH = 0, 1
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int)


class _Desc(C.Structure):
    _fields_ = [
        ("mode", C.c_int), ("num_particles", C.c_int), ("num_pairs", C.c_int), ("num_groups", C.c_int),
        ("num_residues", C.c_int), ("num_constraints", C.c_int), ("has_cm_motion_remover", C.c_int),
        ("mass", _f64p), ("pair_drude", _i32p), ("pair_parent", _i32p), ("group", _i32p), ("resid", _i32p),
        ("constraint_i", _i32p), ("constraint_j", _i32p),
        ("kB", C.c_double), ("temperature", C.c_double), ("coupling_time", C.c_double),
        ("drude_temperature", C.c_double), ("drude_coupling_time", C.c_double), ("step_size", C.c_double),
        ("drude_steps_per_real_step", C.c_int), ("num_nh_chains", C.c_int),
        ("use_drude_nh_chains", C.c_int), ("use_com_temp_group", C.c_int), ("max_drude_distance", C.c_double),
    ]


def build_oracle(force=False):
    src = [os.path.join(HERE, f) for f in ("tgnh_oracle.c", "tgnh_oracle.h", "water_ff.c", "Makefile")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s", "libtgnh_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB


_lib = None
_mut = None


def load_mutants():
    """libtgnh_oracle_mut.so: the oracle with its deliberate, switchable mis-restatements (tgo_set_mutant).  Only
    tests/pin_sensitivity.py uses it -- to measure what the reference's own checks can and cannot see."""
    global _mut
    if _mut is None:
        path = os.path.join(HERE, "libtgnh_oracle_mut.so")
        subprocess.run(["make", "-C", HERE, "-s", "libtgnh_oracle_mut.so"], check=True, stdout=subprocess.DEVNULL)
        _mut = _bind(C.CDLL(path))
        _mut.tgo_set_mutant.argtypes = [C.c_int]
    return _mut


def _load():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = _bind(C.CDLL(LIB))
    return _lib


def _bind(L):
    L.tgo_last_error.restype = C.c_char_p
    L.tgo_create.argtypes = [C.POINTER(_Desc), C.POINTER(C.c_void_p)]
    L.tgo_destroy.argtypes = [C.c_void_p]
    L.tgo_set_step_size.argtypes = [C.c_void_p, C.c_double]
    L.tgo_set_drude_steps.argtypes = [C.c_void_p, C.c_int]
    L.tgo_set_max_drude_distance.argtypes = [C.c_void_p, C.c_double]
    L.tgo_num_normal.argtypes = [C.c_void_p]
    L.tgo_get_normal.argtypes = [C.c_void_p, _i32p]
    L.tgo_num_thermostats.argtypes = [C.c_void_p]
    L.tgo_get_dof.argtypes = [C.c_void_p, _f64p, _f64p]
    L.tgo_chain_len.argtypes = [C.c_void_p, C.c_int]
    L.tgo_get_chain.argtypes = [C.c_void_p, C.c_int, _f64p]
    L.tgo_set_chain.argtypes = [C.c_void_p, C.c_int, _f64p]
    L.tgo_kinetic_energies.argtypes = [C.c_void_p, _f64p, _f64p]
    L.tgo_propagate_nhc.argtypes = [C.c_void_p, _f64p, _f64p, _f64p]
    L.tgo_chain_only.argtypes = [C.c_void_p, _f64p, _f64p]
    L.tgo_scale_velocities.argtypes = [C.c_void_p, _f64p, _f64p]
    L.tgo_half_kick.argtypes = [C.c_void_p, _f64p, _f64p]
    L.tgo_drift.argtypes = [C.c_void_p, _f64p, _f64p]
    L.tgo_hardwall.argtypes = [C.c_void_p, _f64p, _f64p]
    L.tgo_step_begin.argtypes = [C.c_void_p, _f64p, _f64p, _f64p]
    L.tgo_step_end.argtypes = [C.c_void_p, _f64p, _f64p]
    L.tgo_kinetic_energy_query.argtypes = [C.c_void_p, _f64p, _f64p, C.c_int]
    L.tgo_kinetic_energy_query.restype = C.c_double
    L.tgo_harness_force.argtypes = [C.c_void_p, _f64p, _f64p, C.c_double, C.c_double, _f64p]
    L.tgo_run_harness.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, _f64p, C.c_double, C.c_double, C.c_int]
    L.tgo_set_clusters.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _i32p, _f64p]
    L.tgo_shake_positions.argtypes = [C.c_void_p, _f64p, _f64p, C.c_double]
    L.tgo_shake_velocities.argtypes = [C.c_void_p, _f64p, _f64p, C.c_double]
    L.tgo_set_virtual_sites.argtypes = [C.c_void_p, C.c_int, _i32p, _f64p]
    L.tgo_virtual_sites.argtypes = [C.c_void_p, _f64p]
    L.tgo_run_harness_constrained.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, _f64p, C.c_double, C.c_double, C.c_double, C.c_int]
    L.tgo_water_forces.argtypes = [C.c_int, _f64p, C.c_double, C.c_double, _f64p]
    L.tgo_water_forces.restype = C.c_double
    L.tgo_time.argtypes = [C.c_void_p]
    L.tgo_time.restype = C.c_double
    L.tgo_step_count.argtypes = [C.c_void_p]
    L.tgo_step_count.restype = C.c_long
    return L


class OracleError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


def _p(a):
    return a.ctypes.data_as(_f64p)


class Oracle:
    """One oracle integrator over a DrudeSystem-like object (mass, pair_drude, pair_parent, resid, constraints)."""

    def __init__(self, system, group, num_groups, mode, temperature, coupling_time, drude_temperature,
                 drude_coupling_time, step_size, drude_steps=20, num_nh_chains=1, use_drude_nh_chains=False,
                 use_com_temp_group=True, max_drude_distance=0.0, kB=8.31446261815324e-3, lib=None):
        L = lib if lib is not None else _load()
        self.L = L
        self._keep = [np.ascontiguousarray(system.mass, np.float64),
                      np.ascontiguousarray(system.pair_drude, np.int32),
                      np.ascontiguousarray(system.pair_parent, np.int32),
                      np.ascontiguousarray(group, np.int32), np.ascontiguousarray(system.resid, np.int32)]
        d = _Desc()
        d.mode = mode
        d.num_particles, d.num_pairs = len(self._keep[0]), len(self._keep[1])
        d.num_groups, d.num_residues = num_groups, int(self._keep[4].max()) + 1
        cons = np.ascontiguousarray(getattr(system, "constraints", np.zeros((0, 2))), np.int32).reshape(-1, 2)
        d.num_constraints = len(cons)
        d.has_cm_motion_remover = int(getattr(system, "has_cm_motion_remover", False))
        d.mass = _p(self._keep[0])
        d.pair_drude = self._keep[1].ctypes.data_as(_i32p)
        d.pair_parent = self._keep[2].ctypes.data_as(_i32p)
        d.group = self._keep[3].ctypes.data_as(_i32p)
        d.resid = self._keep[4].ctypes.data_as(_i32p)
        if len(cons):
            self._ci, self._cj = np.ascontiguousarray(cons[:, 0]), np.ascontiguousarray(cons[:, 1])
            d.constraint_i, d.constraint_j = self._ci.ctypes.data_as(_i32p), self._cj.ctypes.data_as(_i32p)
        d.kB = kB
        d.temperature, d.coupling_time = temperature, coupling_time
        d.drude_temperature, d.drude_coupling_time = drude_temperature, drude_coupling_time
        d.step_size, d.drude_steps_per_real_step, d.num_nh_chains = step_size, drude_steps, num_nh_chains
        d.use_drude_nh_chains, d.use_com_temp_group = int(use_drude_nh_chains), int(use_com_temp_group)
        d.max_drude_distance = max_drude_distance
        h = C.c_void_p()
        rc = L.tgo_create(C.byref(d), C.byref(h))
        if rc != 0:
            raise OracleError(rc, L.tgo_last_error().decode())
        self.h = h
        self.n = d.num_particles
        self.mode = mode
        ca = getattr(system, "cluster_atoms", None)
        if ca is not None and len(ca):
            self.set_clusters(ca, system.cluster_dist)
        sa = getattr(system, "site_atoms", None)
        if sa is not None and len(sa):
            self.set_virtual_sites(sa, system.site_weights)

    @classmethod
    def from_integrator(cls, system, integ, group, num_groups, mode, **kw):
        return cls(system, group, num_groups, mode, integ.getTemperature(), integ.getCouplingTime(),
                   integ.getDrudeTemperature(), integ.getDrudeCouplingTime(), integ.getStepSize(),
                   integ.getDrudeStepsPerRealStep(), integ.getNumNHChains(), bool(integ.getUseDrudeNHChains()),
                   bool(integ.getUseCOMTempGroup()), integ.getMaxDrudeDistance(), **kw)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.tgo_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(rc, self.L.tgo_last_error().decode())

    def set_step_size(self, dt): self.L.tgo_set_step_size(self.h, dt)
    def set_drude_steps(self, n): self.L.tgo_set_drude_steps(self.h, n)
    def set_max_drude_distance(self, d): self.L.tgo_set_max_drude_distance(self.h, d)

    def normal_particles(self):
        out = np.zeros(self.L.tgo_num_normal(self.h), np.int32)
        self.L.tgo_get_normal(self.h, out.ctypes.data_as(_i32p))
        return out

    def num_thermostats(self):
        return self.L.tgo_num_thermostats(self.h)

    def dof(self):
        n = self.num_thermostats()
        dof, nkt = np.zeros(n), np.zeros(n)
        self.L.tgo_get_dof(self.h, _p(dof), _p(nkt))
        return dof, nkt

    def chain(self, which):
        out = np.zeros(self.L.tgo_chain_len(self.h, which))
        self.L.tgo_get_chain(self.h, which, _p(out))
        return out

    def set_chain(self, which, arr):
        arr = np.ascontiguousarray(arr, np.float64)
        assert len(arr) == self.L.tgo_chain_len(self.h, which)
        self.L.tgo_set_chain(self.h, which, _p(arr))

    def kinetic_energies(self, vel):
        ke = np.zeros(self.num_thermostats())
        self.L.tgo_kinetic_energies(self.h, _p(vel), _p(ke))
        return ke

    def propagate_nhc(self, vel):
        n = self.num_thermostats()
        ke, sc = np.zeros(n), np.zeros(n)
        self.L.tgo_propagate_nhc(self.h, _p(vel), _p(ke), _p(sc))
        return ke, sc

    def chain_only(self, ke):
        sc = np.zeros(self.num_thermostats())
        self.L.tgo_chain_only(self.h, _p(np.ascontiguousarray(ke, np.float64)), _p(sc))
        return sc

    def scale_velocities(self, vel, scale):
        self.L.tgo_scale_velocities(self.h, _p(vel), _p(np.ascontiguousarray(scale, np.float64)))

    def half_kick(self, vel, force): self.L.tgo_half_kick(self.h, _p(vel), _p(force))
    def drift(self, pos, vel): self.L.tgo_drift(self.h, _p(pos), _p(vel))
    def hardwall(self, pos, vel): self._chk(self.L.tgo_hardwall(self.h, _p(pos), _p(vel)))
    def step_begin(self, pos, vel, force): self._chk(self.L.tgo_step_begin(self.h, _p(pos), _p(vel), _p(force)))
    def step_end(self, vel, force): self._chk(self.L.tgo_step_end(self.h, _p(vel), _p(force)))

    def kinetic_energy_query(self, vel, force, valid):
        return self.L.tgo_kinetic_energy_query(self.h, _p(vel), _p(force), int(valid))

    def harness_force(self, pos, x0, k_drude, k_tether):
        f = np.zeros((self.n, 3))
        self.L.tgo_harness_force(self.h, _p(pos), _p(x0), k_drude, k_tether, _p(f))
        return f

    def run_harness(self, pos, vel, force, x0, k_drude, k_tether, nsteps):
        self._chk(self.L.tgo_run_harness(self.h, _p(pos), _p(vel), _p(force), _p(x0), k_drude, k_tether, nsteps))

    PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))

    def set_clusters(self, atoms, dist6):
        """Clusters in the library's canonical form ([K,4] atoms, [K,6] distances, 0 = none) -> the oracle's
        general (pairs, distances) lists, kept in the same sweep order."""
        atoms = np.ascontiguousarray(atoms, np.int32).reshape(-1, 4)
        dist6 = np.asarray(dist6, np.float64).reshape(-1, 6)
        k = len(atoms)
        ncons = np.zeros(k, np.int32)
        pairs = np.zeros((k, 6, 2), np.int32)
        dist = np.zeros((k, 6))
        for c in range(k):
            n = 0
            for p, (a, b) in enumerate(self.PAIRS):
                if dist6[c, p] > 0:
                    pairs[c, n] = (a, b); dist[c, n] = dist6[c, p]; n += 1
            ncons[c] = n
        self._cl = (atoms, ncons, np.ascontiguousarray(pairs), np.ascontiguousarray(dist))
        self.L.tgo_set_clusters(self.h, k, atoms.ctypes.data_as(_i32p), ncons.ctypes.data_as(_i32p),
                                self._cl[2].ctypes.data_as(_i32p), _p(self._cl[3]))

    def set_virtual_sites(self, atoms, w):
        atoms = np.ascontiguousarray(atoms, np.int32).reshape(-1, 4)
        w = np.ascontiguousarray(w, np.float64).reshape(-1, 3)
        self._vs = (atoms, w)
        self.L.tgo_set_virtual_sites(self.h, len(atoms), atoms.ctypes.data_as(_i32p), _p(w))

    def shake_positions(self, pos, delta, tol): self._chk(self.L.tgo_shake_positions(self.h, _p(pos), _p(delta), tol))
    def shake_velocities(self, pos, vel, tol): self._chk(self.L.tgo_shake_velocities(self.h, _p(pos), _p(vel), tol))
    def virtual_sites(self, pos): self.L.tgo_virtual_sites(self.h, _p(pos))

    def run_harness_constrained(self, pos, vel, force, x0, k_drude, k_tether, tol, nsteps):
        self._chk(self.L.tgo_run_harness_constrained(self.h, _p(pos), _p(vel), _p(force), _p(x0), k_drude, k_tether, tol, nsteps))

    def time(self): return self.L.tgo_time(self.h)
    def step_count(self): return self.L.tgo_step_count(self.h)


def water_forces(pos, box, cutoff=1.0):
    """Forces (and energy) of the reference's testWater force field (oracle/water_ff.c) for N/5 SWM4 molecules."""
    L = _load()
    pos = np.ascontiguousarray(pos, np.float64)
    f = np.zeros_like(pos)
    e = L.tgo_water_forces(pos.shape[0] // 5, _p(pos), box, cutoff, _p(f))
    return f, e
