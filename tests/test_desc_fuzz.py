"""Malformed descriptors at the boundary (CPU, host-only handles): whatever a binding hands to tgnh_create -- indices out of
range, a particle in two pairs, groups / residues that do not exist, residues in several runs, masses of 0 / negative / NaN /
inf, counts of zero or absurd size, chain lengths and sub-step counts outside what the kernels take -- the library answers with
a handle or with TGNH_ERR_ARG / TGNH_ERR_UNSUPPORTED / TGNH_ERR_GROUP_MISMATCH (the reference's own check, Ref :128-131) and a message, and never crashes; a handle it does give answers its queries
and is destroyed cleanly.  The reference throws OpenMMException for what it checks (DrudeTGNHIntegrator.cpp:98-99, :110-124) and
reads out of bounds for what it does not (ReferenceDrudeTGNHKernels.cpp:124-137 takes pair indices as they come); here the
boundary checks.  tools/sanitize/run_host_asan.sh runs this file against an AddressSanitizer + UBSan build of the host code."""
import ctypes as C

import numpy as np
import pytest

from openmm_drudenose_amd import _lib, DrudeTGNHIntegrator
from helpers import random_topology

OK_CODES = None


def base_desc(k, keep):
    mass, pd, pp, resid, group, ngroups, cons, sizes, first, rng = random_topology(k)
    n = len(mass)
    a = dict(mass=np.ascontiguousarray(mass, np.float64), pd=np.ascontiguousarray(pd, np.int32), pp=np.ascontiguousarray(pp, np.int32),
             group=np.ascontiguousarray(group, np.int32), resid=np.ascontiguousarray(resid, np.int32),
             ci=np.ascontiguousarray([c[0] for c in cons], np.int32), cj=np.ascontiguousarray([c[1] for c in cons], np.int32))
    d = _lib.TgnhDesc()
    d.struct_size = C.sizeof(_lib.TgnhDesc)
    d.mode, d.precision, d.flags, d.device = _lib.MODE_TGNH, _lib.PREC_MIXED, 0, -1
    d.num_particles, d.padded_num_particles = n, (n + 31) // 32 * 32
    d.num_pairs, d.num_groups, d.num_residues = len(pd), ngroups, int(resid.max()) + 1
    d.num_constraints = len(cons)
    d.kB, d.temperature, d.coupling_time, d.drude_temperature, d.drude_coupling_time = 8.314462618e-3, 300.0, 0.1, 1.0, 0.005
    d.step_size, d.drude_steps_per_real_step, d.num_nh_chains = 0.001, 20, 3
    d.use_drude_nh_chains, d.use_com_temp_group, d.max_drude_distance = 1, 1, 0.02
    keep.append(a)
    return d, a, rng


def point(d, a):
    d.mass = a["mass"].ctypes.data_as(_lib.c_f64p)
    d.pair_drude = a["pd"].ctypes.data_as(_lib.c_i32p)
    d.pair_parent = a["pp"].ctypes.data_as(_lib.c_i32p)
    d.group = a["group"].ctypes.data_as(_lib.c_i32p)
    d.resid = a["resid"].ctypes.data_as(_lib.c_i32p)
    d.constraint_i = a["ci"].ctypes.data_as(_lib.c_i32p) if len(a["ci"]) else None
    d.constraint_j = a["cj"].ctypes.data_as(_lib.c_i32p) if len(a["cj"]) else None
    for name in a.get("null", ()):
        setattr(d, name, None)


def mutate(d, a, rng):
    """one to three random corruptions; returns their names"""
    n, done = d.num_particles, []
    for _ in range(int(rng.integers(1, 4))):
        m = int(rng.integers(0, 24))
        done.append(m)
        idx = lambda arr: int(rng.integers(0, max(len(arr), 1)))          # noqa: E731
        if m == 0 and len(a["pd"]):
            a["pd"][idx(a["pd"])] = int(rng.choice([-1, n, n + 7, -2**31, 2**31 - 1]))
        elif m == 1 and len(a["pp"]):
            a["pp"][idx(a["pp"])] = int(rng.choice([-1, n, 2**31 - 1]))
        elif m == 2 and len(a["pd"]) > 1:
            a["pd"][0] = a["pd"][1]                                    # a particle in two pairs
        elif m == 3 and len(a["pd"]):
            i = idx(a["pd"]); a["pp"][i] = a["pd"][i]                  # a pair with itself
        elif m == 4:
            a["group"][idx(a["group"])] = int(rng.choice([-1, d.num_groups, 255, 2**31 - 1]))
        elif m == 5:
            a["resid"][idx(a["resid"])] = int(rng.choice([-1, d.num_residues, 2**31 - 1]))
        elif m == 6:
            rng.shuffle(a["resid"])                                    # residues in many runs
        elif m == 7:
            a["mass"][idx(a["mass"])] = float(rng.choice([0.0, -1.0, np.nan, np.inf, 1e-300, 1e300]))
        elif m == 8:
            a["mass"][:] = 0.0
        elif m == 9:
            d.num_groups = int(rng.choice([0, -1, 33, 2046, 2047, 100000]))
        elif m == 10:
            d.num_residues = int(rng.choice([0, -1, 1, n + 5]))
        elif m == 11:
            d.num_nh_chains = int(rng.choice([0, -3, 5, 16, 17, 64, 100000]))
        elif m == 12:
            d.drude_steps_per_real_step = int(rng.choice([0, -1, 1, 10**6]))
        elif m == 13:
            d.step_size = float(rng.choice([0.0, -0.001, np.nan, np.inf]))
        elif m == 14:
            d.max_drude_distance = float(rng.choice([-0.01, np.nan, np.inf, 1e-30]))
        elif m == 15:
            d.padded_num_particles = int(rng.choice([0, n - 1, -32, 715_827_883, 2**31 - 1]))
        elif m == 16 and len(a["ci"]):
            a["ci"][idx(a["ci"])] = int(rng.choice([-1, n, 2**31 - 1]))
        elif m == 17:
            d.mode = int(rng.choice([-1, 2, 77]))
        elif m == 18:
            d.precision = int(rng.choice([-1, 3, 99]))
        elif m == 19:
            d.flags = int(rng.integers(0, 2**31 - 1))
        elif m == 20:
            for name in ("temperature", "coupling_time", "drude_temperature", "drude_coupling_time", "kB"):
                if rng.integers(0, 3) == 0:
                    setattr(d, name, float(rng.choice([0.0, -1.0, np.nan, np.inf])))
        elif m == 21:
            d.mode = _lib.MODE_DUALNH                                  # (valid: the other semantic mode over the same arrays)
        elif m == 22:
            a["null"] = a.get("null", ()) + (str(rng.choice(["group", "resid", "mass", "pair_drude", "constraint_i"])),)   # a null array pointer (legal for group / resid in dualNH mode)
        elif m == 23:
            d.mode = _lib.MODE_DUALNH
            a["null"] = a.get("null", ()) + ("group", "resid")
    return done


@pytest.mark.parametrize("seed", range(6))
def test_malformed_descriptors_are_answered_not_crashed_on(seed):
    lib = _lib.load()
    rng0 = np.random.default_rng(90_000 + seed)
    counts = {}
    for case in range(120):
        keep = []
        d, a, rng = base_desc(int(rng0.integers(0, 200)), keep)
        rng = np.random.default_rng(int(rng0.integers(0, 2**31)))
        what = mutate(d, a, rng)
        point(d, a)
        h = C.c_void_p()
        rc = lib.tgnh_create(C.byref(d), C.byref(h))
        counts[rc] = counts.get(rc, 0) + 1
        assert rc in (_lib.TGNH_OK, _lib.ERR_ARG, _lib.ERR_UNSUPPORTED, _lib.ERR_GROUP_MISMATCH), (seed, case, what, rc, lib.tgnh_last_error())
        if rc != _lib.TGNH_OK:
            assert not h.value and lib.tgnh_last_error(), (seed, case, what)
            continue
        # a handle: its queries answer and it goes away cleanly
        nt = C.c_int()
        assert lib.tgnh_get_num_thermostats(h, C.byref(nt)) == _lib.TGNH_OK and nt.value >= 2
        dof, nkt = np.zeros(nt.value), np.zeros(nt.value)
        assert lib.tgnh_get_dof(h, dof.ctypes.data_as(_lib.c_f64p), nkt.ctypes.data_as(_lib.c_f64p)) == _lib.TGNH_OK
        g, why = C.c_int(), C.c_char_p()
        assert lib.tgnh_get_step_path(h, C.byref(g), C.byref(why)) == _lib.TGNH_OK
        for which in range(9):
            ln = C.c_int()
            if lib.tgnh_get_topology_len(h, which, C.byref(ln)) == _lib.TGNH_OK and ln.value > 0:
                out = np.zeros(ln.value, np.int32)
                assert lib.tgnh_get_topology(h, which, out.ctypes.data_as(_lib.c_i32p)) == _lib.TGNH_OK
        assert lib.tgnh_step_begin(h, None) != _lib.TGNH_OK                 # host-only: nothing launches
        assert lib.tgnh_destroy(h) == _lib.TGNH_OK
    print(f"seed {seed}: {counts}")
    assert len(counts) >= 2                                            # (both outcomes occur: the mutations are neither all fatal nor all harmless)


def _odd_args(argtypes, handle, rng):
    """arguments for one call: the handle first where the entry point takes one, then null pointers / odd scalars"""
    out = []
    for k, t in enumerate(argtypes):
        if k == 0 and t is C.c_void_p:
            out.append(handle)
        elif t is C.c_double:
            out.append(float(rng.choice([0.0, -1.0, 1e300, np.nan, np.inf])))
        elif t in (C.c_int, C.c_int32, C.c_int64):
            out.append(int(rng.choice([0, -1, 1, 7, 99, 2**31 - 1, -2**31])))
        elif t is _lib.ALLREDUCE_FN:
            out.append(_lib.ALLREDUCE_FN(0))
        else:
            out.append(None)                                           # every pointer: null
    return out


def test_every_entry_point_answers_null_and_odd_arguments():
    """All 70-odd exported functions, called (a) with a null handle, (b) with a host-only handle (device -1: nothing may launch),
    null pointers and odd scalars everywhere else, several times over: a status comes back (TGNH_OK only where the call has
    nothing to do) and the process lives.  tgnh_destroy is left for last; tgnh_create has its own test above."""
    lib = _lib.load()
    keep = []
    d, a, _ = base_desc(7, keep)
    point(d, a)
    h = C.c_void_p()
    assert lib.tgnh_create(C.byref(d), C.byref(h)) == _lib.TGNH_OK
    rng = np.random.default_rng(4)
    skip = {"tgnh_last_error", "tgnh_abi_version", "tgnh_create", "tgnh_destroy", "tgnh_rccl_unique_id"}
    seen = {}
    for rnd in range(6):
        for name, (res, argtypes) in _lib.SIGNATURES.items():
            if name in skip or not hasattr(lib, name):                # (the sanitizer build holds the host code only: no harness)
                continue
            for handle in (None, h):
                rc = getattr(lib, name)(*_odd_args(argtypes, handle, rng))
                seen.setdefault(name, set()).add(rc)
                assert rc in (_lib.TGNH_OK, _lib.ERR_ARG, _lib.ERR_STATE, _lib.ERR_UNSUPPORTED, _lib.ERR_HIP), (name, handle is not None, rc, lib.tgnh_last_error())
                if handle is None and argtypes and argtypes[0] is C.c_void_p:
                    assert rc != _lib.TGNH_OK, name                   # a null handle is never fine
    assert lib.tgnh_rccl_unique_id(None) != _lib.TGNH_OK
    # the handle is still what it was
    nt = C.c_int()
    assert lib.tgnh_get_num_thermostats(h, C.byref(nt)) == _lib.TGNH_OK and nt.value >= 2
    assert lib.tgnh_destroy(h) == _lib.TGNH_OK
    assert lib.tgnh_destroy(None) != _lib.TGNH_OK
