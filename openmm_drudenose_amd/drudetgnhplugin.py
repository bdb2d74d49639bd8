"""Host-side mirror of the reference's Python surface over the HIP C ABI.

`DrudeTGNHIntegrator` keeps the method names, argument meaning, defaults and error
behaviour of the SWIG class (python/drudetgnhplugin.i:60-92 of scychon/openmm_drudeNose,
C++ side openmmapi/src/DrudeTGNHIntegrator.cpp).  OpenMM is not available here, so
`HipContext` stands where OpenMM's Context + HIP platform would: it owns the device
arrays in OpenMM's layouts (posq real4, velm mixed4 with w = 1/m, force int64 x 2^32 in
three planes, posqCorrection in mixed precision) and drives the C ABI.  torch is used
for device memory, streams and torch.distributed only.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (MODE_DUALNH, MODE_TGNH, PREC_SINGLE, PREC_MIXED, PREC_DOUBLE,  # noqa: F401
                   FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_WAVE_TILES, FLAG_TRUST_STATE_CHANGED, FLAG_GATHER)
from .synth import KB

_PREC = {"single": PREC_SINGLE, "mixed": PREC_MIXED, "double": PREC_DOUBLE}
_MODE = {"dualNH": MODE_DUALNH, "TGNH": MODE_TGNH}


class TgnhError(RuntimeError):
    """Stands for OpenMMException; .status is the C-ABI status code."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def _check(rc):
    if rc != _lib.TGNH_OK:
        raise TgnhError(rc, _lib.load().tgnh_last_error().decode())


class DrudeTGNHIntegrator:
    """Same constructor and methods as the reference's SWIG class (drudetgnhplugin.i:60-92).
    Note the Python default useDrudeNHChains=True (the C++ default is false,
    DrudeTGNHIntegrator.h:71)."""

    def __init__(self, temperature, couplingTime, drudeTemperature, drudeCouplingTime, stepSize,
                 drudeStepsPerRealStep=20, numNHChains=1, useDrudeNHChains=True, useCOMTempGroup=True):
        self._temperature = float(temperature)
        self._couplingTime = float(couplingTime)
        self._drudeTemperature = float(drudeTemperature)
        self._drudeCouplingTime = float(drudeCouplingTime)
        self._maxDrudeDistance = 0.0
        self._stepSize = float(stepSize)
        self._drudeSteps = int(drudeStepsPerRealStep)
        self._numNHChains = int(numNHChains)
        self._useDrudeNHChains = bool(useDrudeNHChains)
        self._useCOMTempGroup = bool(useCOMTempGroup)
        self._constraintTolerance = 1e-5                     # DrudeTGNHIntegrator.cpp:58
        self._tempGroups = []
        self._particleTempGroup = []
        self._context = None

    # --- scalar properties (DrudeTGNHIntegrator.h:77-211) ---
    def getTemperature(self): return self._temperature
    def setTemperature(self, temp): self._temperature = float(temp)
    def getCouplingTime(self): return self._couplingTime
    def setCouplingTime(self, tau): self._couplingTime = float(tau)
    def getDrudeTemperature(self): return self._drudeTemperature
    def setDrudeTemperature(self, temp): self._drudeTemperature = float(temp)
    def getDrudeCouplingTime(self): return self._drudeCouplingTime
    def setDrudeCouplingTime(self, tau): self._drudeCouplingTime = float(tau)
    def getMaxDrudeDistance(self): return self._maxDrudeDistance

    def setMaxDrudeDistance(self, distance):
        if distance < 0:                                     # DrudeTGNHIntegrator.cpp:97-100
            raise TgnhError(_lib.ERR_ARG, "setMaxDrudeDistance: Distance cannot be negative")
        self._maxDrudeDistance = float(distance)
        if self._context is not None:
            self._context._push_scalars()

    def getStepSize(self): return self._stepSize

    def setStepSize(self, dt):
        self._stepSize = float(dt)
        if self._context is not None:
            self._context._push_scalars()

    def getConstraintTolerance(self): return self._constraintTolerance
    def setConstraintTolerance(self, tol): self._constraintTolerance = float(tol)
    def getDrudeStepsPerRealStep(self): return self._drudeSteps

    def setDrudeStepsPerRealStep(self, drudeSteps):
        self._drudeSteps = int(drudeSteps)
        if self._context is not None:
            self._context._push_scalars()

    def getNumNHChains(self): return self._numNHChains
    def setNumNHChains(self, numChains): self._numNHChains = int(numChains)
    def getUseDrudeNHChains(self): return int(self._useDrudeNHChains)
    def setUseDrudeNHChains(self, use): self._useDrudeNHChains = bool(use)
    def getUseCOMTempGroup(self): return int(self._useCOMTempGroup)
    def setUseCOMTempGroup(self, use): self._useCOMTempGroup = bool(use)

    # --- temperature groups (DrudeTGNHIntegrator.cpp:61-86) ---
    def getNumTempGroups(self): return len(self._tempGroups)

    def addTempGroup(self):
        self._tempGroups.append(len(self._tempGroups))
        return len(self._tempGroups) - 1

    def addParticleTempGroup(self, tempGroup):
        if not 0 <= tempGroup < len(self._tempGroups):       # ASSERT_VALID_INDEX
            raise TgnhError(_lib.ERR_ARG, "Index out of range")
        if not isinstance(self._particleTempGroup, list):    # (the defaults of _resolve_groups are an array)
            self._particleTempGroup = [int(x) for x in self._particleTempGroup]
        self._particleTempGroup.append(int(tempGroup))
        return len(self._particleTempGroup) - 1

    def setParticleTempGroup(self, particle, tempGroup):
        if not 0 <= particle < len(self._particleTempGroup) or not 0 <= tempGroup < len(self._tempGroups):
            raise TgnhError(_lib.ERR_ARG, "Index out of range")
        self._particleTempGroup[particle] = int(tempGroup)

    def getParticleTempGroup(self, particle):
        if not 0 <= particle < len(self._particleTempGroup):
            raise TgnhError(_lib.ERR_ARG, "Index out of range")
        return int(self._particleTempGroup[particle])

    # --- stepping (DrudeTGNHIntegrator.cpp:182-194) ---
    def step(self, steps):
        if self._context is None:
            raise TgnhError(_lib.ERR_STATE, "This Integrator is not bound to a context!")
        self._context.step(steps)

    def computeKineticEnergy(self):
        return self._context.kinetic_energy()

    def _resolve_groups(self, num_particles):
        """DrudeTGNHIntegrator.cpp:126-134: default everything to one group; else the count must match."""
        if len(self._particleTempGroup) == 0:
            if len(self._tempGroups) == 0:
                self._tempGroups.append(0)
            # (an array: a Python list of millions of ints is walked by every full pass of the garbage collector)
            self._particleTempGroup = np.zeros(num_particles, np.int32)
        elif len(self._particleTempGroup) != num_particles:
            raise TgnhError(_lib.ERR_ARG, "Number of particles assigned with temperature groups does not match the number of system particles")
        return np.asarray(self._particleTempGroup, np.int32), len(self._tempGroups)


def create_handle(lib, system, integrator, group, ngroups, mode, precision, device, flags, kB, padded):
    """tgnh_create from a DrudeSystem + integrator mirror.  device = -1 gives a host-only handle."""
    n = system.num_particles
    d = _lib.TgnhDesc()
    d.struct_size = C.sizeof(_lib.TgnhDesc)
    d.mode, d.precision, d.flags, d.device = mode, precision, int(flags), device
    d.num_particles, d.padded_num_particles = n, padded
    d.num_pairs, d.num_groups, d.num_residues = system.num_pairs, ngroups, system.num_residues
    d.num_constraints = len(system.constraints)
    d.has_cm_motion_remover = int(system.has_cm_motion_remover)
    group = np.ascontiguousarray(group, np.int32)
    ci = np.ascontiguousarray(system.constraints[:, 0]) if len(system.constraints) else None
    cj = np.ascontiguousarray(system.constraints[:, 1]) if len(system.constraints) else None
    d.mass = system.mass.ctypes.data_as(_lib.c_f64p)
    d.pair_drude = system.pair_drude.ctypes.data_as(_lib.c_i32p)
    d.pair_parent = system.pair_parent.ctypes.data_as(_lib.c_i32p)
    d.group = group.ctypes.data_as(_lib.c_i32p)
    d.resid = system.resid.ctypes.data_as(_lib.c_i32p)
    if ci is not None:
        d.constraint_i = ci.ctypes.data_as(_lib.c_i32p)
        d.constraint_j = cj.ctypes.data_as(_lib.c_i32p)
    d.kB = kB
    d.temperature, d.coupling_time = integrator.getTemperature(), integrator.getCouplingTime()
    d.drude_temperature, d.drude_coupling_time = integrator.getDrudeTemperature(), integrator.getDrudeCouplingTime()
    d.step_size = integrator.getStepSize()
    d.drude_steps_per_real_step = integrator.getDrudeStepsPerRealStep()
    d.num_nh_chains = integrator.getNumNHChains()
    d.use_drude_nh_chains = integrator.getUseDrudeNHChains()
    d.use_com_temp_group = integrator.getUseCOMTempGroup()
    d.max_drude_distance = integrator.getMaxDrudeDistance()
    h = C.c_void_p()
    _check(lib.tgnh_create(C.byref(d), C.byref(h)))
    return h


class _HandleQueries:
    """Queries shared by device contexts and host-only handles."""

    def _stream(self):
        return None

    def step_path(self):
        """('tiled', '') or ('gather', why): the reference's own un-fused kernels by global index, for what the tiles cannot hold"""
        g, why = C.c_int(), C.c_char_p()
        _check(self.lib.tgnh_get_step_path(self.h, C.byref(g), C.byref(why)))
        return ("tiled", "gather", "gather")[g.value], (why.value or b"").decode()

    def local_dof_terms(self):
        n = C.c_int()
        _check(self.lib.tgnh_get_local_dof_terms(self.h, None, C.byref(n)))
        out = np.zeros(n.value)
        _check(self.lib.tgnh_get_local_dof_terms(self.h, out.ctypes.data_as(_lib.c_f64p), C.byref(n)))
        return out

    def set_global_dof_terms(self, total):
        total = np.ascontiguousarray(total, np.float64)
        _check(self.lib.tgnh_set_global_dof_terms(self.h, total.ctypes.data_as(_lib.c_f64p), len(total)))

    def num_thermostats(self):
        n = C.c_int()
        _check(self.lib.tgnh_get_num_thermostats(self.h, C.byref(n)))
        return n.value

    def dof(self):
        n = self.num_thermostats()
        dof, nkt = np.zeros(n), np.zeros(n)
        _check(self.lib.tgnh_get_dof(self.h, dof.ctypes.data_as(_lib.c_f64p), nkt.ctypes.data_as(_lib.c_f64p)))
        return dof, nkt

    def thermostat_state(self, which):
        n = C.c_int()
        _check(self.lib.tgnh_get_thermostat_len(self.h, which, C.byref(n)))
        out = np.zeros(n.value)
        _check(self.lib.tgnh_get_thermostat_state(self.h, which, self._stream(), out.ctypes.data_as(_lib.c_f64p)))
        return out

    def set_thermostat_state(self, which, arr):
        arr = np.ascontiguousarray(arr, np.float64)
        _check(self.lib.tgnh_set_thermostat_state(self.h, which, self._stream(), arr.ctypes.data_as(_lib.c_f64p)))

    def topology(self, which):
        n = C.c_int()
        _check(self.lib.tgnh_get_topology_len(self.h, which, C.byref(n)))
        out = np.zeros(n.value, np.int32)
        _check(self.lib.tgnh_get_topology(self.h, which, out.ctypes.data_as(_lib.c_i32p)))
        return out


class HostTopology(_HandleQueries):
    """Host-only handle (device -1): the library's topology / tile / dof logic without a GPU.  Used by the CPU
    tests; it cannot launch anything."""

    def __init__(self, system, integrator, mode="TGNH", precision="mixed", flags=0, kB=KB):
        self.lib = _lib.load()
        group, ngroups = integrator._resolve_groups(system.num_particles)
        self.group, self.num_groups = group, ngroups
        n = system.num_particles
        self.h = create_handle(self.lib, system, integrator, group, ngroups, _MODE[mode], _PREC[precision], -1, flags, kB,
                               (n + 31) // 32 * 32)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tgnh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipContext(_HandleQueries):
    """Device state + handle: what OpenMM's Context / HIP platform data are to the reference kernel.

    force_fn(ctx) is the force call-out (calcForcesAndEnergy); the default is the
    harness force (Drude spring + tether) computed by the library's own kernel.
    """

    def __init__(self, system, integrator, mode="TGNH", precision="mixed", device=0, flags=0, kB=KB,
                 k_drude=None, k_tether=None, allreduce=None, global_dof_sum=None, lattice_sites=True):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("the DrudeTGNH HIP path needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.system, self.integrator = system, integrator
        self.mode, self.precision = _MODE[mode], _PREC[precision]
        self.dev = torch.device("cuda", device)
        n = system.num_particles
        self.n = n
        self.padded = (n + 31) // 32 * 32                    # OpenMM PADDED_NUM_ATOMS (TileSize 32)
        group, ngroups = integrator._resolve_groups(n)
        self.group, self.num_groups = group, ngroups
        from .synth import K_DRUDE, K_TETHER
        self.k_drude = K_DRUDE if k_drude is None else k_drude
        self.k_tether = K_TETHER if k_tether is None else k_tether

        with torch.cuda.device(self.dev):
            h = create_handle(self.lib, system, integrator, group, ngroups, self.mode, self.precision, device, flags, kB,
                              self.padded)
        self.h = h
        self._hook = None
        if global_dof_sum is not None:                       # particle sharding: dof terms are additive over ranks
            self.set_global_dof_terms(global_dof_sum(self.local_dof_terms()))
        if allreduce is not None:
            self.set_allreduce(allreduce)

        rdt = torch.float32 if self.precision != PREC_DOUBLE else torch.float64
        mdt = torch.float32 if self.precision == PREC_SINGLE else torch.float64
        self.rdt, self.mdt = rdt, mdt
        self.posq = torch.zeros((n, 4), dtype=rdt, device=self.dev)
        self.posq_corr = torch.zeros((n, 4), dtype=torch.float32, device=self.dev) if self.precision == PREC_MIXED else None
        self.velm = torch.zeros((n, 4), dtype=mdt, device=self.dev)
        self.force = torch.zeros(3 * self.padded, dtype=torch.int64, device=self.dev)
        self.pos_delta = torch.zeros((n, 4), dtype=mdt, device=self.dev)
        self.x0 = torch.zeros((n, 4), dtype=rdt, device=self.dev)        # harness tether sites, w = tether flag
        self.sites_packed = False
        self.unpacked_sites = False                                      # (tests: the harness force from x0 itself, the round-1 kernel)
        self.use_lattice_sites = bool(lattice_sites)                     # (tests: False keeps the packed sites where the lattice form would hold)
        inv = np.where(system.mass == 0.0, 0.0, 1.0 / np.where(system.mass == 0.0, 1.0, system.mass))
        self.velm[:, 3] = torch.from_numpy(inv).to(self.dev, mdt)
        _check(self.lib.tgnh_bind_buffers(self.h, self.posq.data_ptr(),
                                          self.posq_corr.data_ptr() if self.posq_corr is not None else None,
                                          self.velm.data_ptr(), self.force.data_ptr(), self.pos_delta.data_ptr()))
        integrator._context = self
        self.force_fn = None
        self.state_hook = None
        self.ke_sum_valid = False
        self.constrained = False
        if system.cluster_atoms is not None and len(system.cluster_atoms):
            _check(self.lib.tgnh_harness_set_clusters(self.h, len(system.cluster_atoms),
                                                      system.cluster_atoms.ctypes.data_as(_lib.c_i32p),
                                                      system.cluster_dist.ctypes.data_as(_lib.c_f64p)))
            self.constrained = True
        if system.site_atoms is not None and len(system.site_atoms):
            _check(self.lib.tgnh_harness_set_virtual_sites(self.h, len(system.site_atoms),
                                                           system.site_atoms.ctypes.data_as(_lib.c_i32p),
                                                           system.site_weights.ctypes.data_as(_lib.c_f64p)))
            self.constrained = True        # the split path is the one with the virtual-site call-out
        if system.positions is not None:
            self.setPositions(system.positions)
            self.set_sites(system.positions)
        if system.velocities is not None:
            self.setVelocities(system.velocities)
        if system.positions is not None:
            self.compute_forces()

    # ---- plumbing ----
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def _push_scalars(self):
        i = self.integrator
        _check(self.lib.tgnh_set_step_size(self.h, i.getStepSize()))
        _check(self.lib.tgnh_set_drude_steps_per_real_step(self.h, i.getDrudeStepsPerRealStep()))
        _check(self.lib.tgnh_set_max_drude_distance(self.h, i.getMaxDrudeDistance()))

    def close(self):
        if getattr(self, "h", None):
            self.torch.cuda.synchronize(self.dev)
            self.lib.tgnh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_allreduce(self, fn):
        """fn(tensor_view) must sum the float64 device tensor in place over all ranks (stream-ordered)."""
        torch = self.torch

        def _hook(buf, count, stream, user):
            try:
                t = self._ke_view(buf, count)
                fn(t)
                return 0
            except Exception as e:     # never unwind through C
                print("all-reduce hook failed:", e)
                return 1
        self._hook = _lib.ALLREDUCE_FN(_hook)
        self._views = {}
        _check(self.lib.tgnh_set_allreduce(self.h, self._hook, None))

    # ---- the library's own RCCL all-reduce (include/drude_tgnh.h: tgnh_rccl_*) ----
    RCCL_ID_BYTES = 128

    def rccl_init(self, world, rank, unique_id):
        """Collective: every rank of `world` calls this with rank 0's id (rccl_unique_id())."""
        _check(self.lib.tgnh_rccl_init(self.h, int(world), int(rank), C.c_char_p(unique_id)))
        self._hook = None

    def rccl_unique_id(self):
        buf = C.create_string_buffer(self.RCCL_ID_BYTES)
        _check(self.lib.tgnh_rccl_unique_id(buf))
        return buf.raw

    def rccl_shutdown(self):
        _check(self.lib.tgnh_rccl_shutdown(self.h))

    def rccl_init_over(self, dist, rank, world):
        """The library's communicator set up through a torch.distributed process group (which only carries the 128-byte id):
        all ranks return True (the library enqueues ncclAllReduce itself from now on) or all return False (nothing changed)."""
        ok, err = 1, None
        ids = [None]
        # ncclCommInitRank is collective: a rank that cannot even bind RCCL must say so BEFORE anybody enters it, or its peers
        # wait inside it for good.  Every rank therefore makes an id of its own first (binds the library, not collective) and
        # the ranks vote; only rank 0's id is used.
        try:
            mine = self.rccl_unique_id()
            if rank == 0:
                ids[0] = mine
        except Exception as e:                      # noqa: BLE001
            ok, err = 0, e
        votes = [None] * world
        dist.all_gather_object(votes, ok)
        if not all(votes):
            self.rccl_error = err or RuntimeError("RCCL could not be bound on rank(s) " + str([r for r, v in enumerate(votes) if not v]))
            return False
        dist.broadcast_object_list(ids, src=0)
        try:
            self.rccl_init(world, rank, ids[0])
        except Exception as e:                      # noqa: BLE001
            ok, err = 0, e
        flags = [None] * world
        dist.all_gather_object(flags, ok)
        if not all(flags):
            if ok:
                self.rccl_shutdown()
            self.rccl_error = err
            return False
        return True

    # ---- mailbox exchange over xGMI (include/drude_tgnh.h: tgnh_exchange_*) ----
    XCHG_HANDLE_BYTES = 64

    def exchange_wait_stats(self):
        """(mean us, max us, exchanges) of this rank's waits for its peers' sums since the last call; resets."""
        mean, mx, n = C.c_double(), C.c_double(), C.c_int64()
        _check(self.lib.tgnh_exchange_wait_stats(self.h, self._stream(), C.byref(mean), C.byref(mx), C.byref(n)))
        return mean.value, mx.value, n.value

    def exchange_create(self, world, rank):
        """-> (IPC handle bytes for peers in other processes, device pointer for peers in this process)"""
        buf = C.create_string_buffer(self.XCHG_HANDLE_BYTES)
        ptr = C.c_void_p()
        _check(self.lib.tgnh_exchange_create(self.h, int(world), int(rank), buf, C.byref(ptr)))
        return buf.raw, ptr.value

    def exchange_attach(self, handles):
        """handles: every rank's IPC handle bytes, in rank order (own entry ignored)."""
        blob = b"".join(handles)
        _check(self.lib.tgnh_exchange_attach(self.h, blob))

    def exchange_attach_pointers(self, pointers):
        arr = (C.c_void_p * len(pointers))(*[C.c_void_p(p) for p in pointers])
        _check(self.lib.tgnh_exchange_attach_pointers(self.h, arr))

    def set_resident_share(self, share):
        """FLAG_RESIDENT_STEP: this context may fill 1/share of the device (several contexts stepping concurrently)."""
        _check(self.lib.tgnh_set_resident_share(self.h, int(share)))

    def resident_work_groups(self):
        """Work-groups of step_kernel per compute unit found resident together at create (0: the deferred launches run)."""
        n = C.c_int()
        _check(self.lib.tgnh_get_resident_work_groups(self.h, C.byref(n)))
        return n.value

    def resident_kernel(self):
        """The kernel the next step_begin runs as its one launch: None, 'step_kernel' or 'wstep_kernel'."""
        n = C.c_int()
        _check(self.lib.tgnh_get_resident_kernel(self.h, C.byref(n)))
        return (None, "step_kernel", "wstep_kernel")[n.value]

    def exchange_detach(self):
        _check(self.lib.tgnh_exchange_detach(self.h))

    def attach_exchange_over(self, dist, rank, world, tensor_device="cuda"):
        """Collective over a torch.distributed process group: every rank creates its mailbox, the hipIpc handles go
        round with all_gather_object, every rank maps its peers.  All ranks return the same answer: True = the
        mailbox exchange is attached everywhere; False = nowhere (a rank could not create / map: the all-reduce hook,
        if set, keeps doing the exchange)."""
        torch = self.torch
        ok, err = 1, None
        try:
            handle, _ = self.exchange_create(world, rank)
            handles = [None] * world
            dist.all_gather_object(handles, handle)
            self.exchange_attach(handles)
        except Exception as e:                      # noqa: BLE001 -- any failure means "not here", decided collectively below
            ok, err = 0, e
        t = torch.tensor([ok], dtype=torch.int32, device=tensor_device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            if ok:
                self.exchange_detach()
            self.exchange_error = err
            return False
        return True

    def _ke_view(self, ptr, count):
        key = (ptr, count)
        v = self._views.get(key)
        if v is None:
            torch = self.torch

            class _Holder:
                pass
            hold = _Holder()
            hold.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}
            v = torch.as_tensor(hold, device=self.dev)
            self._views[key] = v
        return v

    # ---- state (Context::setPositions / setVelocities / getState) ----
    def setPositions(self, pos):
        torch = self.torch
        self._state_changed()
        p = torch.from_numpy(np.ascontiguousarray(pos, np.float64)).to(self.dev)
        if self.precision == PREC_DOUBLE:
            self.posq[:, :3] = p
        else:
            hi = p.to(torch.float32)
            self.posq[:, :3] = hi
            if self.posq_corr is not None:
                self.posq_corr[:, :3] = (p - hi.to(torch.float64)).to(torch.float32)

    def set_sites(self, x0):
        """Harness tether sites (stored in the position type, so float-rounded unless precision is double);
        every massive particle that is not a Drude particle is tethered."""
        self.x0[:, :3] = self.torch.from_numpy(np.ascontiguousarray(x0, np.float64)).to(self.dev, self.rdt)
        flag = (self.system.mass > 0)
        flag[self.system.pair_drude] = False
        self.x0[:, 3] = self.torch.from_numpy(flag.astype(np.float64)).to(self.dev, self.rdt)
        self.torch.cuda.synchronize(self.dev)
        lat = getattr(self.system, "lattice", None)
        if lat is not None and self.use_lattice_sites:
            # (a hint only: the library checks every slot against it and packs the sites as before where it does not hold)
            k, side, spacing, geom, first = lat
            geom = np.ascontiguousarray(geom, np.float64)
            _check(self.lib.tgnh_harness_lattice_hint(self.h, int(k), int(side), float(spacing), geom.ctypes.data_as(_lib.c_f64p), int(first)))
        else:
            _check(self.lib.tgnh_harness_lattice_hint(self.h, 0, 0, 0.0, None, 0))
        _check(self.lib.tgnh_harness_pack_sites(self.h, self.x0.data_ptr()))
        self.sites_packed = True

    def sites_kind(self):
        """how the harness force finds its tether sites: 'x0' (explicit array), 'packed', 'lattice'"""
        n = C.c_int()
        _check(self.lib.tgnh_harness_sites_kind(self.h, C.byref(n)))
        return ("x0", "packed", "lattice")[n.value]

    def _x0_arg(self):
        """None (NULL: the library reads its packed copy) once set_sites has packed them, else the float4 / double4 array"""
        return None if self.sites_packed and not self.unpacked_sites else self.x0.data_ptr()

    def sites(self):
        """The tether sites exactly as the harness kernel sees them (for the oracle)."""
        return self.x0[:, :3].to(self.torch.float64).cpu().numpy()

    def setVelocities(self, vel):
        self._state_changed()                                # (first: refused between the steps of a deferred sequence, and then nothing is written)
        self.velm[:, :3] = self.torch.from_numpy(np.ascontiguousarray(vel, np.float64)).to(self.dev, self.mdt)

    def _state_changed(self):                                # DrudeTGNHIntegrator.cpp:166-170
        self.ke_sum_valid = False
        _check(self.lib.tgnh_state_changed(self.h))

    def getPositions(self):
        p = self.posq[:, :3].to(self.torch.float64)
        if self.posq_corr is not None:
            p = p + self.posq_corr[:, :3].to(self.torch.float64)
        return p.cpu().numpy()

    def flush(self):
        """velm <- the reference's end-of-step velocities (a deferred variant's pending half kick and factors applied)"""
        _check(self.lib.tgnh_flush(self.h, self._stream()))

    def getVelocities(self):
        _check(self.lib.tgnh_flush(self.h, self._stream()))
        return self.velm[:, :3].to(self.torch.float64).cpu().numpy()

    def getForces(self):
        f = self.force.view(3, self.padded)[:, :self.n].to(self.torch.float64) / 4294967296.0
        return f.t().contiguous().cpu().numpy()

    def setForces(self, f):
        """Forces in kJ/mol/nm -> OpenMM fixed point (long long)(f * 2^32)."""
        t = self.torch.from_numpy(np.ascontiguousarray(f, np.float64)).to(self.dev)
        self.force.view(3, self.padded)[:, :self.n] = (t * 4294967296.0).to(self.torch.int64).t()

    # ---- the step ----
    def compute_forces(self):
        if self.force_fn is not None:
            self.force_fn(self)
        else:
            _check(self.lib.tgnh_harness_force(self.h, self._x0_arg(), self.k_drude, self.k_tether,
                                               self.force.data_ptr(), self._stream()))

    def step_begin(self):
        _check(self.lib.tgnh_step_begin(self.h, self._stream()))

    def step_end(self):
        _check(self.lib.tgnh_step_end(self.h, self._stream()))
        self.ke_sum_valid = True                             # DrudeTGNHIntegrator.cpp:192

    def step(self, steps):
        if self.force_fn is None and self.constrained:
            _check(self.lib.tgnh_run_harness_constrained(self.h, self._x0_arg(), self.k_drude, self.k_tether,
                                                         self.integrator.getConstraintTolerance(), int(steps), self._stream()))
            if steps > 0:
                self.ke_sum_valid = True
            return
        if self.force_fn is None:
            _check(self.lib.tgnh_run_harness(self.h, self._x0_arg(), self.k_drude, self.k_tether, int(steps), self._stream()))
            if steps > 0:
                self.ke_sum_valid = True
            return
        # caller-supplied force call-out (and, optionally, a state hook: Context::updateContextState, e.g. CMMotionRemover)
        lib, h = self.lib, self.h
        for _ in range(steps):
            if self.state_hook is not None:
                self.state_hook(self)                                   # DrudeTGNHIntegrator.cpp:186
            if not self.constrained:
                self.step_begin()
                self.compute_forces()
                self.step_end()
                continue
            tol = self.integrator.getConstraintTolerance()
            _check(lib.tgnh_step_begin_kick(h, self._stream()))          # Cu :336-360
            _check(lib.tgnh_harness_shake_positions(h, tol, self._stream()))   # Cu :363
            _check(lib.tgnh_step_begin_move(h, self._stream()))          # Cu :366-376
            _check(lib.tgnh_harness_virtual_sites(h, self._stream()))    # Cu :377
            self.compute_forces()                                        # Cu :380
            _check(lib.tgnh_step_end_kick(h, self._stream()))            # Cu :384-388
            if self.mode == MODE_TGNH:
                _check(lib.tgnh_harness_shake_velocities(h, tol, self._stream()))   # Cu :391
            _check(lib.tgnh_step_end_thermo(h, self._stream()))          # Cu :394-406
            self.ke_sum_valid = True

    def step_without_forces(self, steps):
        """`steps` x (step_begin, step_end) with NO force call-out, on whatever the force buffer holds: for timing the integrator's
        own launches (bench.py's integrator_only leg); not a trajectory of anything."""
        if self.constrained:
            raise TgnhError(_lib.ERR_STATE, "step_without_forces: unconstrained systems only")
        _check(self.lib.tgnh_run_steps(self.h, int(steps), self._stream()))
        if steps > 0:
            self.ke_sum_valid = True

    def capture_steps(self, steps, forces=True):
        """Captures `steps` time steps (harness force call-out included -- and the harness' constraint / virtual-site call-outs for a
        system with constraints --, and the KE all-reduce when sharded; forces=False: no call-out, see step_without_forces) into
        a hipGraph on a side stream and returns a callable that replays it.  The step sequence must not change afterwards
        (no setters).  A handle that is not in the steady state of its step sequence is brought there first, by up to
        three times `steps` real steps (see below): read the step count afterwards if it matters."""
        torch = self.torch
        if self.force_fn is not None:
            raise TgnhError(_lib.ERR_STATE, "capture_steps supports the harness force call-out only")
        # A graph replays launches, not the decisions that chose them: it may only be replayed from the state it was recorded
        # in.  The first steps of a handle differ from the later ones (nothing pending yet, a staged thermostat block to
        # commit), and so does a step after a query or a split step that settled part of what a step leaves owed.  So the
        # recording is checked: what the handle owes the trajectory (tgnh_get_pending_state) must be the same after the
        # recorded steps as before.  If it is not, the recording is a transition into the steady state: it is replayed once --
        # which takes exactly those steps, correctly, from the state they were recorded in -- and dropped, and the steps are
        # recorded again from where that leaves the handle.
        torch.cuda.synchronize(self.dev)
        # Consecutive streaming launches sweep the tiles in alternating directions, and a time step holds an odd number
        # of them: the launches of `steps` steps captured now and the launches of the `steps` steps after them differ in
        # that direction when `steps` is odd.  Two graphs are therefore captured back to back and replayed in turn, so a
        # replayed run issues exactly the launches the eager loop would (bitwise the same trajectory, for any `steps`).
        def record():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if self.constrained and not forces:
                    raise TgnhError(_lib.ERR_STATE, "capture_steps(forces=False): unconstrained systems only")
                if self.constrained:                         # the step loop step() runs for such a system: the split entry points around the harness' constraint call-outs
                    _check(self.lib.tgnh_run_harness_constrained(self.h, self._x0_arg(), self.k_drude, self.k_tether,
                                                                 self.integrator.getConstraintTolerance(), int(steps), self._stream()))
                elif not forces:
                    _check(self.lib.tgnh_run_steps(self.h, int(steps), self._stream()))
                else:
                    _check(self.lib.tgnh_run_harness(self.h, self._x0_arg(), self.k_drude, self.k_tether, int(steps), self._stream()))
            return g
        for _ in range(4):
            owed_before = self.pending_state() & 0x2ff
            first = record()
            if self.pending_state() & 0x2ff == owed_before:
                break
            first.replay()                                   # a transition: taken for real (the recording has advanced the clock)
            torch.cuda.synchronize(self.dev)
            self.ke_sum_valid = True
        else:
            raise TgnhError(_lib.ERR_STATE, "capture_steps: the handle does not reach a steady state of its step sequence")
        clock = self.time()
        graphs = [first, record()]
        if self.pending_state() & 0x2ff != owed_before:
            raise TgnhError(_lib.ERR_STATE, "capture_steps: the step sequence is not periodic")
        # a capture advances the host-side clock without running anything: the first one stands for the first replay,
        # the second one is taken back
        self.set_time(*clock)
        self._captured_not_run = int(steps)
        turn = [0]

        def replay():
            graphs[turn[0] & 1].replay()
            turn[0] += 1
            if self._captured_not_run:
                self._captured_not_run = 0          # first replay is the run the capture already counted
            else:
                _check(self.lib.tgnh_note_replayed_steps(self.h, int(steps)))
            self.ke_sum_valid = True
        replay.graph = graphs[0]
        replay.graphs = graphs
        return replay

    def pending_state(self):
        """tgnh_get_pending_state: what the handle still owes the trajectory (bit set, include/drude_tgnh.h)."""
        bits = C.c_uint32()
        _check(self.lib.tgnh_get_pending_state(self.h, C.byref(bits)))
        return bits.value

    def status_flags(self):
        """The device's status word (bit 0 Drude beyond 2x the hard wall, bit 1 harness SHAKE not converged, bit 2 mailbox
        exchange timed out), whether or not it spells a failure.  Synchronises the stream."""
        flags = C.c_uint32()
        self.lib.tgnh_get_status_flags(self.h, self._stream(), C.byref(flags))
        return flags.value

    def check(self):
        """The status word; raises TgnhError for the failures among its bits: a Drude beyond 2x the hard wall in dualNH
        mode (Ref :311-312), a mailbox exchange that timed out.  Once seen, a failure is sticky in the library: every
        later step or query raises too."""
        flags = C.c_uint32()
        _check(self.lib.tgnh_get_status_flags(self.h, self._stream(), C.byref(flags)))
        return flags.value

    # ---- queries ----
    def _vec(self, fn, n):
        out = np.zeros(n)
        _check(fn(self.h, self._stream(), out.ctypes.data_as(_lib.c_f64p)))
        return out

    def last_kinetic_energies(self):
        return self._vec(self.lib.tgnh_get_last_kinetic_energies, self.num_thermostats())

    def last_scale_factors(self):
        return self._vec(self.lib.tgnh_get_last_scale_factors, self.num_thermostats())

    def compute_kinetic_energies(self):
        _check(self.lib.tgnh_compute_kinetic_energies(self.h, self._stream()))
        return self.last_kinetic_energies()

    def half_kick(self):
        _check(self.lib.tgnh_half_kick(self.h, self._stream()))

    def kinetic_energy(self):
        out = C.c_double()
        _check(self.lib.tgnh_get_kinetic_energy(self.h, int(self.ke_sum_valid), self._stream(), C.byref(out)))
        return out.value

    def time(self):
        t, k = C.c_double(), C.c_int64()
        _check(self.lib.tgnh_get_time(self.h, C.byref(t), C.byref(k)))
        return t.value, k.value

    def set_time(self, time, step_count):
        """Context::setTime / setStepCount of a restored checkpoint."""
        _check(self.lib.tgnh_set_time(self.h, float(time), int(step_count)))

    def timing(self, on):
        _check(self.lib.tgnh_timing_enable(self.h, int(on)))

    def timing_read(self, kid):
        ms, n = C.c_double(), C.c_int64()
        _check(self.lib.tgnh_timing_read(self.h, kid, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def algorithmic_bytes(self, kid):
        b = C.c_double()
        _check(self.lib.tgnh_algorithmic_bytes(self.h, kid, C.byref(b)))
        return b.value
