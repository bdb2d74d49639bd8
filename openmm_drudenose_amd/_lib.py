"""ctypes binding of include/drude_tgnh.h (the drop-in C ABI)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TGNH_LIB") or os.path.join(HERE, "libdrudetgnh_hip.so")   # TGNH_LIB: tuning builds (tools/build_variant.py)

TGNH_OK = 0
ERR_ARG, ERR_GROUP_MISMATCH, ERR_HARDWALL, ERR_UNSUPPORTED, ERR_HIP, ERR_STATE = -1, -2, -3, -4, -5, -6
MODE_DUALNH, MODE_TGNH = 0, 1
PREC_SINGLE, PREC_MIXED, PREC_DOUBLE = 0, 1, 2
FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_WAVE_TILES, FLAG_TRUST_STATE_CHANGED, FLAG_GATHER = 2, 4, 8, 16, 32
KID_SKD, KID_KICK_KE, KID_SCALE, KID_KE, KID_CHAIN, KID_FORCE, KID_OTHER, KID_STEP = range(8)
KERNEL_NAMES = {KID_SKD: "scale+kick+drift", KID_KICK_KE: "kick+KE", KID_SCALE: "rescale", KID_KE: "KE",
                KID_CHAIN: "chain", KID_FORCE: "harness force", KID_OTHER: "other", KID_STEP: "resident step"}

c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)


class TgnhDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("mode", C.c_int32), ("precision", C.c_int32), ("flags", C.c_int32),
        ("device", C.c_int32), ("num_particles", C.c_int32), ("padded_num_particles", C.c_int32),
        ("num_pairs", C.c_int32), ("num_groups", C.c_int32), ("num_residues", C.c_int32),
        ("num_constraints", C.c_int32), ("has_cm_motion_remover", C.c_int32),
        ("mass", c_f64p), ("pair_drude", c_i32p), ("pair_parent", c_i32p), ("group", c_i32p), ("resid", c_i32p),
        ("constraint_i", c_i32p), ("constraint_j", c_i32p),
        ("kB", C.c_double), ("temperature", C.c_double), ("coupling_time", C.c_double),
        ("drude_temperature", C.c_double), ("drude_coupling_time", C.c_double), ("step_size", C.c_double),
        ("drude_steps_per_real_step", C.c_int32), ("num_nh_chains", C.c_int32),
        ("use_drude_nh_chains", C.c_int32), ("use_com_temp_group", C.c_int32),
        ("max_drude_distance", C.c_double),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)

# name -> (restype, argtypes); every symbol include/drude_tgnh.h declares
SIGNATURES = {
    "tgnh_last_error": (C.c_char_p, []),
    "tgnh_abi_version": (C.c_int, []),
    "tgnh_create": (C.c_int, [C.POINTER(TgnhDesc), C.POINTER(C.c_void_p)]),
    "tgnh_destroy": (C.c_int, [C.c_void_p]),
    "tgnh_bind_buffers": (C.c_int, [C.c_void_p] * 6),
    "tgnh_set_step_size": (C.c_int, [C.c_void_p, C.c_double]),
    "tgnh_set_drude_steps_per_real_step": (C.c_int, [C.c_void_p, C.c_int]),
    "tgnh_set_max_drude_distance": (C.c_int, [C.c_void_p, C.c_double]),
    "tgnh_get_local_dof_terms": (C.c_int, [C.c_void_p, c_f64p, C.POINTER(C.c_int)]),
    "tgnh_set_global_dof_terms": (C.c_int, [C.c_void_p, c_f64p, C.c_int]),
    "tgnh_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "tgnh_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "tgnh_rccl_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tgnh_set_rccl_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_rccl_shutdown": (C.c_int, [C.c_void_p]),
    "tgnh_exchange_wait_stats": (C.c_int, [C.c_void_p, C.c_void_p, c_f64p, c_f64p, C.POINTER(C.c_int64)]),
    "tgnh_get_pending_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "tgnh_set_resident_share": (C.c_int, [C.c_void_p, C.c_int]),
    "tgnh_get_resident_work_groups": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "tgnh_get_resident_kernel": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "tgnh_get_step_path": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "tgnh_exchange_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "tgnh_exchange_attach": (C.c_int, [C.c_void_p, C.c_char_p]),
    "tgnh_exchange_attach_pointers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "tgnh_exchange_detach": (C.c_int, [C.c_void_p]),
    "tgnh_step_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_step_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_step_begin_kick": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_step_begin_move": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_step_end_kick": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_step_end_thermo": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_flush": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_state_changed": (C.c_int, [C.c_void_p]),
    "tgnh_note_replayed_steps": (C.c_int, [C.c_void_p, C.c_int]),
    "tgnh_set_time": (C.c_int, [C.c_void_p, C.c_double, C.c_int64]),
    "tgnh_get_kinetic_energy": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, c_f64p]),
    "tgnh_get_num_thermostats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "tgnh_get_last_kinetic_energies": (C.c_int, [C.c_void_p, C.c_void_p, c_f64p]),
    "tgnh_get_last_scale_factors": (C.c_int, [C.c_void_p, C.c_void_p, c_f64p]),
    "tgnh_get_status_flags": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "tgnh_get_time": (C.c_int, [C.c_void_p, c_f64p, C.POINTER(C.c_int64)]),
    "tgnh_get_dof": (C.c_int, [C.c_void_p, c_f64p, c_f64p]),
    "tgnh_get_thermostat_len": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "tgnh_get_thermostat_state": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, c_f64p]),
    "tgnh_set_thermostat_state": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, c_f64p]),
    "tgnh_get_topology_len": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "tgnh_get_topology": (C.c_int, [C.c_void_p, C.c_int, c_i32p]),
    "tgnh_get_launch_bounds": (C.c_int, [C.c_void_p, c_i32p]),
    "tgnh_compute_kinetic_energies": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_half_kick": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_harness_force": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "tgnh_harness_pack_sites": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_harness_lattice_hint": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, c_f64p, C.c_int]),
    "tgnh_harness_sites_kind": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "tgnh_run_harness": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "tgnh_run_steps": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "tgnh_harness_set_clusters": (C.c_int, [C.c_void_p, C.c_int, c_i32p, c_f64p]),
    "tgnh_harness_shake_positions": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p]),
    "tgnh_harness_shake_velocities": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p]),
    "tgnh_harness_set_virtual_sites": (C.c_int, [C.c_void_p, C.c_int, c_i32p, c_f64p]),
    "tgnh_harness_virtual_sites": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_run_harness_constrained": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "tgnh_harness_water_force": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "tgnh_harness_remove_cm_motion": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tgnh_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "tgnh_timing_read": (C.c_int, [C.c_void_p, C.c_int, c_f64p, C.POINTER(C.c_int64)]),
    "tgnh_algorithmic_bytes": (C.c_int, [C.c_void_p, C.c_int, c_f64p]),
}

_lib = None


def load():
    """Loads the HIP library; raises loudly when it is missing (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m openmm_drudenose_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the integrator path.")
    # ONE HIP runtime per process.  PyTorch ships its own libamdhip64 / libhsa-runtime64; this library links the system's
    # (/opt/rocm).  Loaded after torch, its libamdhip64.so.7 resolves to the copy the process already holds and everybody talks
    # to the same runtime; loaded BEFORE torch, the process ends up with two, and the second to touch the device finds none
    # ("hipGetDeviceCount: no ROCm-capable device is detected" -- build() followed by smoke() in one interpreter, round 4).  The
    # Python side of this package hands the library torch's device memory anyway, so torch goes first whenever it is there.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("TGNH_LIB") and not hasattr(lib, name):
            continue                # older tuning build selected with TGNH_LIB: tolerate symbols it predates
        fn = getattr(lib, name)     # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
