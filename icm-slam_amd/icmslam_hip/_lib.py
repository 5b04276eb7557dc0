"""ctypes binding of the C-ABI in include/icmslam.h + include/icmslam_tuning.h (libicmslam_hip.so).

There is no CPU fallback: if the shared library has not been built (or no MI355X is
present when a solver is created) the import / constructor fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libicmslam_hip.so")

ICM_OK = 0
ICM_ERR_ARG = -1
ICM_ERR_HIP = -2
ICM_ERR_INDEX = -3
ICM_ERR_EMPTY_MAP = -4
ICM_ERR_CAPACITY = -5
ICM_ERR_UNSUPPORTED = -6
RETRY_CAREFUL = 1000   # ICM_RETRY_CAREFUL (include/icmslam.h)
SCHEDULES = {"sequential": 0, "redblack": 1}


class IcmConfig(C.Structure):
    _fields_ = [("deltat", C.c_double), ("Q", C.c_double * 2), ("R", C.c_double * 3),
                ("cte_odom", C.c_double), ("cota", C.c_double), ("dist_thr", C.c_double),
                ("rango_laser_max", C.c_double), ("L", C.c_int64)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_H = C.c_void_p

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)   # icm_allgather_fn

# name -> (restype, argtypes); every symbol declared in include/*.h
SIGNATURES = {
    "icm_create": (C.c_int, [C.POINTER(IcmConfig), C.c_int, C.POINTER(_H)]),
    "icm_destroy": (C.c_int, [_H]),
    "icm_last_error": (C.c_char_p, [_H]),
    "icm_set_stream": (C.c_int, [_H, C.c_void_p]),
    "icm_upload": (C.c_int, [_H, _dp, _dp, _dp, _dp, _dp, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "icm_prefilter": (C.c_int, [_H, _lp]),
    "icm_get_kept": (C.c_int, [_H, _lp, _ip, _dp, _dp, _dp]),
    "icm_sweep": (C.c_int, [_H, _dp, _dp, _dp, C.c_int64, C.c_int64, C.c_int, _dp, _dp, _lp]),
    "icm_pin_host": (C.c_int, [_H, C.c_void_p, C.c_size_t]),
    "icm_unpin_host": (C.c_int, [_H, C.c_void_p]),
    "icm_set_state": (C.c_int, [_H, _dp, _dp, _dp, C.c_int64, C.c_int64]),
    "icm_sweep_device": (C.c_int, [_H, C.c_int]),
    "icm_get_state": (C.c_int, [_H, _dp, _dp, _dp, _lp]),
    "icm_stats_stride": (C.c_int64, [_H]),
    "icm_bind_exchange": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int]),
    "icm_bind_exchange_send": (C.c_int, [_H, C.c_void_p]),
    "icm_upload_ghost_scan": (C.c_int, [_H, _dp]),
    "icm_shard_block": (C.c_int64, [C.c_int64, C.c_int]),
    "icm_comm_init_transport": (C.c_int, [_H, C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p]),
    "icm_bind_pose_buffer": (C.c_int, [_H, C.c_void_p]),
    "icm_pose_buffer": (C.c_void_p, [_H]),
    "icm_comm_set_library": (C.c_int, [C.c_char_p]),
    "icm_comm_available": (C.c_int, []),
    "icm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "icm_comm_init": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int]),
    "icm_comm_destroy": (C.c_int, [_H]),
    "icm_sweep_sharded": (C.c_int, [_H]),
    "icm_gather_poses": (C.c_int, [_H]),
    "icm_sharded_end": (C.c_int, [_H]),
    "icm_set_optimistic": (C.c_int, [_H, C.c_int]),
    "icm_sweep_local": (C.c_int, [_H]),
    "icm_sweep_targets": (C.c_int, [_H]),
    "icm_sweep_solve": (C.c_int, [_H, C.c_int, C.c_int]),
    "icm_sweep_finish": (C.c_int, [_H]),
    "icm_mark_failed": (C.c_int, [_H, C.c_int]),
    "icm_failed_rank": (C.c_int, [_H, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "icm_exchange_status": (C.c_int, [_H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "icm_get_optimistic": (C.c_int, [_H]),
    "icm_set_fault": (C.c_int, [_H, C.c_int]),
    "icm_set_phase_timing": (C.c_int, [_H, C.c_int]),
    "icm_get_phase_times": (C.c_int, [_H, _dp, _lp]),
    "icm_get_association": (C.c_int, [_H, _ip, _dp, _dp]),
    "icm_get_raw_map": (C.c_int, [_H, _dp, _dp, _lp]),
    "icm_solve_one": (C.c_int, [_H, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int64, _dp]),
    "icm_energy_one": (C.c_int, [_H, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int64, _dp]),
    "icm_cluster_first_scan": (C.c_int, [_dp, C.c_int64, C.c_double, _ip]),
    "icm_associate": (C.c_int, [_H, _dp, C.c_int64, _dp, C.c_int64, _dp, _dp, _lp, _lp]),
    "icm_init_pass": (C.c_int, [_H, _dp, _dp, _dp, _lp, _dp]),
    "icm_filtrar": (C.c_int, [C.POINTER(IcmConfig), _dp, _dp, C.c_int64, _dp, _dp, _lp]),
    "icm_enable_timing": (C.c_int, [_H, C.c_int]),
    "icm_reset_timing": (C.c_int, [_H]),
    "icm_kernel_count": (C.c_int, [_H]),
    "icm_kernel_time": (C.c_int, [_H, C.c_int, C.POINTER(C.c_char_p), _dp, _lp]),
    "icm_last_stats": (C.c_int, [_H, _lp]),
    "icm_set_brute_force": (C.c_int, [_H, C.c_int]),
    "icm_get_wait_giveups": (C.c_int, [_H, _lp]),
    "icm_set_assoc_form": (C.c_int, [_H, C.c_int]),
    "icm_get_run_counts": (C.c_int, [_H, _lp]),
    "icm_get_runs": (C.c_int, [_H, _lp, _dp, _dp, C.POINTER(C.c_float), _ip, _ip]),
    "icm_set_debug": (C.c_int, [_H, C.c_int]),
    "icm_get_solve_diag": (C.c_int, [_H, _dp]),
    "icm_set_gpu_filtrar": (C.c_int, [_H, C.c_int]),
    "icm_last_filtrar_info": (C.c_int, [_H, _lp]),
    "icm_filtrar_device": (C.c_int, [_H, _dp, _dp, C.c_int64, _dp, _dp, _lp, C.POINTER(C.c_int)]),
    "icm_set_solve_lanes": (C.c_int, [_H, C.c_int]),
    "icm_snapshot_state": (C.c_int, [_H]),
    "icm_restore_state": (C.c_int, [_H]),
    "icm_set_colour_fusion": (C.c_int, [_H, C.c_int]),
    "icm_set_fold_mode": (C.c_int, [_H, C.c_int]),
    "icm_get_fixup_poses": (C.c_int, [_H, _lp]),
    "icm_get_dropin_counts": (C.c_int, [_H, _lp]),
    "icm_staging_layout": (C.c_int, [C.c_int64, C.c_int64, _lp]),
    "icm_set_fused_spin_limit": (C.c_int, [_H, C.c_int]),
    "icm_get_fused_deferred": (C.c_int, [_H, _lp]),
    "icm_set_entry_path": (C.c_int, [_H, C.c_int]),
    "icm_get_entry_path": (C.c_int, [_H]),
    "icm_set_energy_form": (C.c_int, [_H, C.c_int]),
    "icm_version": (C.c_char_p, []),
    "icm_build_id": (C.c_char_p, []),
    "icm_flop_per_eval": (C.c_int, []),
    "icm_valu_per_eval": (C.c_int, []),
}

_lib = None


def _torch_lib(name):
    """Path of a library a PyTorch-ROCm wheel bundles (None if there is no such wheel / file); torch is not imported."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    for d in spec.submodule_search_locations:
        cand = os.path.join(d, "lib", name)
        if os.path.exists(cand):
            return cand
    return None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  A process must use ONE HIP
    runtime: if ours (the system ROCm) were loaded first, a later `import torch` would find
    "No HIP GPUs".  So when a torch wheel with a bundled runtime is installed, load that copy
    first (same SONAME, so libicmslam_hip.so then binds to it); torch itself is not imported."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for d in spec.submodule_search_locations:
        cand = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load():
    """Load libicmslam_hip.so and declare every entry point.  Raises ImportError if the
    library is missing -- build it with `python __graft_entry__.py` (or `make -C
    icm-slam_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libicmslam_hip.so not found at %s: the HIP extension is not built and there is no "
            "CPU fallback. Run `make -C icm-slam_amd/csrc` (hipcc, --offload-arch=gfx950)." % LIB_PATH)
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # same for RCCL, but lazily: the library dlopen()s it only when collectives are asked for, and then takes the
    # wheel's copy unless the process has one loaded already (two copies in one process crash at exit)
    rccl = _torch_lib("librccl.so")
    if rccl:
        lib.icm_comm_set_library(rccl.encode())
    _lib = lib
    return lib


def dptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def iptr(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def lptr(a):
    return a.ctypes.data_as(_lp) if a is not None else None
