"""ctypes binding of libpeapods_amd.so (include/peapods_amd.h).

Loading fails loudly when the shared library is missing: there is no Python or
CPU fallback for the hot path.
"""

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpeapods_amd.so")

PA_OK, PA_ERR_INVALID, PA_ERR_UNSUPPORTED, PA_ERR_DEVICE, PA_ERR_INTERRUPTED = range(5)
PA_COMM_ID_BYTES = 128


class PaSimDesc(C.Structure):
    _fields_ = [
        ("n_dims", C.c_uint32),
        ("shape", C.POINTER(C.c_uint64)),
        ("n_neighbors", C.c_uint32),
        ("neighbor_offsets", C.POINTER(C.c_int64)),
        ("couplings", C.POINTER(C.c_float)),
        ("n_realizations", C.c_uint64),
        ("temperatures", C.POINTER(C.c_float)),
        ("n_temps", C.c_uint64),
        ("n_replicas", C.c_uint64),
        ("seed", C.c_uint64),
        ("device", C.c_int32),
        ("first_realization", C.c_uint64),
        ("stream_groups", C.c_uint32),
    ]


class PaSampleConfig(C.Structure):
    _fields_ = [
        ("n_sweeps", C.c_uint64),
        ("warmup_sweeps", C.c_uint64),
        ("sweep_mode", C.c_int32),
        ("pt_interval", C.c_uint64),
        ("pt_schedule", C.c_int32),
        ("overlap_cluster_interval", C.c_uint64),
        ("overlap_cluster_modes", C.c_int32 * 4),
        ("n_overlap_cluster_modes", C.c_int32),
        ("overlap_cluster_mode", C.c_int32),
        ("measure_overlap", C.c_int32),
        ("cluster_interval", C.c_uint64),
        ("cluster_mode", C.c_int32),
        ("equilibration_diagnostic", C.c_int32),
        ("autocorrelation_max_lag", C.c_uint64),
        ("collect_cluster_stats", C.c_int32),
        ("cluster_action", C.c_int32),
        ("overlap_cluster_action", C.c_int32),
        ("autocorrelation_backend", C.c_int32),
        ("snapshot_interval", C.c_uint64),
    ]


_u64, _i64, _u32, _i32, _vp = C.c_uint64, C.c_int64, C.c_uint32, C.c_int32, C.c_void_p

# every symbol include/peapods_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "pa_sim_create": (C.c_int, [C.POINTER(PaSimDesc), C.POINTER(_vp), C.c_char_p, C.c_size_t]),
    "pa_sim_destroy": (None, [_vp]),
    "pa_sim_reset": (C.c_int, [_vp, _u64]),
    "pa_sim_last_error": (C.c_char_p, [_vp]),
    "pa_validate_sample_config": (C.c_int, [C.POINTER(PaSampleConfig), _u64, C.c_char_p, C.c_size_t]),
    "pa_sim_sample": (C.c_int, [_vp, C.POINTER(PaSampleConfig)]),
    "pa_sim_sample_async": (C.c_int, [_vp, C.POINTER(PaSampleConfig)]),
    "pa_sim_synchronize": (C.c_int, [_vp]),
    "pa_sim_request_interrupt": (None, [_vp]),
    "pa_sim_sweeps_completed": (_u64, [_vp]),
    "pa_sim_profile_step": (C.c_int, [_vp, C.POINTER(PaSampleConfig), _u64, _vp, _u64, _vp, _vp, _u64, _vp, _vp]),
    "pa_sim_metropolis_sweep": (C.c_int, [_vp, _u64]),
    "pa_sim_gibbs_sweep": (C.c_int, [_vp, _u64]),
    "pa_sim_compute_energies": (C.c_int, [_vp]),
    "pa_sim_parallel_tempering": (C.c_int, [_vp, _i32]),
    "pa_sim_overlap_dots": (C.c_int, [_vp, _vp, _vp]),
    "pa_sim_n_spins": (_u64, [_vp]),
    "pa_sim_n_neighbors": (_u64, [_vp]),
    "pa_sim_n_systems": (_u64, [_vp]),
    "pa_sim_get_spins": (C.c_int, [_vp, _u64, _vp]),
    "pa_sim_get_system_ids": (C.c_int, [_vp, _u64, _vp]),
    "pa_sim_get_energies": (C.c_int, [_vp, _u64, _vp]),
    "pa_sim_get_magnetization_sums": (C.c_int, [_vp, _u64, _vp]),
    "pa_sim_get_rng_state": (C.c_int, [_vp, _u64, _u64, _vp]),
    "pa_sim_set_spins": (C.c_int, [_vp, _u64, _vp]),
    "pa_sim_set_system_ids": (C.c_int, [_vp, _u64, _vp]),
    "pa_sim_set_rng_state": (C.c_int, [_vp, _u64, _u64, _vp]),
    "pa_sim_get_pt_counters": (C.c_int, [_vp, _u64, _vp, _vp, _vp]),
    "pa_sim_get_result_moments": (C.c_int, [_vp, _u64, _vp]),
    "pa_sim_get_result_overlap": (C.c_int, [_vp, _u64, _vp]),
    "pa_sim_get_result_histograms": (C.c_int, [_vp, _u64, _vp, _vp, _vp]),
    "pa_sim_result_block_len": (_u64, [_vp]),
    "pa_sim_get_autocorrelation_tau": (C.c_int, [_vp, _u64, _vp, _vp]),
    "pa_sim_n_equil_checkpoints": (_u64, [_vp]),
    "pa_sim_get_equilibration": (C.c_int, [_vp, _u64, _vp, _vp, _vp]),
    "pa_sim_get_cluster_stats": (C.c_int, [_vp, _u64, _vp, _vp, _vp]),
    "pa_sim_get_observation": (C.c_int, [_vp, _u64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pa_sim_n_snapshots": (_u64, [_vp]),
    "pa_sim_get_snapshot": (C.c_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pa_sim_get_result_blocks": (C.c_int, [_vp, _vp]),
    "pa_splitmix64": (_u64, [_u64]),
    "pa_child_seed": (_u64, [_u64, _u64, _u64]),
    "pa_realization_seed": (_u64, [_u64, _u64]),
    "pa_xoshiro_state": (None, [_u64, _u64, _vp]),
    "pa_xoshiro_jump": (None, [_vp, _u64, _vp]),
    "pa_build_metropolis_table": (C.c_int, [_vp, _u64, _vp, _u64, _u64, _vp]),
    "pa_build_gibbs_table": (C.c_int, [_vp, _u64, _vp, _u64, _u64, _vp]),
    "pa_build_sweep_levels": (_u64, [_u32, _vp, _u32, _vp, _vp]),
    "pa_autocorrelation_tau": (None, [_vp, _u64, _u64, _u64, _vp, _i32]),
    "pa_aggregate_mean": (None, [_vp, _u64, _u64, _vp]),
    "pa_aggregate_sum_u64": (None, [_vp, _u64, _u64, _vp]),
    "pa_aggregate_sum_f64": (None, [_vp, _u64, _u64, _vp]),
    "pa_comm_unique_id": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "pa_comm_init_rank": (C.c_int, [_vp, _i32, _i32, _i32, C.POINTER(_vp), C.c_char_p, C.c_size_t]),
    "pa_comm_destroy": (None, [_vp]),
    "pa_comm_last_error": (C.c_char_p, [_vp]),
    "pa_comm_n_ranks": (_i32, [_vp]),
    "pa_comm_rank": (_i32, [_vp]),
    "pa_comm_allgather": (C.c_int, [_vp, _vp, _vp, _u64]),
    "pa_comm_allreduce_max_f64": (C.c_int, [_vp, _vp, _u64]),
    "pa_comm_allreduce_sum_u64": (C.c_int, [_vp, _vp, _u64]),
    "pa_comm_barrier": (C.c_int, [_vp]),
    "pa_comm_gatherv": (C.c_int, [_vp, _vp, _u64, _vp, _u64, _i32, _vp]),
    "pa_comm_gather_result_blocks": (C.c_int, [_vp, _vp, _vp, _u64, _vp]),
    "pa_sim_n_realizations": (_u64, [_vp]),
    "pa_device_copy_bandwidth": (C.c_int, [_i32, _u64, _i32, _vp, C.c_char_p, C.c_size_t]),
    "pa_abi_layout": (_u64, [_vp, _u64]),
    "pa_device_count": (C.c_int, []),
    "pa_version": (C.c_char_p, []),
    "pa_active_switches": (C.c_char_p, []),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m peapods_amd.build` "
                "(hipcc, gfx950). peapods_amd has no fallback path."
            )
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def ptr(array):
    return array.ctypes.data_as(C.c_void_p)
