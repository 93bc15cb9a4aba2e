"""ctypes binding of libsubzero_hip.so (the C-ABI of include/subzero_hip.h).

There is NO CPU fallback: if the library or a HIP device is missing, every entry point raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_fp = C.POINTER(C.c_float)

OPEN, PERIODIC, COLLISION, MOVING = 0, 1, 2, 3
NORTH, SOUTH, EAST, WEST = 0, 1, 2, 3
ACTIVE, REMOVE, FUSE = 1, 2, 3
COLLISIONS_ON, COUPLING_ON, NO_STOP = 1, 2, 4
K_GHOSTS, K_BROAD, K_NARROW, K_REDUCE, K_FORCING, K_INTEGRATE, K_NARROW_LARGE = range(7)
KERNEL_CLASS_NAMES = ["ghosts", "broad", "narrow", "reduce", "forcing", "integrate", "narrow_large", "exchange"]

DCOLS = ["cx", "cy", "rmax", "area", "height", "mass", "moment", "alpha", "u", "v", "xi",
         "p_dxdt", "p_dydt", "p_dalphadt", "p_dudt", "p_dvdt", "p_dxidt",
         "fxOA", "fyOA", "trqOA", "hflx_factor", "overarea", "coll_fx", "coll_fy", "coll_trq"]
TCOLS = ["stress_accum", "stress_instant", "strain"]


class SzParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("E", "nu", "mu", "rho_o", "rho_a", "Cd_io", "Cd_ia", "f", "turn_theta",
                 "floe_floe_max_overlap", "floe_domain_max_overlap",
                 "rho_i", "max_floe_height", "maximum_xi", "lam")] + [("coupling_dd", C.c_int32), ("_pad", C.c_int32)]


class SzFloeColumns(C.Structure):
    _fields_ = ([(n, _dp) for n in DCOLS] + [(n, _dp) for n in TCOLS] +
                [("id", _lp), ("ghost_id", _lp), ("status", _ip),
                 ("vert_off", _ip), ("vx", _dp), ("vy", _dp),
                 ("sub_off", _ip), ("sx", _dp), ("sy", _dp),
                 ("ghost_off", _ip), ("ghost_idx", _ip)])


class SzFloeColumnsF32(C.Structure):
    """sz_floe_columns_f32: the columns of a Floe{Float32} host (the engine widens on the way in, rounds on the way out)"""
    _fields_ = ([(n, _fp) for n in DCOLS] + [(n, _fp) for n in TCOLS] +
                [("id", _lp), ("ghost_id", _lp), ("status", _ip),
                 ("vert_off", _ip), ("vx", _fp), ("vy", _fp),
                 ("sub_off", _ip), ("sx", _fp), ("sy", _fp),
                 ("ghost_off", _ip), ("ghost_idx", _ip)])


class SzStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("M", "N", "n_ring_points", "n_sub_points", "n_pairs", "n_pair_ring_points", "n_pair_rows",
                 "n_elem_items", "n_elem_rows", "n_inter_rows", "n_ghosts",
                 "warn_height", "warn_force", "warn_vel", "warn_xi", "n_trace_fail", "n_halo", "n_pairs_clipped", "n_status_remove", "n_status_fuse", "n_retry",
                 "acc_narrow_launches", "acc_pair_items", "acc_pair_ring_points", "acc_pair_rows", "acc_elem_items", "acc_elem_rows", "acc_dir_checks", "acc_dir_checks_certified")]


_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
_SR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, _ip, C.POINTER(C.c_void_p), _lp, C.POINTER(C.c_void_p), _lp)
_AR = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.c_int64)


class SzHostTransport(C.Structure):
    """sz_host_transport of include/subzero_hip.h: the host's own collectives for sz_comm_init_host"""
    _fields_ = [("user", C.c_void_p), ("allgather", _AG), ("sendrecv", _SR), ("allreduce_sum_f64", _AR)]


EXPORTS = [
    "sz_create", "sz_destroy", "sz_last_error", "sz_version", "sz_set_params", "sz_set_domain",
    "sz_set_topography", "sz_set_fields", "sz_upload_floes", "sz_get_stats", "sz_download_floes",
    "sz_download_interactions", "sz_download_pairs", "sz_download_fuse", "sz_get_boundary_vals",
    "sz_add_ghosts", "sz_remove_ghosts", "sz_timestep_collisions", "sz_collide_pairs", "sz_collide_domain",
    "sz_timestep_coupling", "sz_timestep_floe_properties", "sz_step",
    "sz_upload_interactions", "sz_calc_stress", "sz_calc_strain",
    "sz_set_two_way", "sz_set_temps", "sz_download_ocean_stress", "sz_two_way_partial", "sz_two_way_finish", "sz_set_precision",
    "sz_eulerian_data", "sz_eulerian_partial", "sz_eulerian_finish", "sz_simplify_check",
    "sz_profile_enable", "sz_profile_reset", "sz_kernel_time_ms", "sz_forcing_launch", "sz_narrow_kernel_name",
    "sz_tile_enable", "sz_owned_box", "sz_halo_record_doubles", "sz_halo_record_doubles_ctx", "sz_halo_set_boxes", "sz_halo_pack", "sz_halo_counts", "sz_tile_forcing", "sz_tile_step", "sz_sync", "sz_set_stream", "sz_debug_stamps", "sz_debug_crec_mismatches", "sz_upload_floes_f32", "sz_download_floes_f32", "sz_set_fields_f32", "sz_download_interactions_f32",
    "sz_get_boundary_rects", "sz_debug_match_vertices", "sz_debug_sample_fields", "sz_debug_pipelined",
    "sz_comm_available", "sz_comm_unique_id", "sz_comm_init", "sz_comm_init_host", "sz_comm_destroy", "sz_comm_selftest", "sz_comm_allreduce", "sz_tile_setup", "sz_tile_set_center", "sz_tile_run", "sz_tile_migrate", "sz_tile_owned_gidx", "sz_debug_migrate_path", "sz_debug_find_key", "sz_debug_pairs_of_ids", "sz_download_subpoints",
]

EUL_PARTIAL = 17      # SZ_EUL_PARTIAL: per-cell partial fields of sz_eulerian_partial

_LIB = None


def lib_path():
    # (SZ_LIB_PATH: another build of the same library -- A/B experiments on compile-time choices, tools/ only)
    return os.environ.get("SZ_LIB_PATH") or _build.LIB


def load(build_if_missing=True):
    """Loads the shared library (no GPU needed for this); raises if it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise RuntimeError(f"{path} not built; run __graft_entry__.build()")
        _build.build()
    L = C.CDLL(path)
    L.sz_create.restype = C.c_void_p
    L.sz_create.argtypes = [C.c_int]
    L.sz_destroy.argtypes = [C.c_void_p]
    L.sz_last_error.restype = C.c_char_p
    L.sz_last_error.argtypes = [C.c_void_p]
    L.sz_version.restype = C.c_char_p
    L.sz_set_params.argtypes = [C.c_void_p, C.POINTER(SzParams)]
    L.sz_set_domain.argtypes = [C.c_void_p, _ip, _dp, _dp, _dp, _dp]
    L.sz_set_topography.argtypes = [C.c_void_p, C.c_int32, _ip, _dp, _dp, _dp, _dp, _dp]
    L.sz_set_fields.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                _dp, _dp, _dp, _dp, _dp]
    L.sz_upload_floes.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(SzFloeColumns)]
    L.sz_get_stats.argtypes = [C.c_void_p, C.POINTER(SzStats)]
    L.sz_download_floes.argtypes = [C.c_void_p, C.POINTER(SzFloeColumns)]
    L.sz_download_interactions.argtypes = [C.c_void_p, _ip, _dp]
    L.sz_upload_floes_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(SzFloeColumnsF32)]
    L.sz_download_floes_f32.argtypes = [C.c_void_p, C.POINTER(SzFloeColumnsF32)]
    L.sz_set_fields_f32.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, _fp, _fp, _fp, _fp, _fp]
    L.sz_download_interactions_f32.argtypes = [C.c_void_p, _ip, _fp]
    L.sz_download_pairs.argtypes = [C.c_void_p, _ip, _ip]
    L.sz_download_fuse.argtypes = [C.c_void_p, _ip, _ip]
    L.sz_get_boundary_vals.argtypes = [C.c_void_p, _dp]
    L.sz_add_ghosts.argtypes = [C.c_void_p]
    L.sz_remove_ghosts.argtypes = [C.c_void_p]
    L.sz_timestep_collisions.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    L.sz_collide_pairs.argtypes = [C.c_void_p, C.c_int64, _ip, _ip, C.c_int32, C.c_double]
    L.sz_collide_domain.argtypes = [C.c_void_p, C.c_int32, C.c_double]
    L.sz_timestep_coupling.argtypes = [C.c_void_p]
    L.sz_timestep_floe_properties.argtypes = [C.c_void_p, C.c_int32]
    L.sz_step.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _ip]
    L.sz_set_two_way.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32]
    L.sz_set_temps.argtypes = [C.c_void_p, _dp, _dp]
    L.sz_set_precision.argtypes = [C.c_void_p, C.c_int32]
    L.sz_two_way_partial.argtypes = [C.c_void_p, C.c_void_p]
    L.sz_two_way_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.sz_download_ocean_stress.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
    L.sz_eulerian_data.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, C.c_int32, _ip, _dp]
    L.sz_eulerian_partial.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, C.c_void_p]
    L.sz_eulerian_finish.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, C.c_void_p, C.c_int32, _ip, _dp]
    L.sz_simplify_check.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, _lp]
    L.sz_upload_interactions.argtypes = [C.c_void_p, _ip, _dp]
    L.sz_calc_stress.argtypes = [C.c_void_p]
    L.sz_calc_strain.argtypes = [C.c_void_p]
    L.sz_profile_enable.argtypes = [C.c_void_p, C.c_int32]
    L.sz_profile_reset.argtypes = [C.c_void_p]
    L.sz_kernel_time_ms.argtypes = [C.c_void_p, C.c_int32, _dp, _lp]
    L.sz_forcing_launch.argtypes = [C.c_void_p, _ip]
    L.sz_tile_enable.argtypes = [C.c_void_p, _lp, C.c_double, C.c_double]
    L.sz_owned_box.argtypes = [C.c_void_p, _dp]
    L.sz_halo_record_doubles.argtypes = []
    L.sz_halo_record_doubles_ctx.argtypes = [C.c_void_p]
    L.sz_halo_set_boxes.argtypes = [C.c_void_p, C.c_int32, _dp]
    L.sz_halo_pack.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32,
                               C.c_void_p, C.c_int32]
    L.sz_halo_counts.argtypes = [C.c_void_p, C.c_int32, _ip]
    L.sz_tile_forcing.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.sz_tile_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.sz_sync.argtypes = [C.c_void_p]
    L.sz_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.sz_debug_stamps.argtypes = [C.c_void_p, _lp]
    L.sz_debug_crec_mismatches.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.sz_comm_available.argtypes = []
    L.sz_narrow_kernel_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    L.sz_comm_unique_id.argtypes = [C.c_void_p]
    L.sz_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.sz_comm_init_host.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SzHostTransport)]
    L.sz_comm_destroy.argtypes = [C.c_void_p]
    L.sz_comm_selftest.argtypes = [C.c_void_p]
    L.sz_comm_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.sz_tile_setup.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_double, C.c_int32]
    L.sz_tile_run.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _ip]
    L.sz_tile_set_center.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.sz_tile_migrate.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _ip, _lp, _lp]
    L.sz_download_subpoints.argtypes = [C.c_void_p, _ip, _dp, _dp]
    L.sz_tile_owned_gidx.argtypes = [C.c_void_p, _lp, C.c_int64]
    L.sz_debug_migrate_path.argtypes = [C.c_void_p]
    L.sz_debug_find_key.argtypes = [C.c_void_p, C.c_int32, C.c_int64, _dp]
    L.sz_debug_pairs_of_ids.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, _dp]
    L.sz_get_boundary_rects.argtypes = [C.c_void_p, _dp]
    L.sz_debug_match_vertices.argtypes = [C.c_void_p, C.c_int32, _dp, _dp, C.c_int32, _dp, _dp, _ip, _ip]
    L.sz_debug_sample_fields.argtypes = [C.c_void_p, C.c_int32, _dp, _dp, _dp]
    L.sz_debug_pipelined.argtypes = [C.c_void_p]
    for n in EXPORTS:
        if n not in ("sz_create", "sz_destroy", "sz_last_error", "sz_version"):
            getattr(L, n).restype = C.c_int
    _LIB = L
    return L


def ptr(a, t=_dp):
    return None if a is None else a.ctypes.data_as(t)


class SzError(RuntimeError):
    pass
