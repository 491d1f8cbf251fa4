"""ctypes binding of the C ABI in include/karto_hip.h (libkartohip.so).

The library is the product: if it is missing or cannot be loaded this module raises -- there is
no Python / CPU fallback path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KH_LIBRARY") or os.path.join(HERE, "libkartohip.so")     # KH_LIBRARY: another build of the same sources (measurements)

KH_OK, KH_ERR_INVALID_ARG, KH_ERR_NO_DEVICE, KH_ERR_HIP, KH_ERR_SEARCH, KH_ERR_NOT_FOUND, KH_ERR_SOLVER, KH_ERR_IO = range(8)
KH_GRAPH_TEXT, KH_GRAPH_BINARY = 0, 1

dptr = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
iptr = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
bptr = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class KhScan(C.Structure):
    _fields_ = [("n", C.c_int32), ("ranges", C.POINTER(C.c_double)), ("points_xy", C.POINTER(C.c_double)),
                ("sensor_pose", C.c_double * 3), ("device_points_xy", C.c_void_p)]


class KhMatchParams(C.Structure):
    _fields_ = [("coarse_search_angle_offset", C.c_double), ("coarse_angle_resolution", C.c_double),
                ("fine_search_angle_offset", C.c_double), ("use_response_expansion", C.c_int32),
                ("distance_variance_penalty", C.c_double), ("minimum_distance_penalty", C.c_double),
                ("angle_variance_penalty", C.c_double), ("minimum_angle_penalty", C.c_double)]


class KhGridInfo(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("width", "height", "width_step", "data_size", "roi_x", "roi_y", "roi_w",
                                          "roi_h", "kernel_size", "search_side")] + \
               [(k, C.c_double) for k in ("offset_x", "offset_y", "scale")]


KH_LOSS_NONE, KH_LOSS_HUBER, KH_LOSS_CAUCHY = 0, 1, 2


class KhSpaOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("min_relative_decrease", C.c_double), ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("max_num_consecutive_invalid_steps", C.c_int32), ("use_nonmonotonic_steps", C.c_int32),
                ("max_consecutive_nonmonotonic_steps", C.c_int32), ("jacobi_scaling", C.c_int32),
                ("loss_function", C.c_int32), ("loss_scale", C.c_double)]


class KhSpaSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("successful_steps", C.c_int32), ("termination", C.c_int32),
                ("usable", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("linearize_ms", C.c_double), ("solve_ms", C.c_double), ("total_ms", C.c_double),
                ("nnz_factor", C.c_int64), ("factor_flops", C.c_int64), ("factorizations", C.c_int32),
                ("levels", C.c_int32), ("factor_gpu_ms", C.c_double), ("backward_gpu_ms", C.c_double),
                ("linearize_gpu_ms", C.c_double), ("symbolic_ms", C.c_double),
                ("worst_linear_residual", C.c_double), ("analysis", C.c_int32), ("analysis_pad", C.c_int32)]


# every symbol include/karto_hip.h declares (tests check that the built library exports all of them)
SYMBOLS = [
    "kh_last_error", "kh_device_count", "kh_version", "kh_scan_points", "kh_match_params_default",
    "kh_matcher_create", "kh_matcher_destroy", "kh_matcher_set_params", "kh_matcher_match",
    "kh_matcher_match_batch", "kh_matcher_add_scans", "kh_matcher_correlate", "kh_matcher_correlate_batch",
    "kh_matcher_grid_info", "kh_matcher_read_grid", "kh_matcher_read_kernel", "kh_matcher_read_lookup",
    "kh_matcher_positional_covariance", "kh_matcher_angular_covariance",
    "kh_matcher_read_volume", "kh_matcher_set_debug", "kh_matcher_stream", "kh_matcher_profile", "kh_matcher_score_loads", "kh_matcher_seq_stats", "kh_matcher_profile_side",
    "kh_matcher_group_create", "kh_matcher_group_destroy", "kh_matcher_group_set_params", "kh_matcher_group_size", "kh_matcher_group_member",
    "kh_matcher_group_device", "kh_matcher_group_match_batch", "kh_loop_closure_batch",
    "kh_spa_options_default", "kh_spa_create", "kh_spa_set_debug", "kh_spa_destroy", "kh_spa_set_options", "kh_spa_reset",
    "kh_spa_clear", "kh_spa_add_node", "kh_spa_add_constraint", "kh_spa_remove_node",
    "kh_spa_remove_constraint", "kh_spa_modify_node", "kh_spa_get_node", "kh_spa_num_nodes",
    "kh_spa_num_constraints", "kh_spa_compute", "kh_spa_iteration_log", "kh_spa_get_corrections", "kh_link_info", "kh_spa_set_sharding",
    "kh_spa_save", "kh_spa_load", "kh_spa_add_constraint_information", "kh_spa_get_node_at", "kh_spa_get_constraint", "kh_spa_get_nodes",
    "kh_graph_create", "kh_graph_destroy", "kh_graph_set", "kh_graph_set_positions", "kh_graph_find_loop_candidates",
    "kh_graph_last_kernel_ms", "kh_graph_find_near_chains", "kh_graph_closest_scan_to_pose", "kh_weighted_mean",
    "kh_occupancy_compute_dimensions", "kh_occupancy_create", "kh_occupancy_destroy", "kh_occupancy_clear",
    "kh_occupancy_add_scans", "kh_occupancy_update", "kh_occupancy_read", "kh_occupancy_info",
    "kh_decay_params_default", "kh_lifelong_scores",
    "kh_spa_set_comm", "kh_comm_unique_id", "kh_comm_create", "kh_comm_destroy", "kh_comm_rank", "kh_comm_world", "kh_comm_device",
    "kh_comm_allreduce_sum_f64", "kh_comm_allgather_f64",
    "kh_device_malloc", "kh_device_free", "kh_device_upload", "kh_device_upload_on", "kh_device_download", "kh_selftest_lds_attr",
    "kh_graph_find_loop_candidates_from",
    "kh_mapper_params_default", "kh_mapper_create", "kh_mapper_create_on_devices", "kh_mapper_destroy", "kh_mapper_process", "kh_mapper_num_scans",
    "kh_mapper_num_edges", "kh_mapper_get_poses", "kh_mapper_get_scan", "kh_mapper_get_stats", "kh_mapper_solver",
    "kh_mapper_set_log", "kh_mapper_remove_node", "kh_mapper_get_adjacency", "kh_mapper_set_node_score", "kh_mapper_set_lifelong", "kh_mapper_num_alive", "kh_mapper_get_alive",
    "kh_graph_set_scan_limit", "kh_graph_find_near_linked", "kh_graph_append_scan", "kh_graph_add_edge", "kh_graph_set_position",
]


class KhScanBox(C.Structure):
    _fields_ = [("barycenter", C.c_double * 2), ("bbox_size", C.c_double * 2), ("unique_id", C.c_int32),
                ("n_edges", C.c_int32), ("score", C.c_double), ("n_points", C.c_int32),
                ("points_xy", C.POINTER(C.c_double))]


class KhDecayParams(C.Structure):
    _fields_ = [("iou_thresh", C.c_double), ("iou_match", C.c_double), ("removal_score", C.c_double),
                ("overlap_scale", C.c_double), ("constraint_scale", C.c_double), ("nearby_penalty", C.c_double),
                ("candidates_scale", C.c_double), ("scan_buffer_size", C.c_int32)]


class KhLaser(C.Structure):
    _fields_ = [("n_beams", C.c_int32), ("minimum_angle", C.c_double), ("angular_resolution", C.c_double),
                ("minimum_range", C.c_double), ("maximum_range", C.c_double), ("range_threshold", C.c_double),
                ("offset_x", C.c_double), ("offset_y", C.c_double), ("offset_heading", C.c_double)]


class KhMapperParams(C.Structure):
    _fields_ = [("use_scan_matching", C.c_int32), ("use_scan_barycenter", C.c_int32),
                ("minimum_time_interval", C.c_double), ("minimum_travel_distance", C.c_double),
                ("minimum_travel_heading", C.c_double), ("scan_buffer_size", C.c_int32),
                ("scan_buffer_maximum_scan_distance", C.c_double), ("link_match_minimum_response_fine", C.c_double),
                ("link_scan_maximum_distance", C.c_double), ("loop_search_maximum_distance", C.c_double),
                ("do_loop_closing", C.c_int32), ("loop_match_minimum_chain_size", C.c_int32),
                ("loop_match_maximum_variance_coarse", C.c_double), ("loop_match_minimum_response_coarse", C.c_double),
                ("loop_match_minimum_response_fine", C.c_double), ("correlation_search_space_dimension", C.c_double),
                ("correlation_search_space_resolution", C.c_double), ("correlation_search_space_smear_deviation", C.c_double),
                ("loop_search_space_dimension", C.c_double), ("loop_search_space_resolution", C.c_double),
                ("loop_search_space_smear_deviation", C.c_double), ("match", KhMatchParams)]


class KhMapperStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("scans_processed", "matches", "loop_candidates", "loop_closures", "speculation_discarded", "nodes_removed")] + \
               [(k, C.c_double) for k in ("process_ms", "match_ms", "solver_ms", "update_ms", "lifelong_ms")] + \
               [(k, C.c_int64) for k in ("fused_declined", "fused_declined_reason", "fused_matches", "fused_fine_passes")]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

_lib = None


class KartoHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = ""
        try:
            msg = lib().kh_last_error().decode()
        except Exception:
            pass
        super().__init__(f"{where}: status {code} {msg}")


def lib():
    """Loads libkartohip.so; raises if it has not been built (run `python -m slam_toolbox_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m slam_toolbox_amd.build` "
                          "(the HIP extension is the product; there is no fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i32, dbl = C.c_void_p, C.c_int32, C.c_double
    L.kh_last_error.restype = C.c_char_p
    L.kh_version.restype = C.c_char_p
    L.kh_device_count.restype = C.c_int
    L.kh_scan_points.argtypes = [dptr, i32, dptr, dbl, dbl, dptr]
    L.kh_match_params_default.argtypes = [C.POINTER(KhMatchParams)]
    L.kh_matcher_create.argtypes = [dbl, dbl, dbl, dbl, i32, i32, C.POINTER(vp)]
    L.kh_matcher_destroy.argtypes = [vp]
    L.kh_matcher_destroy.restype = None
    L.kh_matcher_set_params.argtypes = [vp, C.POINTER(KhMatchParams)]
    L.kh_matcher_match.argtypes = [vp, C.POINTER(KhScan), C.POINTER(KhScan), i32, i32, i32, dptr, dptr, C.POINTER(dbl)]
    L.kh_matcher_match_batch.argtypes = [vp, i32, C.POINTER(KhScan), C.POINTER(KhScan), iptr, i32, i32, dptr, dptr, dptr, iptr]
    if hasattr(L, "kh_matcher_group_create"):
        L.kh_matcher_group_create.argtypes = [dbl, dbl, dbl, dbl, iptr, i32, i32, C.POINTER(vp)]
        L.kh_matcher_group_destroy.argtypes = [vp]
        L.kh_matcher_group_destroy.restype = None
        L.kh_matcher_group_set_params.argtypes = [vp, C.POINTER(KhMatchParams)]
        L.kh_matcher_group_size.argtypes = [vp]
        L.kh_matcher_group_member.argtypes = [vp, i32]
        L.kh_matcher_group_member.restype = vp
        L.kh_matcher_group_device.argtypes = [vp, i32]
        L.kh_matcher_group_match_batch.argtypes = [vp, i32, C.POINTER(KhScan), C.POINTER(KhScan), iptr, vp, i32, i32, dptr, dptr, dptr, iptr]
        L.kh_loop_closure_batch.argtypes = [vp, vp, i32, C.POINTER(KhScan), C.POINTER(KhScan), iptr, dbl, dbl, dbl, dbl, i32,
                                            dptr, dptr, dptr, iptr, dptr, dptr, dptr]
    L.kh_matcher_add_scans.argtypes = [vp, i32, C.POINTER(KhScan), C.POINTER(KhScan), i32]
    L.kh_matcher_correlate.argtypes = [vp, i32, C.POINTER(KhScan), dptr, dptr, dptr, dbl, dbl, i32, i32, dptr, dptr, C.POINTER(dbl)]
    L.kh_matcher_correlate_batch.argtypes = [vp, i32, C.POINTER(KhScan), dptr, dptr, dptr, dbl, dbl, i32, i32, dptr, dptr, dptr, iptr]
    L.kh_matcher_positional_covariance.argtypes = [vp, i32, dptr, dbl, dptr, dptr, dptr, dbl, dptr]
    L.kh_matcher_angular_covariance.argtypes = [vp, i32, C.POINTER(KhScan), dptr, dbl, dptr, dbl, dbl, dptr]
    L.kh_matcher_grid_info.argtypes = [vp, i32, C.POINTER(KhGridInfo)]
    L.kh_matcher_read_grid.argtypes = [vp, i32, bptr]
    L.kh_matcher_read_kernel.argtypes = [vp, bptr]
    L.kh_matcher_read_lookup.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), vp]
    L.kh_matcher_read_volume.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), vp, vp]
    L.kh_matcher_set_debug.argtypes = [vp, i32]
    L.kh_matcher_stream.argtypes = [vp]
    L.kh_matcher_stream.restype = vp
    L.kh_matcher_profile.argtypes = [vp, i32, C.POINTER(dbl), C.POINTER(C.c_int64), C.POINTER(dbl), C.POINTER(C.c_int64)]
    L.kh_matcher_score_loads.argtypes = [vp, C.POINTER(C.c_int64), i32]
    L.kh_matcher_seq_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.kh_matcher_profile_side.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(L, "kh_spa_create"):
        L.kh_spa_options_default.argtypes = [C.POINTER(KhSpaOptions)]
        L.kh_spa_create.argtypes = [i32, C.POINTER(vp)]
        L.kh_spa_destroy.argtypes = [vp]
        L.kh_spa_destroy.restype = None
        L.kh_spa_set_debug.argtypes = [vp, i32]
        L.kh_spa_set_options.argtypes = [vp, C.POINTER(KhSpaOptions)]
        L.kh_spa_reset.argtypes = [vp]
        L.kh_spa_clear.argtypes = [vp]
        L.kh_spa_add_node.argtypes = [vp, i32, dptr]
        L.kh_spa_add_constraint.argtypes = [vp, i32, i32, dptr, dptr]
        L.kh_spa_remove_node.argtypes = [vp, i32]
        L.kh_spa_remove_constraint.argtypes = [vp, i32, i32]
        L.kh_spa_modify_node.argtypes = [vp, i32, dptr]
        L.kh_spa_get_node.argtypes = [vp, i32, dptr]
        L.kh_spa_num_nodes.argtypes = [vp]
        L.kh_spa_num_constraints.argtypes = [vp]
        L.kh_spa_compute.argtypes = [vp, C.POINTER(KhSpaSummary)]
        L.kh_spa_iteration_log.argtypes = [vp, C.c_int32, vp, C.POINTER(C.c_int32)]
        L.kh_spa_get_corrections.argtypes = [vp, C.POINTER(i32), vp, vp]
        L.kh_link_info.argtypes = [dptr, dptr, dptr, dptr, dptr]
        L.kh_spa_set_sharding.argtypes = [vp, i32, i32, ALLREDUCE_FN, vp]
        L.kh_spa_save.argtypes = [vp, C.c_char_p, i32]
        L.kh_spa_load.argtypes = [vp, C.c_char_p]
        L.kh_spa_add_constraint_information.argtypes = [vp, i32, i32, dptr, dptr]
        L.kh_spa_get_node_at.argtypes = [vp, i32, C.POINTER(i32), dptr]
        L.kh_spa_get_nodes.argtypes = [vp, vp, vp]
        L.kh_spa_get_constraint.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), dptr, dptr]
    if hasattr(L, "kh_comm_create"):
        L.kh_comm_unique_id.argtypes = [bptr]
        L.kh_comm_create.argtypes = [i32, i32, i32, bptr, C.POINTER(vp)]
        L.kh_comm_destroy.argtypes = [vp]
        L.kh_comm_destroy.restype = None
        L.kh_comm_rank.argtypes = [vp]
        L.kh_comm_world.argtypes = [vp]
        L.kh_comm_device.argtypes = [vp]
        L.kh_comm_allreduce_sum_f64.argtypes = [vp, vp, C.c_int64, vp]
        L.kh_comm_allgather_f64.argtypes = [vp, vp, vp, C.c_int64, vp]
        L.kh_spa_set_comm.argtypes = [vp, vp]
        L.kh_device_malloc.argtypes = [i32, C.c_int64, C.POINTER(vp)]
        L.kh_device_free.argtypes = [vp]
        L.kh_device_free.restype = None
        L.kh_device_upload.argtypes = [vp, vp, C.c_int64]
        L.kh_device_upload_on.argtypes = [vp, vp, C.c_int64, vp]
        L.kh_device_download.argtypes = [vp, vp, C.c_int64]
        L.kh_selftest_lds_attr.argtypes = []
        L.kh_selftest_lds_attr.restype = C.c_int
    if hasattr(L, "kh_graph_create"):
        L.kh_graph_create.argtypes = [i32, C.POINTER(vp)]
        L.kh_graph_destroy.argtypes = [vp]
        L.kh_graph_destroy.restype = None
        L.kh_graph_set.argtypes = [vp, i32, dptr, iptr, iptr]
        L.kh_graph_set_positions.argtypes = [vp, i32, dptr]
        L.kh_graph_find_loop_candidates.argtypes = [vp, i32, iptr, dbl, i32, iptr, iptr, i32, C.POINTER(i32)]
        L.kh_graph_last_kernel_ms.argtypes = [vp]
        L.kh_graph_find_near_chains.argtypes = [vp, i32, dbl, iptr, i32, C.POINTER(i32)]
        L.kh_graph_closest_scan_to_pose.argtypes = [vp, iptr, i32, dptr, C.POINTER(i32)]
        L.kh_weighted_mean.argtypes = [i32, dptr, dptr, dptr]
        L.kh_graph_last_kernel_ms.restype = dbl
    if hasattr(L, "kh_mapper_create"):
        L.kh_graph_find_loop_candidates_from.argtypes = [vp, i32, iptr, vp, dbl, i32, iptr, iptr, i32, C.POINTER(i32)]
        L.kh_mapper_params_default.argtypes = [C.POINTER(KhMapperParams)]
        L.kh_mapper_params_default.restype = None
        L.kh_mapper_create.argtypes = [C.POINTER(KhMapperParams), C.POINTER(KhLaser), i32, i32, C.POINTER(vp)]
        L.kh_mapper_get_adjacency.argtypes = [vp, i32, iptr, i32, C.POINTER(i32)]
        L.kh_mapper_set_node_score.argtypes = [vp, i32, dbl]
        L.kh_mapper_create_on_devices.argtypes = [C.POINTER(KhMapperParams), C.POINTER(KhLaser), iptr, i32, i32, C.POINTER(vp)]
        L.kh_mapper_destroy.argtypes = [vp]
        L.kh_mapper_destroy.restype = None
        L.kh_mapper_process.argtypes = [vp, dptr, dptr, dbl, C.POINTER(i32), dptr, dptr]
        L.kh_mapper_num_scans.argtypes = [vp]
        L.kh_mapper_num_edges.argtypes = [vp]
        L.kh_mapper_num_edges.restype = C.c_int64
        L.kh_mapper_get_poses.argtypes = [vp, dptr]
        L.kh_mapper_get_scan.argtypes = [vp, i32, C.POINTER(KhScan), C.POINTER(KhScanBox)]
        L.kh_mapper_get_stats.argtypes = [vp, C.POINTER(KhMapperStats)]
        L.kh_mapper_solver.argtypes = [vp]
        L.kh_mapper_solver.restype = vp
        L.kh_mapper_set_log.argtypes = [vp, C.c_char_p]
        L.kh_mapper_remove_node.argtypes = [vp, i32]
        L.kh_mapper_set_lifelong.argtypes = [vp, C.POINTER(KhDecayParams)]
        L.kh_mapper_num_alive.argtypes = [vp]
        L.kh_mapper_get_alive.argtypes = [vp, iptr]
        L.kh_graph_set_scan_limit.argtypes = [vp, i32]
        L.kh_graph_append_scan.argtypes = [vp, dptr]
        L.kh_graph_add_edge.argtypes = [vp, i32, i32]
        L.kh_graph_set_position.argtypes = [vp, i32, dptr]
        L.kh_graph_find_near_linked.argtypes = [vp, i32, dbl, iptr, i32, C.POINTER(i32)]
    if hasattr(L, "kh_lifelong_scores"):
        L.kh_decay_params_default.argtypes = [C.POINTER(KhDecayParams)]
        L.kh_decay_params_default.restype = None
        L.kh_lifelong_scores.argtypes = [i32, C.POINTER(KhScanBox), i32, C.POINTER(KhScanBox), C.POINTER(KhDecayParams),
                                         vp, vp, vp, vp, vp]
    if hasattr(L, "kh_occupancy_create"):
        L.kh_occupancy_compute_dimensions.argtypes = [i32, C.POINTER(KhScan), dbl, dbl, dbl, C.POINTER(i32), C.POINTER(i32), dptr]
        L.kh_occupancy_create.argtypes = [i32, i32, dbl, dbl, dbl, i32, C.POINTER(vp)]
        L.kh_occupancy_destroy.argtypes = [vp]
        L.kh_occupancy_destroy.restype = None
        L.kh_occupancy_clear.argtypes = [vp]
        L.kh_occupancy_add_scans.argtypes = [vp, i32, C.POINTER(KhScan), dbl, dbl, dbl]
        L.kh_occupancy_update.argtypes = [vp, C.c_uint32, dbl]
        L.kh_occupancy_read.argtypes = [vp, vp, vp, vp]
        L.kh_occupancy_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(dbl), C.POINTER(C.c_int64)]
    _lib = L
    return L


def check(code, where):
    if code != KH_OK:
        raise KartoHipError(code, where)
