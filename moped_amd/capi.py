"""ctypes binding of libmoped_hip.so (include/moped_hip.h).

This is plumbing for tests/ and bench.py: it adds no computation of its own.
There is no CPU fallback -- if the HIP library is missing or no gfx950 device is
present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MH_LIB_PATH") or os.path.join(_HERE, "libmoped_hip.so")  # override: A/B kernel builds

MH_OK = 0


class MhError(RuntimeError):
    pass


class mh_corr(C.Structure):
    _fields_ = [("u", C.c_float), ("v", C.c_float), ("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class mh_cam(C.Structure):
    _fields_ = [("K", C.c_float * 4), ("cam", C.c_float * 7)]


class mh_pose_params(C.Structure):
    _fields_ = [("n_hypotheses", C.c_int), ("max_objects_per_cluster", C.c_int),
                ("n_pts_align", C.c_int), ("min_n_pts_object", C.c_int),
                ("error_threshold", C.c_float), ("lm_iters_l2", C.c_int), ("lm_iters_l4", C.c_int)]


class mh_pose_out(C.Structure):
    _fields_ = [("pose", C.c_float * 7), ("cluster", C.c_int32), ("n_inliers", C.c_int32),
                ("err", C.c_float)]


class mh_frame_params(C.Structure):
    _fields_ = [("ratio", C.c_float), ("ms_radius", C.c_float), ("ms_merge", C.c_float),
                ("ms_min_pts", C.c_int), ("ms_max_iter", C.c_int), ("pose1", mh_pose_params),
                ("f1_min_points", C.c_int), ("f1_feature_distance", C.c_float),
                ("f1_min_score", C.c_float), ("pose2", mh_pose_params),
                ("f2_min_points", C.c_int), ("f2_feature_distance", C.c_float),
                ("f2_min_score", C.c_float), ("run_stage2", C.c_int)]


class mh_object(C.Structure):
    _fields_ = [("model", C.c_int32), ("pose", C.c_float * 7), ("score", C.c_float),
                ("n_points", C.c_int32)]


class mh_times(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("match_ms", "group_ms", "cluster_ms", "pose1_ms",
                                         "filter1_ms", "pose2_ms", "filter2_ms", "total_ms")]


CORR_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
OBJECT_DTYPE = np.dtype([("model", "<i4"), ("pose", "<f4", (7,)), ("score", "<f4"), ("n_points", "<i4")])
STEP_OBJECT_DTYPE = np.dtype([("model", "<i4"), ("pose", "<f4", (7,))])   # mh_step_object
DEPTH_DTYPE = np.dtype([("wx", "<f4"), ("wy", "<f4"), ("wz", "<f4"), ("w", "<f4")])
DEPTH_BACKPROJECTION, DEPTH_REPROJECTION = 1, 2
POSE_OUT_DTYPE = np.dtype([("pose", "<f4", (7,)), ("cluster", "<i4"), ("n_inliers", "<i4"), ("err", "<f4")])

# every symbol include/moped_hip.h declares
MAX_BATCH = 32   # MH_MAX_BATCH


class mh_linkage_params(C.Structure):
    _fields_ = [("cutoff", C.c_float), ("min_pts", C.c_int32), ("use3d_filter", C.c_int32), ("sigma2d", C.c_float),
                ("sigma3d", C.c_float), ("linkage_type", C.c_int32)]

    def __init__(self, cutoff=0.1, min_pts=7, use3d_filter=2, sigma2d=-1.0, sigma3d=-1.0, linkage_type=1):
        # (average linkage unless said otherwise: a positional call with the five older fields must not mean "minimum")
        super().__init__(cutoff, min_pts, use3d_filter, sigma2d, sigma3d, linkage_type)


def default_linkage_params():
    """CLUSTER_LINKAGE_CPU( 0.1, 7, 2, 1, 0.0, 1, -1, -1 ) (moped3d/libmoped/src/config.hpp:45)."""
    return mh_linkage_params(0.1, 7, 2, -1.0, -1.0, 1)


class mh_depth_rules(C.Structure):
    _fields_ = [("patch_size", C.c_int32), ("feature_density", C.c_float), ("match_density", C.c_float),
                ("ratio_table", C.c_void_p), ("n_models", C.c_int32), ("maximum_depth", C.c_float),
                ("default_depth", C.c_float), ("cauchy_scale", C.c_float)]


EXPORTS = [
    "mh_create", "mh_destroy", "mh_last_error", "mh_set_stream", "mh_synchronize", "mh_reserve",
    "mh_db_upload", "mh_db_size", "mh_normalize", "mh_match", "mh_normalize_match", "mh_match_local_dev",
    "mh_match_merge_dev", "mh_normalize_dev", "mh_meanshift", "mh_meanshift_batch", "mh_pose_ransac", "mh_pose_ransac_depth",
    "mh_frame_set_depth", "mh_project_test",
    "mh_filter", "mh_frame_default_params", "mh_frame_enqueue", "mh_frame_set_depth_image", "mh_frame_enqueue_match_local",
    "mh_frame_enqueue_rest", "mh_frame_fetch", "mh_frame_result_dev", "mh_enable_timing", "mh_timing",
    "mh_frame_set_cluster_linkage", "mh_cluster_linkage", "mh_frame_set_depth_image_host",
    "mh_depth_fill", "mh_depth_fill_status", "mh_depth_fill_host",
    "mh_frame_enqueue_rest_batch", "mh_frame_enqueue_rest_frames", "mh_frame_fetch_slot", "mh_frame_result_copy_slots_dev",
    "mh_frame_set_depth_rules", "mh_frame_fetch_matches", "mh_frame_enqueue_rest_strided", "mh_frame_result_copy_dev",
    "mh_sift_extract", "mh_sift_extract_dev", "mh_frame_enqueue_image", "mh_frame_enqueue_image_batch", "mh_frame_features_dev", "mh_frame_keypoints",
    "mh_models_create", "mh_models_destroy", "mh_models_last_error", "mh_models_add_xml",
    "mh_models_add_xml_buffer", "mh_models_count", "mh_models_rows", "mh_models_name", "mh_models_range",
    "mh_models_desc", "mh_models_xyz", "mh_models_save", "mh_models_load", "mh_db_upload_models",
    "mh_db_upload_raw", "mh_db_share", "mh_match_stats", "mh_match_set_mode", "mh_match_launches", "mh_pose_set_split", "mh_frame_fetch_match_points", "mh_screen_margin", "mh_frame_counters", "mh_match_timing", "mh_frame_set_images", "mh_filter_images",
    "mh_pose_ransac_images",
    "mh_comm_unique_id", "mh_comm_create", "mh_comm_create_all", "mh_comm_create_host", "mh_comm_destroy", "mh_comm_info",
    "mh_frame_enqueue_sharded", "mh_frame_enqueue_sharded_batch", "mh_frame_enqueue_sharded_all",
    "mh_frame_previous_objects", "mh_frame_gather_objects", "mh_frame_enqueue_batch", "mh_frame_set_depth_image_batch",
    "mh_pose_kernel_info", "mh_db_upload_blocks", "mh_frame_fetch_matches_slot",
    "mh_screen_values", "mh_screen_record_value", "mh_screen_record_bounds", "mh_reserve_batch", "mh_frame_run_host",
    "mh_frame_block_stride", "mh_frame_fetch_batch_async", "mh_frame_fetch_previous_async", "mh_frame_fetch_wait",
    "mh_frame_fetch_query", "mh_host_alloc", "mh_host_free", "mh_frame_run_host_begin", "mh_frame_wait_descriptors",
    "mh_step_match", "mh_step_match_fetch", "mh_step_cluster", "mh_step_pose", "mh_step_filter",
    "mh_set_linkage_scratch_limit",
]
COMM_ID_BYTES = 128      # MH_COMM_ID_BYTES
EX2_OBJECTS = 62         # MH_EX2_OBJECTS
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)   # mh_allgather_fn

_lib = None


def load():
    """dlopen the HIP library and declare prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # When PyTorch-ROCm shares the process (bench.py, multi-GPU tests) its bundled
    # HIP/HSA runtime must be the one in the process: two HSA runtimes cannot both
    # own the GPU.  Importing torch first makes libmoped_hip.so bind to that copy
    # (same SONAME libamdhip64.so.7); without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch absent: plain ROCm runtime
        pass
    if not os.path.exists(LIB_PATH):
        raise MhError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    L.mh_create.argtypes = [i32, C.POINTER(vp)]
    L.mh_destroy.argtypes = [vp]
    L.mh_destroy.restype = None
    L.mh_last_error.argtypes = [vp]
    L.mh_last_error.restype = C.c_char_p
    L.mh_set_stream.argtypes = [vp, vp]
    L.mh_synchronize.argtypes = [vp]
    L.mh_reserve.argtypes = [vp, i32, i32, i32]
    L.mh_reserve_batch.argtypes = [vp, i32, i32, i32, i32]
    if hasattr(L, "mh_lane_create"):   # experiment builds only (csrc/api.hip)
        L.mh_lane_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
        L.mh_lane_destroy.argtypes = [vp]
        L.mh_set_lane.argtypes = [vp, vp]
    L.mh_db_upload.argtypes = [vp, vp, vp, vp, i32, i32, C.c_int32]
    L.mh_db_size.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.mh_normalize.argtypes = [vp, vp, i32]
    L.mh_match.argtypes = [vp, vp, i32, f32, vp, vp, vp, vp]
    L.mh_normalize_match.argtypes = [vp, vp, i32, f32, vp, vp, vp, vp]
    L.mh_match_local_dev.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.mh_match_merge_dev.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.mh_normalize_dev.argtypes = [vp, vp, vp, i32]
    L.mh_meanshift.argtypes = [vp, vp, i32, i32, f32, f32, i32, i32, vp, vp, C.POINTER(C.c_int32)]
    L.mh_meanshift_batch.argtypes = [vp, vp, vp, i32, i32, f32, f32, i32, i32, vp, vp, vp]
    L.mh_pose_ransac.argtypes = [vp, vp, vp, i32, C.POINTER(mh_cam), C.POINTER(mh_pose_params),
                                 C.c_uint64, vp, C.POINTER(C.c_int32)]
    L.mh_pose_ransac_depth.argtypes = [vp, vp, vp, vp, i32, C.POINTER(mh_cam), C.POINTER(mh_pose_params), i32, f32,
                                       C.c_uint64, vp, C.POINTER(C.c_int32)]
    L.mh_frame_set_depth.argtypes = [vp, vp, i32, f32]
    L.mh_project_test.argtypes = [vp, vp, vp, i32, C.POINTER(mh_cam), f32, vp, vp, C.POINTER(C.c_int32)]
    L.mh_filter.argtypes = [vp, vp, vp, i32, vp, vp, i32, C.POINTER(mh_cam), i32, f32, f32,
                            vp, vp, vp, vp, vp, C.POINTER(C.c_int32)]
    L.mh_frame_default_params.argtypes = [C.POINTER(mh_frame_params)]
    L.mh_frame_default_params.restype = None
    L.mh_frame_enqueue.argtypes = [vp, vp, vp, i32, C.POINTER(mh_cam), C.POINTER(mh_frame_params), C.c_uint64]
    L.mh_frame_run_host.argtypes = [vp, vp, vp, vp, i32, C.POINTER(mh_cam), i32, C.POINTER(mh_frame_params), C.c_uint64, i32,
                                    vp, i32, C.POINTER(C.c_int32), vp]
    L.mh_frame_run_host_begin.argtypes = [vp, vp, vp, vp, i32, C.POINTER(mh_cam), i32, C.POINTER(mh_frame_params), C.c_uint64, i32]
    L.mh_sift_extract.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, i32, C.POINTER(C.c_int32)]
    L.mh_sift_extract_dev.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, i32, vp]
    L.mh_frame_enqueue_image.argtypes = [vp, vp, i32, i32, i32, i32, C.POINTER(mh_cam), C.POINTER(mh_frame_params),
                                         C.c_uint64]
    L.mh_frame_enqueue_image_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.POINTER(mh_cam), C.POINTER(mh_frame_params), vp]
    L.mh_frame_set_depth_image_host.argtypes = [vp, vp, vp, i32, i32, i32, f32, f32]
    L.mh_depth_fill.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
    L.mh_depth_fill_status.argtypes = [vp]
    L.mh_depth_fill_host.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
    L.mh_frame_set_cluster_linkage.argtypes = [vp, C.POINTER(mh_linkage_params)]
    L.mh_cluster_linkage.argtypes = [vp, vp, vp, vp, i32, C.POINTER(mh_linkage_params), vp, vp, vp]
    L.mh_frame_set_depth_rules.argtypes = [vp, C.POINTER(mh_depth_rules), vp]
    L.mh_frame_enqueue_rest_strided.argtypes = [vp, vp, i32, vp, i32, i32, C.POINTER(mh_cam),
                                                C.POINTER(mh_frame_params), C.c_uint64]
    L.mh_frame_result_copy_dev.argtypes = [vp, vp, i32]
    L.mh_frame_enqueue_rest_batch.argtypes = [vp, vp, i32, vp, i32, i32, i32, i32, C.POINTER(mh_cam),
                                              C.POINTER(mh_frame_params), C.c_uint64]
    L.mh_frame_fetch_slot.argtypes = [vp, i32, vp, i32, C.POINTER(C.c_int32), vp]
    L.mh_frame_result_copy_slots_dev.argtypes = [vp, vp, i32, i32]
    L.mh_frame_fetch_matches.argtypes = [vp, vp, vp, i32, C.POINTER(C.c_int32)]
    L.mh_frame_features_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.mh_frame_keypoints.argtypes = [vp, C.POINTER(C.c_int32)]
    L.mh_models_create.argtypes = [C.POINTER(vp), C.c_char_p]
    L.mh_models_destroy.argtypes = [vp]
    L.mh_models_destroy.restype = None
    L.mh_models_last_error.argtypes = [vp]
    L.mh_models_last_error.restype = C.c_char_p
    L.mh_models_add_xml.argtypes = [vp, C.c_char_p]
    L.mh_models_add_xml_buffer.argtypes = [vp, C.c_char_p, C.c_int64]
    L.mh_models_count.argtypes = [vp]
    L.mh_models_rows.argtypes = [vp]
    L.mh_models_rows.restype = C.c_int64
    L.mh_models_name.argtypes = [vp, i32]
    L.mh_models_name.restype = C.c_char_p
    L.mh_models_range.argtypes = [vp, i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp]
    L.mh_models_desc.argtypes = [vp]
    L.mh_models_desc.restype = vp
    L.mh_models_xyz.argtypes = [vp]
    L.mh_models_xyz.restype = vp
    L.mh_models_save.argtypes = [vp, C.c_char_p]
    L.mh_models_load.argtypes = [C.POINTER(vp), C.c_char_p]
    L.mh_db_upload_models.argtypes = [vp, vp, i32, i32]
    L.mh_db_upload_raw.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32]
    L.mh_frame_set_depth_image.argtypes = [vp, vp, vp, i32, i32, i32, f32, f32]
    L.mh_frame_enqueue_match_local.argtypes = [vp, vp, i32, vp]
    L.mh_frame_enqueue_rest.argtypes = [vp, vp, i32, vp, i32, C.POINTER(mh_cam),
                                        C.POINTER(mh_frame_params), C.c_uint64]
    L.mh_frame_fetch.argtypes = [vp, vp, i32, C.POINTER(C.c_int32), vp]
    L.mh_frame_result_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.mh_db_share.argtypes = [vp, vp]
    L.mh_match_stats.argtypes = [vp, i32, vp, i32]
    L.mh_match_set_mode.argtypes = [vp, i32]
    L.mh_match_launches.argtypes = [vp, vp]
    L.mh_pose_set_split.argtypes = [vp, i32]
    L.mh_frame_counters.argtypes = [vp, vp]
    L.mh_frame_set_images.argtypes = [vp, vp, vp, i32]
    L.mh_filter_images.argtypes = [vp, vp, vp, vp, i32, vp, vp, i32, vp, i32, i32, f32, f32,
                                   vp, vp, vp, vp, vp, C.POINTER(C.c_int32)]
    L.mh_pose_ransac_images.argtypes = [vp, vp, vp, vp, i32, vp, i32, C.POINTER(mh_pose_params),
                                        C.c_uint64, vp, C.POINTER(C.c_int32)]
    L.mh_match_timing.argtypes = [vp, vp]
    L.mh_screen_margin.argtypes = [f32, f32]
    L.mh_screen_margin.restype = f32
    L.mh_enable_timing.argtypes = [vp, i32]
    L.mh_timing.argtypes = [vp, C.POINTER(mh_times)]
    L.mh_comm_unique_id.argtypes = [vp]
    L.mh_comm_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    L.mh_comm_create_all.argtypes = [C.POINTER(vp), i32, C.POINTER(vp)]
    L.mh_comm_create_host.argtypes = [vp, i32, i32, ALLGATHER_FN, vp, C.POINTER(vp)]
    L.mh_comm_destroy.argtypes = [vp]
    L.mh_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.mh_frame_enqueue_sharded.argtypes = [vp, vp, vp, vp, i32, C.POINTER(mh_cam), C.POINTER(mh_frame_params), C.c_uint64]
    L.mh_frame_enqueue_sharded_batch.argtypes = [vp, vp, vp, vp, i32, i32, C.POINTER(mh_cam),
                                                 C.POINTER(mh_frame_params), C.POINTER(C.c_uint64)]
    L.mh_frame_enqueue_sharded_all.argtypes = [C.POINTER(vp), C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(vp), i32, i32,
                                               C.POINTER(mh_cam), C.POINTER(mh_frame_params), C.POINTER(C.c_uint64)]
    L.mh_frame_enqueue_batch.argtypes = [vp, vp, vp, i32, i32, C.POINTER(mh_cam), C.POINTER(mh_frame_params),
                                         C.POINTER(C.c_uint64)]
    L.mh_frame_set_depth_image_batch.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i32, i32, i32, i32, f32, f32]
    L.mh_frame_previous_objects.argtypes = [vp, i32, vp, i32, C.POINTER(C.c_int32)]
    L.mh_frame_gather_objects.argtypes = [vp, vp, i32, vp, i32, C.POINTER(C.c_int32)]
    L.mh_pose_kernel_info.argtypes = [vp, i32, vp]
    L.mh_frame_fetch_matches_slot.argtypes = [vp, i32, vp, vp, i32, C.POINTER(C.c_int32)]
    L.mh_frame_enqueue_rest_frames.argtypes = [vp, vp, i32, vp, i32, i32, i32, i32, C.POINTER(mh_cam),
                                               C.POINTER(mh_frame_params), C.POINTER(C.c_uint64)]
    L.mh_screen_values.argtypes = [vp, vp, i32, i32, vp, C.POINTER(f32), C.POINTER(f32), i32]
    L.mh_screen_record_value.argtypes = [f32, f32]
    L.mh_screen_record_value.restype = C.c_uint16
    L.mh_screen_record_bounds.argtypes = [C.c_uint16, C.c_uint32, f32, f32, i32, f32, C.POINTER(f32), C.POINTER(f32)]
    L.mh_screen_record_bounds.restype = None
    L.mh_db_upload_blocks.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, i32, i32]
    L.mh_frame_block_stride.argtypes = [i32]
    L.mh_frame_block_stride.restype = C.c_size_t
    L.mh_frame_fetch_batch_async.argtypes = [vp, i32, i32, vp, C.c_uint32]
    L.mh_frame_fetch_previous_async.argtypes = [vp, i32, vp, C.c_uint32]
    L.mh_frame_fetch_wait.argtypes = [vp, C.POINTER(C.c_int32)]
    L.mh_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.mh_host_free.argtypes = [vp, vp]
    L.mh_frame_wait_descriptors.argtypes = [vp]
    if hasattr(L, "mh_step_match"):   # (absent only in an older build named by MH_LIB_PATH for an A/B run)
        L.mh_set_linkage_scratch_limit.argtypes = [vp, C.c_size_t]
        L.mh_step_match.argtypes = [vp, vp, vp, i32, C.POINTER(mh_cam), f32, i32]
        L.mh_step_match_fetch.argtypes = [vp, vp, vp, vp, i32, C.POINTER(C.c_int32)]
        L.mh_step_cluster.argtypes = [vp, f32, f32, i32, i32, vp, vp, vp, i32, i32, C.POINTER(C.c_int32)]
        L.mh_step_pose.argtypes = [vp, i32, C.POINTER(mh_pose_params), C.c_uint64, vp, i32, C.POINTER(C.c_int32)]
        L.mh_step_filter.argtypes = [vp, i32, i32, f32, f32, i32, vp, vp, vp, vp, vp, i32, C.POINTER(C.c_int32)]
    L.mh_frame_fetch_query.argtypes = [vp]
    _lib = L
    return L


def make_cam(K, cam) -> mh_cam:
    c = mh_cam()
    c.K[:] = [float(x) for x in K]
    c.cam[:] = [float(x) for x in cam]
    return c


def make_cams(Ks, cams):
    """Array of mh_cam for a frame with several images: Ks [n,4], cams [n,7]."""
    arr = (mh_cam * len(Ks))()
    for i, (K, c) in enumerate(zip(Ks, cams)):
        arr[i] = make_cam(K, c)
    return arr


def make_pose_params(n_hypotheses=1024, max_objects_per_cluster=4, n_pts_align=5,
                     min_n_pts_object=6, error_threshold=10.0, lm_iters_l2=2, lm_iters_l4=10):
    return mh_pose_params(n_hypotheses, max_objects_per_cluster, n_pts_align, min_n_pts_object,
                          error_threshold, lm_iters_l2, lm_iters_l4)


def default_frame_params() -> mh_frame_params:
    p = mh_frame_params()
    load().mh_frame_default_params(C.byref(p))
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pack_depth(world, wgt) -> np.ndarray:
    n = len(world)
    d = np.zeros(n, DEPTH_DTYPE)
    if n:
        world = np.asarray(world, np.float32)
        d["wx"], d["wy"], d["wz"] = world[:, 0], world[:, 1], world[:, 2]
        d["w"] = np.asarray(wgt, np.float32)
    return d


def pack_corr(uv, xyz) -> np.ndarray:
    n = len(uv)
    c = np.zeros(n, CORR_DTYPE)
    if n:
        uv = np.asarray(uv, np.float32)
        xyz = np.asarray(xyz, np.float32)
        c["u"], c["v"] = uv[:, 0], uv[:, 1]
        c["x"], c["y"], c["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    return c


def screen_record_bounds(value_bits, row0, tau, spread, N, dmax):
    """mh_screen_record_bounds (host arithmetic): (lo, hi) of the largest screen value among a record's rows."""
    lo, hi = C.c_float(0), C.c_float(0)
    load().mh_screen_record_bounds(int(value_bits), int(row0), float(tau), float(spread), int(N), float(dmax),
                                   C.byref(lo), C.byref(hi))
    return lo.value, hi.value


FRAME_HEAD_DTYPE = np.dtype([("n_objects", "<i4"), ("flags", "<i4"), ("counts", "<i4", (4,)), ("tag", "<u4"), ("frame", "<i4")])


def frame_block_dtype(max_objects: int) -> np.dtype:
    """One record of a delivery block (mh_frame_fetch_batch_async): mh_frame_head + mh_object[max_objects]."""
    return np.dtype([("head", FRAME_HEAD_DTYPE), ("objects", OBJECT_DTYPE, (max_objects,))])


def frame_block_bytes(B: int, max_objects: int) -> int:
    return B * (FRAME_HEAD_DTYPE.itemsize + OBJECT_DTYPE.itemsize * max_objects)


def comm_unique_id() -> bytes:
    """mh_comm_unique_id: rank 0 makes it, the launcher hands it to the other ranks."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    if load().mh_comm_unique_id(buf) != MH_OK:
        raise MhError("mh_comm_unique_id failed: RCCL not loadable?")
    return buf.raw


class Lane:
    """mh_lane: streams for the chip-filling MATCH passes of all contexts of a device (CU-masked or low priority)."""

    def __init__(self, device=0, n_streams=4, reserve_cus_per_xcd=2, low_priority=False):
        self.L = load()
        if not hasattr(self.L, "mh_lane_create"):
            raise MhError("lanes exist only in experiment builds of the library (make EXTRA=-DMH_EXPERIMENTS; MH_LIB_PATH)")
        h = C.c_void_p()
        rc = self.L.mh_lane_create(device, n_streams, reserve_cus_per_xcd, int(low_priority), C.byref(h))
        if rc != MH_OK:
            raise MhError(f"mh_lane_create -> {rc}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.mh_lane_destroy(self.h)
            self.h = None


class Comm:
    """mh_comm of one rank: RCCL (`create`), or a host transport (`create_host`: fn(send: bytes) -> bytes of all
    ranks in rank order -- ranks that share a device, test rigs)."""

    def __init__(self, handle, keep=None):
        self.L = load()
        self.h = handle
        self._keep = keep   # the ctypes callback must outlive the communicator

    @classmethod
    def create(cls, ctx: "Context", unique_id: bytes, rank: int, world: int) -> "Comm":
        h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, COMM_ID_BYTES)
        ctx._ck(ctx.L.mh_comm_create(ctx.h, buf, rank, world, C.byref(h)), "mh_comm_create")
        return cls(h)

    @classmethod
    def create_host(cls, ctx: "Context", rank: int, world: int, allgather) -> "Comm":
        def _cb(_user, send, recv, nbytes):
            try:
                out = allgather(C.string_at(send, nbytes))
                if len(out) != nbytes * world:
                    return 1
                C.memmove(recv, out, len(out))
                return 0
            except Exception:   # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        fn = ALLGATHER_FN(_cb)
        h = C.c_void_p()
        ctx._ck(ctx.L.mh_comm_create_host(ctx.h, rank, world, fn, None, C.byref(h)), "mh_comm_create_host")
        return cls(h, keep=fn)

    @classmethod
    def create_all(cls, ctxs) -> "list[Comm]":
        n = len(ctxs)
        hs = (C.c_void_p * n)(*[c.h for c in ctxs])
        out = (C.c_void_p * n)()
        ctxs[0]._ck(ctxs[0].L.mh_comm_create_all(hs, n, out), "mh_comm_create_all")
        return [cls(C.c_void_p(out[i])) for i in range(n)]

    def info(self):
        r, w, k = C.c_int(0), C.c_int(0), C.c_int(0)
        self.L.mh_comm_info(self.h, C.byref(r), C.byref(w), C.byref(k))
        return r.value, w.value, bool(k.value)

    def close(self):
        if getattr(self, "h", None):
            self.L.mh_comm_destroy(self.h)
            self.h = None


def frame_enqueue_sharded_all(ctxs, comms, q_desc_ptrs, q_uv_ptrs, Q, B, K, cam, params, seeds):
    """mh_frame_enqueue_sharded_all: one host thread, one context + communicator per device."""
    n = len(ctxs)
    c = make_cam(K, cam)
    sd = (C.c_uint64 * B)(*[int(x) for x in seeds])
    rc = ctxs[0].L.mh_frame_enqueue_sharded_all((C.c_void_p * n)(*[x.h for x in ctxs]), (C.c_void_p * n)(*[x.h for x in comms]),
                                                n, (C.c_void_p * n)(*q_desc_ptrs), (C.c_void_p * n)(*q_uv_ptrs), Q, B,
                                                C.byref(c), C.byref(params), sd)
    if rc != MH_OK:
        msgs = [x.L.mh_last_error(x.h).decode() for x in ctxs]
        raise MhError(f"mh_frame_enqueue_sharded_all -> {rc}: {msgs}")


class Context:
    """One mh_ctx.  Host-array methods mirror the per-step C entry points."""

    def __init__(self, device: int = 0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.mh_create(device, C.byref(h))
        if rc != MH_OK:
            raise MhError(f"mh_create(device={device}) failed with {rc}: no gfx950 device?")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.mh_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, rc, what):
        if rc != MH_OK:
            raise MhError(f"{what} -> {rc}: {self.L.mh_last_error(self.h).decode()}")

    # ---- context ----
    def set_stream(self, stream_ptr):
        self._ck(self.L.mh_set_stream(self.h, C.c_void_p(stream_ptr)), "mh_set_stream")

    def synchronize(self):
        self._ck(self.L.mh_synchronize(self.h), "mh_synchronize")

    def set_lane(self, lane: "Lane | None"):
        self._ck(self.L.mh_set_lane(self.h, lane.h if lane is not None else None), "mh_set_lane")

    def reserve(self, max_queries, max_clusters=1024, max_objects=4096):
        self._ck(self.L.mh_reserve(self.h, max_queries, max_clusters, max_objects), "mh_reserve")

    def reserve_batch(self, queries_per_frame, frames, max_clusters=1024, max_objects=4096):
        self._ck(self.L.mh_reserve_batch(self.h, queries_per_frame, frames, max_clusters, max_objects), "mh_reserve_batch")

    def enable_timing(self, on=True):
        self._ck(self.L.mh_enable_timing(self.h, int(on)), "mh_enable_timing")

    def timing(self) -> dict:
        t = mh_times()
        self._ck(self.L.mh_timing(self.h, C.byref(t)), "mh_timing")
        return {n: getattr(t, n) for n, _ in mh_times._fields_}

    # ---- DB / MATCH ----
    def db_upload(self, desc, model_of, xyz, n_models, index_base=0):
        desc = np.ascontiguousarray(desc, np.float32)
        model_of = np.ascontiguousarray(model_of, np.int32)
        xyz = np.ascontiguousarray(xyz, np.float32)
        self._ck(self.L.mh_db_upload(self.h, _ptr(desc), _ptr(model_of), _ptr(xyz), desc.shape[0],
                                     n_models, index_base), "mh_db_upload")

    def db_upload_blocks(self, desc, model_of, xyz, n_models, block_global_row, block_rows, normalize=False):
        """A shard whose rows are several runs of global rows (round-robin model assignment): mh_db_upload_blocks."""
        desc = np.ascontiguousarray(desc, np.float32)
        model_of = np.ascontiguousarray(model_of, np.int32)
        xyz = np.ascontiguousarray(xyz, np.float32)
        g = np.ascontiguousarray(block_global_row, np.int32)
        r = np.ascontiguousarray(block_rows, np.int32)
        self._ck(self.L.mh_db_upload_blocks(self.h, _ptr(desc), _ptr(model_of), _ptr(xyz), desc.shape[0], n_models,
                                            _ptr(g), _ptr(r), len(g), int(normalize)), "mh_db_upload_blocks")

    def db_share(self, src: "Context"):
        """Use the database `src` holds (no copy; one store per GPU for all frames in flight)."""
        self._ck(self.L.mh_db_share(self.h, src.h), "mh_db_share")

    def frame_counters(self) -> dict:
        """Device-side counters of the last frame on this context (see mh_frame_counters)."""
        c = np.zeros(8, np.int32)
        self._ck(self.L.mh_frame_counters(self.h, _ptr(c)), "mh_frame_counters")
        names = ("matches", "clusters", "r2", "r3", "r4", "pose_tasks", "error", "hypotheses")
        return dict(zip(names, (int(v) for v in c)))

    def pose_kernel_info(self, kind=0):
        """mh_pose_kernel_info: registers / LDS / threads / residency of the RANSAC kernel on this device."""
        o = np.zeros(8, np.int32)
        self._ck(self.L.mh_pose_kernel_info(self.h, int(kind), _ptr(o)), "mh_pose_kernel_info")
        waves = int(o[5])
        return {"vgprs": int(o[0]), "lds_bytes": int(o[1]), "threads": int(o[2]), "workgroups_per_cu": int(o[3]),
                "scratch_bytes": int(o[4]), "waves_per_workgroup": waves,
                "waves_per_simd": int(o[3]) * waves / 4.0}

    def match_timing(self) -> dict:
        """Per-kernel GPU times (ms) of the last two-stage MATCH (after enable_timing)."""
        t = np.zeros(5, np.float32)
        self._ck(self.L.mh_match_timing(self.h, _ptr(t)), "mh_match_timing")
        return dict(zip(("prepare_ms", "pass_a_ms", "thresholds_ms", "pass_b_ms", "pass_c_ms"), (float(v) for v in t)))

    def screen_values(self, qn, n_rows, shape=0):
        """mh_screen_values: [Q][n_rows] screen values as the matrix pipe computes them, + (dmax, spread) of the DB.
        shape: 0 = the MFMA shape the large launches use, 1 = 32x32x16, 2 = 16x16x32."""
        qn = np.ascontiguousarray(qn, np.float32)
        out = np.zeros((qn.shape[0], n_rows), np.float32)
        dmax, spread = C.c_float(0), C.c_float(0)
        self._ck(self.L.mh_screen_values(self.h, _ptr(qn), qn.shape[0], n_rows, _ptr(out), C.byref(dmax), C.byref(spread),
                                         int(shape)), "mh_screen_values")
        return out, dmax.value, spread.value

    def match_set_mode(self, mode: int):
        """-1 auto, 0 exact f32 kernels only, 1 two-stage (f16 screen + exact rescoring) whenever possible,
        2 / 3 the exact VALU / f32 matrix-pipe kernel whatever the query count."""
        self._ck(self.L.mh_match_set_mode(self.h, int(mode)), "mh_match_set_mode")

    def pose_set_split(self, on: bool):
        """POSE of the frame paths as two launches (hypotheses, then one-wavefront refines; the default) or as one."""
        self._ck(self.L.mh_pose_set_split(self.h, int(bool(on))), "mh_pose_set_split")

    def match_kernel_launches(self) -> dict:
        """MATCH launch sequences of this context by the kernel that searched (mh_match_launches)."""
        o = np.zeros(3, np.uint32)
        self._ck(self.L.mh_match_launches(self.h, _ptr(o)), "mh_match_launches")
        return {"valu": int(o[0]), "mfma": int(o[1]), "screen": int(o[2])}

    def match_stats(self, Q=0, reset=False) -> dict:
        """Two-stage MATCH statistics since the last reset + which path Q queries would take."""
        st = np.zeros(4, np.uint32)
        self._ck(self.L.mh_match_stats(self.h, int(Q), _ptr(st), int(reset)), "mh_match_stats")
        return {"candidates": int(st[0]), "brute_force_queries": int(st[1]), "queries": int(st[2]),
                "two_stage": bool(st[3])}

    def normalize(self, desc):
        d = np.ascontiguousarray(desc, np.float32).copy()
        self._ck(self.L.mh_normalize(self.h, _ptr(d), d.shape[0]), "mh_normalize")
        return d

    def match(self, q, ratio=0.8):
        q = np.ascontiguousarray(q, np.float32)
        Q = q.shape[0]
        acc = np.full(Q, -1, np.int32)
        raw = np.full(Q, -1, np.int32)
        d1 = np.zeros(Q, np.float32)
        d2 = np.zeros(Q, np.float32)
        self._ck(self.L.mh_match(self.h, _ptr(q), Q, ratio, _ptr(acc), _ptr(raw), _ptr(d1), _ptr(d2)), "mh_match")
        return acc, raw, d1, d2

    def normalize_match(self, q, ratio=0.8):
        """mh_normalize + mh_match in one call: -> (normalised copy of q, acc, raw, d1, d2)."""
        q = np.ascontiguousarray(q, np.float32).copy()
        Q = q.shape[0]
        acc = np.full(Q, -1, np.int32)
        raw = np.full(Q, -1, np.int32)
        d1 = np.zeros(Q, np.float32)
        d2 = np.zeros(Q, np.float32)
        self._ck(self.L.mh_normalize_match(self.h, _ptr(q), Q, ratio, _ptr(acc), _ptr(raw), _ptr(d1), _ptr(d2)),
                 "mh_normalize_match")
        return q, acc, raw, d1, d2

    # ---- CLUSTER ----
    def meanshift(self, pts, radius=200.0, merge=20.0, min_pts=7, max_iter=100):
        pts = np.ascontiguousarray(pts, np.float32)
        n = pts.shape[0]
        dim = pts.shape[1] if pts.ndim == 2 else 2
        label = np.full(max(n, 1), -1, np.int32)
        order = np.full(max(n, 1), -1, np.int32)
        ncl = C.c_int32(0)
        self._ck(self.L.mh_meanshift(self.h, _ptr(pts), n, dim, radius, merge, min_pts, max_iter,
                                     _ptr(label), _ptr(order), C.byref(ncl)), "mh_meanshift")
        label = label[:n]
        clusters, pos = [], 0
        for c in range(ncl.value):
            sz = int((label == c).sum())
            clusters.append(order[pos:pos + sz].copy())
            pos += sz
        return clusters, label

    def meanshift_batch(self, problems, radius=200.0, merge=20.0, min_pts=7, max_iter=100):
        """problems: list of [n_p, dim] point arrays (same dim) -> list of (clusters, label) like meanshift()."""
        dim = 2
        for p in problems:
            if np.asarray(p).ndim == 2 and np.asarray(p).shape[0]:
                dim = np.asarray(p).shape[1]
        sizes = [int(np.asarray(p).shape[0]) for p in problems]
        off = np.zeros(len(problems) + 1, np.int32)
        off[1:] = np.cumsum(sizes)
        total = int(off[-1])
        pts = (np.concatenate([np.asarray(p, np.float32).reshape(-1, dim) for p in problems])
               if total else np.zeros((0, dim), np.float32))
        pts = np.ascontiguousarray(pts, np.float32)
        label = np.full(max(total, 1), -1, np.int32)
        order = np.full(max(total, 1), -1, np.int32)
        ncl = np.zeros(max(len(problems), 1), np.int32)
        self._ck(self.L.mh_meanshift_batch(self.h, _ptr(pts), _ptr(off), len(problems), dim, radius, merge,
                                           min_pts, max_iter, _ptr(label), _ptr(order), _ptr(ncl)),
                 "mh_meanshift_batch")
        out = []
        for i in range(len(problems)):
            b, e = int(off[i]), int(off[i + 1])
            lab, ordr = label[b:e], order[b:e]
            clusters, pos = [], 0
            for c in range(int(ncl[i])):
                sz = int((lab == c).sum())
                clusters.append(ordr[pos:pos + sz].copy())
                pos += sz
            out.append((clusters, lab.copy()))
        return out

    # ---- POSE ----
    def pose_ransac(self, corr, cluster_off, K, cam, params: mh_pose_params, seed=1):
        corr = np.ascontiguousarray(corr, CORR_DTYPE)
        cluster_off = np.ascontiguousarray(cluster_off, np.int32)
        ncl = len(cluster_off) - 1
        R = max(params.max_objects_per_cluster, 1)
        out = np.zeros(max(ncl * R, 1), POSE_OUT_DTYPE)
        n_out = C.c_int32(0)
        c = make_cam(K, cam)
        self._ck(self.L.mh_pose_ransac(self.h, _ptr(corr), _ptr(cluster_off), ncl, C.byref(c),
                                       C.byref(params), seed, _ptr(out), C.byref(n_out)), "mh_pose_ransac")
        return out[:n_out.value].copy()

    def pose_ransac_images(self, corr, image_of, cluster_off, Ks, cams, params: mh_pose_params, seed=1):
        """pose_ransac with every correspondence in its own image (Ks [n,4], cams [n,7])."""
        corr = np.ascontiguousarray(corr, CORR_DTYPE)
        image_of = np.ascontiguousarray(image_of, np.int32)
        cluster_off = np.ascontiguousarray(cluster_off, np.int32)
        ncl = len(cluster_off) - 1
        R = max(params.max_objects_per_cluster, 1)
        out = np.zeros(max(ncl * R, 1), POSE_OUT_DTYPE)
        n_out = C.c_int32(0)
        arr = make_cams(Ks, cams)
        self._ck(self.L.mh_pose_ransac_images(self.h, _ptr(corr), _ptr(image_of), _ptr(cluster_off), ncl, arr, len(Ks),
                                              C.byref(params), seed, _ptr(out), C.byref(n_out)), "mh_pose_ransac_images")
        return out[:n_out.value].copy()

    def frame_set_images(self, q_image_ptr, Ks=None, cams=None):
        """Frames with several images: q_image_ptr = device int32[Q] (image of every query); 0 / None: one image again."""
        if not q_image_ptr or Ks is None or len(Ks) <= 1:
            self._ck(self.L.mh_frame_set_images(self.h, None, None, 1), "mh_frame_set_images")
            return
        arr = make_cams(Ks, cams)
        self._ck(self.L.mh_frame_set_images(self.h, C.c_void_p(q_image_ptr), arr, len(Ks)), "mh_frame_set_images")

    def pose_ransac_depth(self, corr, depth, cluster_off, K, cam, params: mh_pose_params, kind, alpha=0.5, seed=1):
        corr = np.ascontiguousarray(corr, CORR_DTYPE)
        depth = np.ascontiguousarray(depth, DEPTH_DTYPE)
        cluster_off = np.ascontiguousarray(cluster_off, np.int32)
        ncl = len(cluster_off) - 1
        R = max(params.max_objects_per_cluster, 1)
        out = np.zeros(max(ncl * R, 1), POSE_OUT_DTYPE)
        n_out = C.c_int32(0)
        c = make_cam(K, cam)
        self._ck(self.L.mh_pose_ransac_depth(self.h, _ptr(corr), _ptr(depth), _ptr(cluster_off), ncl, C.byref(c),
                                             C.byref(params), kind, alpha, seed, _ptr(out), C.byref(n_out)),
                 "mh_pose_ransac_depth")
        return out[:n_out.value].copy()

    def frame_set_depth(self, q_depth_ptr, kind, alpha=0.5):
        self._ck(self.L.mh_frame_set_depth(self.h, C.c_void_p(q_depth_ptr) if q_depth_ptr else None, kind, alpha),
                 "mh_frame_set_depth")

    def frame_set_depth_image(self, depth_ptr, fill_ptr, width, height, kind, alpha=0.5, cauchy_scale=0.1):
        self._ck(self.L.mh_frame_set_depth_image(self.h, C.c_void_p(depth_ptr) if depth_ptr else None,
                                                 C.c_void_p(fill_ptr) if fill_ptr else None, width, height, kind,
                                                 alpha, cauchy_scale), "mh_frame_set_depth_image")

    def depth_fill_dev(self, depth_ptr, width, height, K, fill_ptr, scale=8, bilinear=False):
        """moped3d's DEPTHFILL on a device depth map [h, w, 4], in place, + its distance map [h, w]; asynchronous.
        -> the scale factor used."""
        k = np.ascontiguousarray(K, np.float32)
        used = C.c_int(0)
        self._ck(self.L.mh_depth_fill(self.h, C.c_void_p(depth_ptr), width, height, int(scale), 1 if bilinear else 0,
                                      _ptr(k), C.c_void_p(fill_ptr), C.byref(used)), "mh_depth_fill")
        return used.value

    def depth_fill_status(self):
        self._ck(self.L.mh_depth_fill_status(self.h), "mh_depth_fill_status")

    def depth_fill(self, depth_img, K, scale=8, bilinear=False):
        """The same on a host map: -> (filled copy [h, w, 4], distance map [h, w], scale used)."""
        d = np.ascontiguousarray(depth_img, np.float32).copy()
        h, w = d.shape[:2]
        dist = np.zeros((h, w), np.float32)
        k = np.ascontiguousarray(K, np.float32)
        used = C.c_int(0)
        self._ck(self.L.mh_depth_fill_host(self.h, _ptr(d), w, h, int(scale), 1 if bilinear else 0, _ptr(k), _ptr(dist),
                                           C.byref(used)), "mh_depth_fill_host")
        return d, dist, used.value

    def frame_set_depth_image_batch(self, depth_ptrs, fill_ptrs, width, height, kind, alpha=0.5, cauchy_scale=0.1):
        """One depth map (+ distance map, or None for all) per frame of the following batches."""
        n = len(depth_ptrs)
        d = (C.c_void_p * n)(*depth_ptrs)
        f = (C.c_void_p * n)(*fill_ptrs) if fill_ptrs else None
        self._ck(self.L.mh_frame_set_depth_image_batch(self.h, d, f, n, width, height, kind, alpha, cauchy_scale),
                 "mh_frame_set_depth_image_batch")

    # ---- FEAT ----
    def sift(self, gray, double_size=True, cap=16384):
        """-> (xy [n,2] = (col,row), scale_ori [n,2], desc [n,128]) in the reference's list order."""
        g = np.ascontiguousarray(gray, np.uint8)
        h, w = g.shape
        xy = np.zeros((cap, 2), np.float32)
        so = np.zeros((cap, 2), np.float32)
        d = np.zeros((cap, 128), np.float32)
        n = C.c_int32(0)
        self._ck(self.L.mh_sift_extract(self.h, _ptr(g), w, h, int(double_size), _ptr(xy), _ptr(so), _ptr(d), cap,
                                        C.byref(n)), "mh_sift_extract")
        return xy[:n.value].copy(), so[:n.value].copy(), d[:n.value].copy()

    def sift_dev(self, gray_ptr, w, h, double_size, desc_ptr, xy_ptr, scale_ori_ptr, cap, n_ptr):
        self._ck(self.L.mh_sift_extract_dev(self.h, C.c_void_p(gray_ptr), w, h, int(double_size), C.c_void_p(desc_ptr),
                                            C.c_void_p(xy_ptr), C.c_void_p(scale_ori_ptr) if scale_ori_ptr else None,
                                            cap, C.c_void_p(n_ptr)), "mh_sift_extract_dev")

    def project_test(self, pose7, corr, K, cam, thr):
        corr = np.ascontiguousarray(corr, CORR_DTYPE)
        n = corr.shape[0]
        inl = np.zeros(max(n, 1), np.uint8)
        e2 = np.zeros(max(n, 1), np.float32)
        cnt = C.c_int32(0)
        p = np.ascontiguousarray(pose7, np.float32)
        c = make_cam(K, cam)
        self._ck(self.L.mh_project_test(self.h, _ptr(p), _ptr(corr), n, C.byref(c), thr, _ptr(inl),
                                        _ptr(e2), C.byref(cnt)), "mh_project_test")
        return cnt.value, inl[:n].astype(bool), e2[:n]

    # ---- FILTER ----
    def filter(self, corr, model_off, obj_model, obj_pose, K, cam, min_points, feature_distance, min_score):
        corr = np.ascontiguousarray(corr, CORR_DTYPE)
        model_off = np.ascontiguousarray(model_off, np.int32)
        obj_model = np.ascontiguousarray(obj_model, np.int32)
        obj_pose = np.ascontiguousarray(obj_pose, np.float32)
        n_obj = obj_model.shape[0]
        M = corr.shape[0]
        score = np.zeros(max(n_obj, 1), np.float32)
        keep = np.zeros(max(n_obj, 1), np.uint8)
        order = np.zeros(max(n_obj, 1), np.int32)
        members = np.zeros(max(M, 1), np.int32)
        off = np.zeros(n_obj + 2, np.int32)
        kept = C.c_int32(0)
        c = make_cam(K, cam)
        self._ck(self.L.mh_filter(self.h, _ptr(corr), _ptr(model_off), len(model_off) - 1, _ptr(obj_model),
                                  _ptr(obj_pose), n_obj, C.byref(c), min_points, feature_distance,
                                  min_score, _ptr(score), _ptr(keep), _ptr(order), _ptr(members),
                                  _ptr(off), C.byref(kept)), "mh_filter")
        k = kept.value
        clusters = [members[off[i]:off[i + 1]].copy() for i in range(k)]
        return score[:n_obj], keep[:n_obj].astype(bool), order[:k].copy(), clusters

    def filter_images(self, corr, image_of, model_off, obj_model, obj_pose, Ks, cams, min_points, feature_distance,
                      min_score):
        """filter() for matches that come from several images (image_of[i] = image of match i)."""
        corr = np.ascontiguousarray(corr, CORR_DTYPE)
        image_of = np.ascontiguousarray(image_of, np.int32)
        model_off = np.ascontiguousarray(model_off, np.int32)
        obj_model = np.ascontiguousarray(obj_model, np.int32)
        obj_pose = np.ascontiguousarray(obj_pose, np.float32)
        n_obj = obj_model.shape[0]
        M = corr.shape[0]
        score = np.zeros(max(n_obj, 1), np.float32)
        keep = np.zeros(max(n_obj, 1), np.uint8)
        order = np.zeros(max(n_obj, 1), np.int32)
        members = np.zeros(max(M, 1), np.int32)
        off = np.zeros(n_obj + 2, np.int32)
        kept = C.c_int32(0)
        arr = make_cams(Ks, cams)
        self._ck(self.L.mh_filter_images(self.h, _ptr(corr), _ptr(image_of), _ptr(model_off), len(model_off) - 1,
                                         _ptr(obj_model), _ptr(obj_pose), n_obj, arr, len(Ks), min_points,
                                         feature_distance, min_score, _ptr(score), _ptr(keep), _ptr(order),
                                         _ptr(members), _ptr(off), C.byref(kept)), "mh_filter_images")
        k = kept.value
        clusters = [members[off[i]:off[i + 1]].copy() for i in range(k)]
        return score[:n_obj], keep[:n_obj].astype(bool), order[:k].copy(), clusters

    # ---- frame (device pointers as ints) ----
    def frame_enqueue(self, q_desc_ptr, q_uv_ptr, Q, K, cam, params: mh_frame_params, seed=1):
        c = make_cam(K, cam)
        self._ck(self.L.mh_frame_enqueue(self.h, C.c_void_p(q_desc_ptr), C.c_void_p(q_uv_ptr), Q,
                                         C.byref(c), C.byref(params), seed), "mh_frame_enqueue")

    def frame_run_host(self, q_desc, q_uv, Ks, cams, params: mh_frame_params, seed=1, q_image=None, write_back=True,
                       max_objects=4096):
        """The whole frame from host arrays, objects back on return (mh_frame_run_host: what FRAME_RESIDENT_HIP calls).
        q_desc [Q,128] float32 C-contiguous (normalised in place if write_back), q_uv [Q,2]; Ks [n,4], cams [n,7];
        q_image [Q] int32 when n > 1.  -> (objects, counts)."""
        assert q_desc.dtype == np.float32 and q_desc.flags.c_contiguous and q_desc.shape[1] == 128
        uv = np.ascontiguousarray(q_uv, np.float32)
        arr = make_cams(Ks, cams)
        img = None if q_image is None else np.ascontiguousarray(q_image, np.int32)
        objs = np.zeros(max_objects, OBJECT_DTYPE)
        n = C.c_int32(0)
        counts = np.zeros(4, np.int32)
        self._ck(self.L.mh_frame_run_host(self.h, _ptr(q_desc), _ptr(uv), None if img is None else _ptr(img), q_desc.shape[0],
                                          arr, len(Ks), C.byref(params), C.c_uint64(int(seed)), int(bool(write_back)),
                                          _ptr(objs), max_objects, C.byref(n), _ptr(counts)), "mh_frame_run_host")
        return objs[:min(n.value, max_objects)].copy(), counts

    def host_alloc(self, nbytes):
        """Page-locked host memory from the library (mh_host_alloc) as a uint8 array; host_free(array) releases it."""
        p = C.c_void_p()
        self._ck(self.L.mh_host_alloc(self.h, C.c_size_t(int(nbytes)), C.byref(p)), "mh_host_alloc")
        a = np.ctypeslib.as_array((C.c_uint8 * int(nbytes)).from_address(p.value))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, a):
        self._ck(self.L.mh_host_free(self.h, C.c_void_p(self._pinned.pop(a.ctypes.data))), "mh_host_free")

    def frame_run_host_begin(self, q_desc, q_uv, Ks, cams, params: mh_frame_params, seed=1, q_image=None, write_back=True):
        """mh_frame_run_host in halves: upload + enqueue (returns at once); then frame_wait_descriptors(), frame_fetch().
        The arrays must stay alive and untouched until then."""
        assert q_desc.dtype == np.float32 and q_desc.flags.c_contiguous and q_desc.shape[1] == 128
        assert q_uv.dtype == np.float32 and q_uv.flags.c_contiguous
        arr = make_cams(Ks, cams)
        self._rh_keep = (q_desc, q_uv, q_image, arr)
        self._ck(self.L.mh_frame_run_host_begin(self.h, _ptr(q_desc), _ptr(q_uv), None if q_image is None else _ptr(q_image),
                                                q_desc.shape[0], arr, len(Ks), C.byref(params), C.c_uint64(int(seed)),
                                                int(bool(write_back))), "mh_frame_run_host_begin")

    def frame_wait_descriptors(self):
        self._ck(self.L.mh_frame_wait_descriptors(self.h), "mh_frame_wait_descriptors")

    # ---- the six slots one call each on a frame that stays on the device (mh_step_*: the per-step plugins' hand-over) ----
    def step_match(self, q_desc, q_uv, K, cam, ratio=0.8, write_back=True):
        """MATCH: upload, normalise, search, ratio test, per-model lists on the device -> (model_off [n_models + 1],
        match_query [M], match_pts [M] CORR_DTYPE).  q_desc is normalised in place if write_back."""
        assert q_desc.dtype == np.float32 and q_desc.flags.c_contiguous and q_desc.shape[1] == 128
        uv = np.ascontiguousarray(q_uv, np.float32)
        Q = q_desc.shape[0]
        c = make_cam(K, cam)
        self._ck(self.L.mh_step_match(self.h, _ptr(q_desc), _ptr(uv), Q, C.byref(c), C.c_float(ratio), int(bool(write_back))),
                 "mh_step_match")
        if write_back:
            self.frame_wait_descriptors()
        nn_, nm_ = C.c_int(0), C.c_int(0)
        self._ck(self.L.mh_db_size(self.h, C.byref(nn_), C.byref(nm_)), "mh_db_size")
        nm = nm_.value
        off = np.zeros(nm + 1, np.int32)
        mq = np.zeros(Q, np.int32)
        pts = np.zeros(Q, CORR_DTYPE)
        n = C.c_int32(0)
        self._ck(self.L.mh_step_match_fetch(self.h, _ptr(off), _ptr(mq), _ptr(pts), Q, C.byref(n)), "mh_step_match_fetch")
        return off, mq[:n.value].copy(), pts[:n.value].copy()

    def step_cluster(self, radius=200.0, merge=20.0, min_pts=7, max_iter=100, cap_clusters=1024, cap_members=1 << 16):
        """CLUSTER on the resident lists -> (cl_model [n], cl_off [n + 1], members [cl_off[n]]: indices into matches[model])."""
        cm = np.zeros(cap_clusters, np.int32)
        co = np.zeros(cap_clusters + 1, np.int32)
        mem = np.zeros(cap_members, np.int32)
        n = C.c_int32(0)
        self._ck(self.L.mh_step_cluster(self.h, C.c_float(radius), C.c_float(merge), int(min_pts), int(max_iter), _ptr(cm), _ptr(co),
                                        _ptr(mem), cap_clusters, cap_members, C.byref(n)), "mh_step_cluster")
        return cm[:n.value].copy(), co[:n.value + 1].copy(), mem[:co[n.value]].copy()

    def step_pose(self, which, prm: mh_pose_params, seed, cap=4096):
        """POSE (which = 1) / POSE2 (2) on the resident clusters -> the objects the step appends (STEP_OBJECT_DTYPE)."""
        out = np.zeros(cap, STEP_OBJECT_DTYPE)
        n = C.c_int32(0)
        self._ck(self.L.mh_step_pose(self.h, int(which), C.byref(prm), C.c_uint64(int(seed)), _ptr(out), cap, C.byref(n)),
                 "mh_step_pose")
        return out[:n.value].copy()

    def step_filter(self, which, min_points, feature_distance, min_score, n_objects, cap_members=1 << 16):
        """FILTER (1) / FILTER2 (2) on the resident objects -> (score [n_objects], keep [n_objects], out_order [kept],
        cl_off [kept + 1], members)."""
        score = np.zeros(max(n_objects, 1), np.float32)
        keep = np.zeros(max(n_objects, 1), np.uint8)
        order = np.zeros(max(n_objects, 1), np.int32)
        co = np.zeros(n_objects + 1, np.int32)
        mem = np.zeros(cap_members, np.int32)
        kept = C.c_int32(0)
        self._ck(self.L.mh_step_filter(self.h, int(which), int(min_points), C.c_float(feature_distance), C.c_float(min_score),
                                       int(n_objects), _ptr(score), _ptr(keep), _ptr(order), _ptr(mem), _ptr(co), cap_members,
                                       C.byref(kept)), "mh_step_filter")
        k = kept.value
        return score[:n_objects].copy(), keep[:n_objects].copy(), order[:k].copy(), co[:k + 1].copy(), mem[:co[k]].copy()

    def frame_enqueue_image(self, gray_ptr, w, h, double_size, max_keypoints, K, cam, params: mh_frame_params,
                            seed=1, _cam_struct=None):
        """FEAT -> FILTER2 of a device-resident 8-bit image; nothing goes through the host."""
        c = _cam_struct or make_cam(K, cam)
        self._ck(self.L.mh_frame_enqueue_image(self.h, C.c_void_p(gray_ptr), w, h, int(double_size), max_keypoints,
                                               C.byref(c), C.byref(params), seed), "mh_frame_enqueue_image")

    def frame_enqueue_image_batch(self, gray_ptrs, w, h, double_size, max_keypoints, K, cam, params: mh_frame_params, seeds,
                                  _cam_struct=None):
        """FEAT of B device images, ONE MATCH launch sequence over all their keypoints, CLUSTER..FILTER2 image after
        image into result slots 0..B-1 (frame_fetch_slot)."""
        c = _cam_struct or make_cam(K, cam)
        B = len(gray_ptrs)
        g = (C.c_void_p * B)(*gray_ptrs)
        sd = (C.c_uint64 * B)(*[int(x) for x in seeds])
        self._ck(self.L.mh_frame_enqueue_image_batch(self.h, g, B, w, h, int(double_size), max_keypoints, C.byref(c),
                                                     C.byref(params), sd), "mh_frame_enqueue_image_batch")

    def set_linkage_scratch_limit(self, nbytes):
        """Bound of the linkage clusterer's similarity-matrix scratch for this context (0: the default, 4 GiB)."""
        self._ck(self.L.mh_set_linkage_scratch_limit(self.h, C.c_size_t(int(nbytes))), "mh_set_linkage_scratch_limit")

    def frame_set_cluster_linkage(self, params: "mh_linkage_params | None"):
        """CLUSTER of the next frames: moped3d's linkage clusterer (None: mean shift again)."""
        self._ck(self.L.mh_frame_set_cluster_linkage(self.h, C.byref(params) if params is not None else None),
                 "mh_frame_set_cluster_linkage")

    def cluster_linkage(self, problems, params=None):
        """problems: list of (uv [n,2], model_xyz [n,3], world_xyz [n,3]) -> list of (clusters, label);
        the depth map must have been set with frame_set_depth_image."""
        prm = params or default_linkage_params()
        sizes = [len(p[0]) for p in problems]
        off = np.zeros(len(problems) + 1, np.int32)
        off[1:] = np.cumsum(sizes)
        total = int(off[-1])
        if total:
            corr = np.concatenate([pack_corr(np.asarray(p[0], np.float32), np.asarray(p[1], np.float32)) for p in problems if len(p[0])])
            dep = np.concatenate([pack_depth(np.asarray(p[2], np.float32), np.ones(len(p[2]), np.float32)) for p in problems if len(p[0])])
        else:
            corr, dep = np.zeros(0, CORR_DTYPE), pack_depth(np.zeros((0, 3), np.float32), np.zeros(0, np.float32))
        corr, dep = np.ascontiguousarray(corr), np.ascontiguousarray(dep)
        label = np.full(max(total, 1), -1, np.int32)
        order = np.full(max(total, 1), -1, np.int32)
        ncl = np.zeros(max(len(problems), 1), np.int32)
        self._ck(self.L.mh_cluster_linkage(self.h, _ptr(corr), _ptr(dep), _ptr(off), len(problems), C.byref(prm),
                                           _ptr(label), _ptr(order), _ptr(ncl)), "mh_cluster_linkage")
        out = []
        for i in range(len(problems)):
            b, e = int(off[i]), int(off[i + 1])
            lab, ordr = label[b:e], order[b:e]
            clusters, pos = [], 0
            for c in range(int(ncl[i])):
                sz = int((lab == c).sum())
                clusters.append(ordr[pos:pos + sz].copy())
                pos += sz
            out.append((clusters, lab.copy()))
        return out

    def frame_set_depth_rules(self, K=None, patch_size=64, feature_density=-1.0, match_density=-1.0, ratio_table=None,
                              maximum_depth=4.0, default_depth=1.0, cauchy_scale=0.1, off=False):
        """moped3d's DEPTHFILTER / DEPTHFILTER2 / adaptive ratio inside the frame (needs frame_set_depth_image)."""
        if off:
            self._ck(self.L.mh_frame_set_depth_rules(self.h, None, None), "mh_frame_set_depth_rules")
            return
        r = mh_depth_rules(patch_size, feature_density, match_density, None, 0, maximum_depth, default_depth, cauchy_scale)
        tab = None
        if ratio_table is not None:
            tab = np.ascontiguousarray(ratio_table, np.float32)
            r.ratio_table = tab.ctypes.data
            r.n_models = tab.shape[0]
        k = np.ascontiguousarray(K if K is not None else [0, 0, 0, 0], np.float32)
        self._ck(self.L.mh_frame_set_depth_rules(self.h, C.byref(r), _ptr(k)), "mh_frame_set_depth_rules")

    def frame_fetch_matches(self, cap=1 << 16):
        q = np.zeros(cap, np.int32)
        m = np.zeros(cap, np.int32)
        n = C.c_int32(0)
        self._ck(self.L.mh_frame_fetch_matches(self.h, _ptr(q), _ptr(m), cap, C.byref(n)), "mh_frame_fetch_matches")
        return q[:min(n.value, cap)].copy(), m[:min(n.value, cap)].copy()

    def frame_fetch_matches_slot(self, slot, cap=1 << 16):
        """Accepted matches of frame `slot` of the last batch: (query, model), sorted by (model, query)."""
        q = np.zeros(cap, np.int32)
        m = np.zeros(cap, np.int32)
        n = C.c_int32(0)
        self._ck(self.L.mh_frame_fetch_matches_slot(self.h, slot, _ptr(q), _ptr(m), cap, C.byref(n)),
                 "mh_frame_fetch_matches_slot")
        return q[:min(n.value, cap)].copy(), m[:min(n.value, cap)].copy()

    def frame_enqueue_rest_frames(self, q_uv_ptr, Q, gathered_ptr, n_shards, stride_words, plane_words, B, K, cam,
                                  params: mh_frame_params, seeds, _cam_struct=None):
        c = _cam_struct or make_cam(K, cam)
        sd = (C.c_uint64 * B)(*[int(x) for x in seeds])
        self._ck(self.L.mh_frame_enqueue_rest_frames(self.h, C.c_void_p(q_uv_ptr), Q, C.c_void_p(gathered_ptr), n_shards,
                                                     stride_words, plane_words, B, C.byref(c), C.byref(params), sd),
                 "mh_frame_enqueue_rest_frames")

    def frame_features_dev(self):
        d, u, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ck(self.L.mh_frame_features_dev(self.h, C.byref(d), C.byref(u), C.byref(n)), "mh_frame_features_dev")
        return d.value, u.value, n.value

    def frame_keypoints(self):
        n = C.c_int32(0)
        self._ck(self.L.mh_frame_keypoints(self.h, C.byref(n)), "mh_frame_keypoints")
        return n.value

    def frame_enqueue_match_local(self, q_desc_ptr, Q, top2_ptr):
        """top2_ptr: device block of [3][Q] 32-bit words (idx1, d1 bits, d2 bits)."""
        self._ck(self.L.mh_frame_enqueue_match_local(self.h, C.c_void_p(q_desc_ptr), Q, C.c_void_p(top2_ptr)),
                 "mh_frame_enqueue_match_local")

    def frame_enqueue_rest(self, q_uv_ptr, Q, gathered_ptr, n_shards, K, cam, params: mh_frame_params,
                           seed=1, _cam_struct=None):
        """gathered_ptr: device block of [n_shards][3][Q] words (rank order)."""
        c = _cam_struct or make_cam(K, cam)
        self._ck(self.L.mh_frame_enqueue_rest(self.h, C.c_void_p(q_uv_ptr), Q, C.c_void_p(gathered_ptr),
                                              n_shards, C.byref(c), C.byref(params), seed),
                 "mh_frame_enqueue_rest")

    def frame_enqueue_rest_strided(self, q_uv_ptr, Q, gathered_ptr, n_shards, stride_words, K, cam,
                                   params: mh_frame_params, seed=1, _cam_struct=None):
        c = _cam_struct or make_cam(K, cam)
        self._ck(self.L.mh_frame_enqueue_rest_strided(self.h, C.c_void_p(q_uv_ptr), Q, C.c_void_p(gathered_ptr), n_shards,
                                                      stride_words, C.byref(c), C.byref(params), seed),
                 "mh_frame_enqueue_rest_strided")

    def frame_enqueue_rest_batch(self, q_uv_ptr, Q, gathered_ptr, n_shards, stride_words, plane_words, slot, K, cam,
                                 params: mh_frame_params, seed=1, _cam_struct=None):
        c = _cam_struct or make_cam(K, cam)
        self._ck(self.L.mh_frame_enqueue_rest_batch(self.h, C.c_void_p(q_uv_ptr), Q, C.c_void_p(gathered_ptr), n_shards,
                                                    stride_words, plane_words, slot, C.byref(c), C.byref(params), seed),
                 "mh_frame_enqueue_rest_batch")

    def frame_enqueue_batch(self, q_desc_ptr, q_uv_ptr, Q, B, K, cam, params: mh_frame_params, seeds, _cam_struct=None):
        c = _cam_struct or make_cam(K, cam)
        sd = (C.c_uint64 * B)(*[int(x) for x in seeds])
        self._ck(self.L.mh_frame_enqueue_batch(self.h, C.c_void_p(q_desc_ptr), C.c_void_p(q_uv_ptr), Q, B, C.byref(c),
                                               C.byref(params), sd), "mh_frame_enqueue_batch")

    def frame_enqueue_sharded(self, comm: Comm, q_desc_ptr, q_uv_ptr, Q, K, cam, params: mh_frame_params, seed=1,
                              _cam_struct=None):
        c = _cam_struct or make_cam(K, cam)
        self._ck(self.L.mh_frame_enqueue_sharded(self.h, comm.h, C.c_void_p(q_desc_ptr), C.c_void_p(q_uv_ptr), Q,
                                                 C.byref(c), C.byref(params), seed), "mh_frame_enqueue_sharded")

    def frame_enqueue_sharded_batch(self, comm: Comm, q_desc_ptr, q_uv_ptr, Q, B, K, cam, params: mh_frame_params,
                                    seeds, _cam_struct=None):
        c = _cam_struct or make_cam(K, cam)
        sd = (C.c_uint64 * B)(*[int(x) for x in seeds])
        self._ck(self.L.mh_frame_enqueue_sharded_batch(self.h, comm.h, C.c_void_p(q_desc_ptr), C.c_void_p(q_uv_ptr), Q, B,
                                                       C.byref(c), C.byref(params), sd),
                 "mh_frame_enqueue_sharded_batch")

    def frame_previous_objects(self, frame_in_batch=0, cap=8 * EX2_OBJECTS):
        objs = np.zeros(cap, OBJECT_DTYPE)
        n = C.c_int32(0)
        self._ck(self.L.mh_frame_previous_objects(self.h, frame_in_batch, _ptr(objs), cap, C.byref(n)),
                 "mh_frame_previous_objects")
        if n.value > cap:
            return self.frame_previous_objects(frame_in_batch, n.value)
        return objs[:n.value].copy()

    def frame_gather_objects(self, comm: Comm, slot=0, cap=4096):
        objs = np.zeros(cap, OBJECT_DTYPE)
        n = C.c_int32(0)
        self._ck(self.L.mh_frame_gather_objects(self.h, comm.h, slot, _ptr(objs), cap, C.byref(n)),
                 "mh_frame_gather_objects")
        return objs[:min(n.value, cap)].copy()

    def frame_fetch_slot(self, slot, max_objects=4096):
        objs = np.zeros(max_objects, OBJECT_DTYPE)
        n = C.c_int32(0)
        counts = np.zeros(4, np.int32)
        self._ck(self.L.mh_frame_fetch_slot(self.h, slot, _ptr(objs), max_objects, C.byref(n), _ptr(counts)),
                 "mh_frame_fetch_slot")
        return objs[:min(n.value, max_objects)].copy(), counts

    # ---- delivery of whole batches into pinned host memory (mh_frame_fetch_batch_async) ----
    def frame_fetch_batch_async(self, B, max_objects, host_ptr, tag=0):
        self._ck(self.L.mh_frame_fetch_batch_async(self.h, B, max_objects, C.c_void_p(host_ptr), tag & 0xFFFFFFFF),
                 "mh_frame_fetch_batch_async")

    def frame_fetch_previous_async(self, max_objects, host_ptr, tag=0):
        self._ck(self.L.mh_frame_fetch_previous_async(self.h, max_objects, C.c_void_p(host_ptr), tag & 0xFFFFFFFF),
                 "mh_frame_fetch_previous_async")

    def frame_fetch_wait(self):
        """Blocks until the context's pending delivery has landed; raises on capacity / exchange flags."""
        fl = C.c_int32(0)
        self._ck(self.L.mh_frame_fetch_wait(self.h, C.byref(fl)), "mh_frame_fetch_wait")

    def frame_fetch_query(self) -> bool:
        """True when the pending delivery (if any) has landed."""
        rc = self.L.mh_frame_fetch_query(self.h)
        if rc < 0:
            self._ck(rc, "mh_frame_fetch_query")
        return rc == 0

    def frame_result_copy_slots_dev(self, dst_ptr, n_slots, max_objects):
        self._ck(self.L.mh_frame_result_copy_slots_dev(self.h, C.c_void_p(dst_ptr), n_slots, max_objects),
                 "mh_frame_result_copy_slots_dev")

    def frame_result_copy_dev(self, dst_ptr, max_objects):
        self._ck(self.L.mh_frame_result_copy_dev(self.h, C.c_void_p(dst_ptr), max_objects), "mh_frame_result_copy_dev")

    def frame_fetch(self, max_objects=4096):
        objs = np.zeros(max_objects, OBJECT_DTYPE)
        n = C.c_int32(0)
        counts = np.zeros(4, np.int32)
        self._ck(self.L.mh_frame_fetch(self.h, _ptr(objs), max_objects, C.byref(n), _ptr(counts)), "mh_frame_fetch")
        return objs[:min(n.value, max_objects)].copy(), counts

    def frame_result_dev(self):
        p = C.c_void_p()
        b = C.c_int64(0)
        self._ck(self.L.mh_frame_result_dev(self.h, C.byref(p), C.byref(b)), "mh_frame_result_dev")
        return p.value, b.value

    def match_local_dev(self, qn_ptr, qnorm_ptr, Q, idx_ptr, d1_ptr, d2_ptr):
        self._ck(self.L.mh_match_local_dev(self.h, C.c_void_p(qn_ptr), C.c_void_p(qnorm_ptr), Q,
                                           C.c_void_p(idx_ptr), C.c_void_p(d1_ptr), C.c_void_p(d2_ptr)),
                 "mh_match_local_dev")

    def match_merge_dev(self, idx_s_ptr, d1_s_ptr, d2_s_ptr, n_shards, Q, idx_ptr, d1_ptr, d2_ptr):
        self._ck(self.L.mh_match_merge_dev(self.h, C.c_void_p(idx_s_ptr), C.c_void_p(d1_s_ptr),
                                           C.c_void_p(d2_s_ptr), n_shards, Q, C.c_void_p(idx_ptr),
                                           C.c_void_p(d1_ptr), C.c_void_p(d2_ptr)), "mh_match_merge_dev")

    def normalize_dev(self, q_ptr, qnorm_ptr, Q):
        self._ck(self.L.mh_normalize_dev(self.h, C.c_void_p(q_ptr), C.c_void_p(qnorm_ptr), Q), "mh_normalize_dev")


class ModelSet:
    """Host-side models (include/moped_hip.h "model files"): parsed `.moped.xml` files or a
    mapped `.mopeddb` container.  Host code only -- usable without a GPU."""

    def __init__(self, desc_type: str = "SIFT", _handle=None):
        self.L = load()
        if _handle is None:
            h = C.c_void_p()
            if self.L.mh_models_create(C.byref(h), desc_type.encode()) != 0:
                raise MhError("mh_models_create")
            _handle = h
        self.h = _handle

    @classmethod
    def load(cls, path: str) -> "ModelSet":
        L = load()
        h = C.c_void_p()
        if L.mh_models_load(C.byref(h), path.encode()) != 0:
            raise MhError(f"mh_models_load: not a valid .mopeddb file: {path}")
        return cls(_handle=h)

    def _ck(self, rc, what):
        if rc != 0:
            raise MhError(f"{what}: {self.L.mh_models_last_error(self.h).decode('latin-1')}")

    def add_xml(self, path: str):
        self._ck(self.L.mh_models_add_xml(self.h, path.encode()), "mh_models_add_xml")

    def add_xml_buffer(self, data: bytes):
        self._ck(self.L.mh_models_add_xml_buffer(self.h, data, len(data)), "mh_models_add_xml_buffer")

    def save(self, path: str):
        self._ck(self.L.mh_models_save(self.h, path.encode()), "mh_models_save")

    @property
    def n_models(self) -> int:
        return int(self.L.mh_models_count(self.h))

    @property
    def n_rows(self) -> int:
        return int(self.L.mh_models_rows(self.h))

    def name(self, i: int) -> str:
        return self.L.mh_models_name(self.h, i).decode("latin-1")

    def model_range(self, i: int):
        b, n = C.c_int64(0), C.c_int64(0)
        bbox = np.zeros(6, np.float32)
        self._ck(self.L.mh_models_range(self.h, i, C.byref(b), C.byref(n), _ptr(bbox)), "mh_models_range")
        return int(b.value), int(n.value), bbox

    def _view(self, ptr, cols):
        n = self.n_rows
        if n == 0 or not ptr:
            return np.zeros((0, cols), np.float32)
        buf = (C.c_float * (n * cols)).from_address(ptr)
        return np.frombuffer(buf, np.float32).reshape(n, cols)

    @property
    def desc(self) -> np.ndarray:
        """[rows,128] view (as parsed, not normalised); valid while the set lives and is not modified."""
        return self._view(self.L.mh_models_desc(self.h), 128)

    @property
    def xyz(self) -> np.ndarray:
        return self._view(self.L.mh_models_xyz(self.h), 3)

    @property
    def model_of(self) -> np.ndarray:
        out = np.zeros(self.n_rows, np.int32)
        for i in range(self.n_models):
            b, n, _ = self.model_range(i)
            out[b:b + n] = i
        return out

    def upload(self, ctx: "Context", first_model: int = 0, n_models: int | None = None):
        """Update() for a block of models: rows normalised on the device, ids stay global."""
        n_models = self.n_models - first_model if n_models is None else n_models
        ctx._ck(self.L.mh_db_upload_models(ctx.h, self.h, first_model, n_models), "mh_db_upload_models")

    def close(self):
        if self.h:
            self.L.mh_models_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
