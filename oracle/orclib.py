"""ctypes bindings for the CPU oracle (liboracle.so) and, where it was built, the
reference's own libraries (oracle/_ref/libmoped_ref*.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  moped_amd/ never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class PoseParams(C.Structure):
    _fields_ = [("max_ransac_tests", C.c_int), ("max_lm_tests", C.c_int),
                ("max_objects_per_cluster", C.c_int), ("n_pts_align", C.c_int),
                ("min_n_pts_object", C.c_int), ("error_threshold", C.c_float)]


class FrameParams(C.Structure):
    _fields_ = [("ratio", C.c_float), ("ms_radius", C.c_float), ("ms_merge", C.c_float),
                ("ms_min_pts", C.c_int), ("ms_max_iter", C.c_int), ("pose1", PoseParams),
                ("f1_min_points", C.c_int), ("f1_feature_distance", C.c_float), ("f1_min_score", C.c_float),
                ("pose2", PoseParams),
                ("f2_min_points", C.c_int), ("f2_feature_distance", C.c_float), ("f2_min_score", C.c_float),
                ("run_stage2", C.c_int)]


POSE1 = dict(max_ransac_tests=600, max_lm_tests=200, max_objects_per_cluster=4,
             n_pts_align=5, min_n_pts_object=6, error_threshold=10.0)   # config.hpp:110
POSE2 = dict(max_ransac_tests=100, max_lm_tests=500, max_objects_per_cluster=4,
             n_pts_align=6, min_n_pts_object=8, error_threshold=5.0)    # config.hpp:118


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        # ORC_LIB_PATH: another build of the same sources (`make -C oracle asan`: liboracle_asan.so under LD_PRELOAD=libasan)
        path = os.environ.get("ORC_LIB_PATH") or os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.orc_normalize.argtypes = [_f32p, C.c_int, C.c_int]
        L.orc_row_norms.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
        L.orc_match_2nn.argtypes = [_f32p, C.c_int, _f32p, C.c_int, C.c_int, _i32p, _f32p, _f32p, C.c_int]
        L.orc_match_accept.argtypes = [_i32p, _f32p, _f32p, C.c_int, C.c_float, _i32p, C.c_int, _i32p, _i32p]
        L.orc_match_accept.restype = C.c_int
        L.orc_match_merge.argtypes = [_i32p, _f32p, _f32p, C.c_int, C.c_int, _i32p, _f32p, _f32p]
        L.orc_meanshift.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                    _i32p, _i32p, C.POINTER(C.c_int)]
        L.orc_meanshift.restype = C.c_int
        L.orc_project.argtypes = [_f32p, _f32p, C.c_int, _f32p, _f32p, _f32p]
        L.orc_residuals.argtypes = [_f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p]
        L.orc_test_all_points.argtypes = [_f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, C.c_float, _u8p]
        L.orc_test_all_points.restype = C.c_int
        L.orc_optimize_camera.argtypes = [_f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, C.c_int, _f32p]
        L.orc_optimize_camera.restype = C.c_int
        L.orc_ransac.argtypes = [_f32p, _f32p, C.c_int, _f32p, _f32p, C.POINTER(PoseParams), _f32p]
        L.orc_ransac.restype = C.c_int
        L.orc_residuals_depth.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, C.c_float, _f32p]
        L.orc_optimize_camera_depth.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p,
                                                C.c_float, C.c_int, _f32p]
        L.orc_optimize_camera_depth.restype = C.c_int
        L.orc_ransac_depth.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, C.c_float,
                                       C.POINTER(PoseParams), _f32p]
        L.orc_ransac_depth.restype = C.c_int
        L.orc_filter.argtypes = [_f32p, _f32p, _i32p, C.c_int, _i32p, _f32p, C.c_int, _f32p, _f32p,
                                 C.c_int, C.c_float, C.c_float, _f32p, _u8p, _i32p, _i32p, _i32p]
        L.orc_filter.restype = C.c_int
        L.orc_frame_rest.argtypes = [_f32p, _i32p, _f32p, _f32p, C.c_int, C.c_float, _i32p, _f32p, C.c_int,
                                     _f32p, _f32p, C.POINTER(FrameParams), C.c_int, _i32p, _f32p, _f32p,
                                     C.c_int, _i32p]
        L.orc_frame_rest.restype = C.c_int
        L.orc_frame_rest_inliers.argtypes = L.orc_frame_rest.argtypes + [_i32p, _i32p, C.c_int]
        L.orc_frame_rest_inliers.restype = C.c_int
        L.srand = C.CDLL(None).srand
        _lib = L
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ---- thin numpy-level wrappers ------------------------------------------------

def normalize(desc):
    d = _c(desc, np.float32).copy()
    lib().orc_normalize(d, d.shape[0], d.shape[1])
    return d


def row_norms(desc):
    d = _c(desc, np.float32)
    out = np.empty(d.shape[0], np.float32)
    lib().orc_row_norms(d.reshape(-1), d.shape[0], d.shape[1], out)
    return out


def match_2nn(db, q, n_threads=0):
    db = _c(db, np.float32)
    q = _c(q, np.float32)
    Q = q.shape[0]
    idx = np.empty(Q, np.int32)
    d1 = np.empty(Q, np.float32)
    d2 = np.empty(Q, np.float32)
    dim = db.shape[1] if db.ndim == 2 and db.shape[0] else q.shape[1]
    lib().orc_match_2nn(db.reshape(-1), db.shape[0], q.reshape(-1), Q, dim, idx, d1, d2, n_threads)
    return idx, d1, d2


def match_accept(idx, d1, d2, ratio, model_of, n_models):
    Q = idx.shape[0]
    out_q = np.empty(max(Q, 1), np.int32)
    off = np.empty(n_models + 1, np.int32)
    m = lib().orc_match_accept(_c(idx, np.int32), _c(d1, np.float32), _c(d2, np.float32), Q,
                               ratio, _c(model_of, np.int32), n_models, out_q, off)
    return out_q[:m].copy(), off


def match_merge(idx_s, d1_s, d2_s):
    S, Q = idx_s.shape
    idx = np.empty(Q, np.int32)
    d1 = np.empty(Q, np.float32)
    d2 = np.empty(Q, np.float32)
    lib().orc_match_merge(_c(idx_s, np.int32).reshape(-1), _c(d1_s, np.float32).reshape(-1),
                          _c(d2_s, np.float32).reshape(-1), S, Q, idx, d1, d2)
    return idx, d1, d2


def meanshift(pts, radius=200.0, merge=20.0, min_pts=7, max_iter=100):
    """-> (list of member-index arrays in emission/splice order, iterations)."""
    pts = _c(pts, np.float32)
    n, dim = (pts.shape[0], pts.shape[1]) if pts.ndim == 2 else (0, 2)
    members = np.empty(max(n, 1), np.int32)
    off = np.empty(n + 2, np.int32)
    it = C.c_int(0)
    k = lib().orc_meanshift(pts.reshape(-1), n, dim, radius, merge, min_pts, max_iter,
                            members, off, C.byref(it))
    return [members[off[i]:off[i + 1]].copy() for i in range(k)], it.value


def project(pose7, xyz, K, cam):
    xyz = _c(xyz, np.float32)
    uv = np.empty((xyz.shape[0], 2), np.float32)
    lib().orc_project(_c(pose7, np.float32), xyz.reshape(-1), xyz.shape[0], _c(K, np.float32),
                      _c(cam, np.float32), uv.reshape(-1))
    return uv


def residuals(pose7, uv, xyz, K, cam):
    n = uv.shape[0]
    hx = np.empty(2 * n, np.float32)
    lib().orc_residuals(_c(pose7, np.float32), _c(uv, np.float32).reshape(-1),
                        _c(xyz, np.float32).reshape(-1), n, _c(K, np.float32),
                        _c(cam, np.float32), hx)
    return hx


def test_all_points(pose7, uv, xyz, K, cam, thr):
    n = uv.shape[0]
    inl = np.zeros(max(n, 1), np.uint8)
    c = lib().orc_test_all_points(_c(pose7, np.float32), _c(uv, np.float32).reshape(-1),
                                  _c(xyz, np.float32).reshape(-1), n, _c(K, np.float32),
                                  _c(cam, np.float32), thr, inl)
    return c, inl[:n].astype(bool)


test_all_points.__test__ = False  # not a pytest test


def optimize_camera(pose7, uv, xyz, K, cam, itmax):
    p = _c(pose7, np.float32).copy()
    info = np.zeros(3, np.float32)
    ret = lib().orc_optimize_camera(p, _c(uv, np.float32).reshape(-1), _c(xyz, np.float32).reshape(-1),
                                    uv.shape[0], _c(K, np.float32), _c(cam, np.float32), itmax, info)
    return ret, p, info


def ransac(uv, xyz, K, cam, params=POSE1, seed=None):
    if seed is not None:
        lib().srand(C.c_uint(seed))
    prm = PoseParams(**params)
    p = np.zeros(7, np.float32)
    ok = lib().orc_ransac(_c(uv, np.float32).reshape(-1), _c(xyz, np.float32).reshape(-1),
                          uv.shape[0], _c(K, np.float32), _c(cam, np.float32), C.byref(prm), p)
    return bool(ok), p


# moped3d constants (moped3d/libmoped/src/config.hpp:46,48)
POSE1_3D = dict(max_ransac_tests=192, max_lm_tests=100, max_objects_per_cluster=4,
                n_pts_align=5, min_n_pts_object=6, error_threshold=8.0)
POSE2_3D = dict(max_ransac_tests=64, max_lm_tests=250, max_objects_per_cluster=4,
                n_pts_align=6, min_n_pts_object=8, error_threshold=5.0)


def cauchy_weight(fill_distance, scale=0.1):
    """getCauchyWeight (…BACKPROJECTION_DEPTH_CPU.hpp:194-197; scale 0.1 there, 25 in
    …REPROJECTION_DEPTH_CPU.hpp:66)."""
    f = np.asarray(fill_distance, np.float32) / np.float32(scale)
    # `return 1.0 / (1 + factor*factor)`: float sum, double division, rounded to Float on return
    return (1.0 / (np.float32(1) + f * f).astype(np.float64)).astype(np.float32)


def depthmap_prop(depth_img, fill_img, uv, scale=0.1):
    """DEPTHMAP_PROP_CPU::process (moped3d/libmoped/src/depthprop/DEPTHMAP_PROP_CPU.hpp:101-134):
    per keypoint the pixel (int)u, (int)v of the 4-float depth map -> world3D = its x,y,z;
    fillDistance from the distance map or -1; -> (world [n,3], cauchy weight [n]).  Pixels
    outside the map are clamped (the reference indexes out of bounds there)."""
    h, w = depth_img.shape[:2]
    ix = np.clip(uv[:, 0].astype(np.int32), 0, w - 1)
    iy = np.clip(uv[:, 1].astype(np.int32), 0, h - 1)
    world = depth_img[iy, ix, :3].astype(np.float32)
    fd = fill_img[iy, ix].astype(np.float32) if fill_img is not None else np.full(len(uv), -1, np.float32)
    return world, cauchy_weight(fd, scale)


def residuals_depth(mode, pose7, uv, xyz, world, wgt, K, cam, alpha):
    n = uv.shape[0]
    err = np.empty((2 if mode == 1 else 3) * n, np.float32)
    lib().orc_residuals_depth(mode, _c(pose7, np.float32), _c(uv, np.float32).reshape(-1),
                              _c(xyz, np.float32).reshape(-1), _c(world, np.float32).reshape(-1),
                              _c(wgt, np.float32), n, _c(K, np.float32), _c(cam, np.float32), alpha, err)
    return err


def optimize_camera_depth(mode, pose7, uv, xyz, world, wgt, K, cam, alpha, itmax):
    p = _c(pose7, np.float32).copy()
    info = np.zeros(3, np.float32)
    ret = lib().orc_optimize_camera_depth(mode, p, _c(uv, np.float32).reshape(-1), _c(xyz, np.float32).reshape(-1),
                                          _c(world, np.float32).reshape(-1), _c(wgt, np.float32), uv.shape[0],
                                          _c(K, np.float32), _c(cam, np.float32), alpha, itmax, info)
    return ret, p, info


def ransac_depth(mode, uv, xyz, world, wgt, K, cam, alpha, params=POSE1_3D, seed=None):
    if seed is not None:
        lib().srand(C.c_uint(seed))
    prm = PoseParams(**params)
    p = np.zeros(7, np.float32)
    ok = lib().orc_ransac_depth(mode, _c(uv, np.float32).reshape(-1), _c(xyz, np.float32).reshape(-1),
                                _c(world, np.float32).reshape(-1), _c(wgt, np.float32), uv.shape[0],
                                _c(K, np.float32), _c(cam, np.float32), alpha, C.byref(prm), p)
    return bool(ok), p


def filter_projection(uv, xyz, model_off, obj_model, obj_pose, K, cam, min_points,
                      feature_distance, min_score):
    n_models = len(model_off) - 1
    n_obj = len(obj_model)
    M = uv.shape[0]
    score = np.zeros(max(n_obj, 1), np.float32)
    keep = np.zeros(max(n_obj, 1), np.uint8)
    order = np.zeros(max(n_obj, 1), np.int32)
    members = np.zeros(max(M, 1), np.int32)
    off = np.zeros(n_obj + 2, np.int32)
    k = lib().orc_filter(_c(uv, np.float32).reshape(-1), _c(xyz, np.float32).reshape(-1),
                         _c(model_off, np.int32), n_models, _c(obj_model, np.int32),
                         _c(obj_pose, np.float32).reshape(-1), n_obj, _c(K, np.float32),
                         _c(cam, np.float32), min_points, feature_distance, min_score,
                         score, keep, order, members, off)
    clusters = [members[off[i]:off[i + 1]].copy() for i in range(k)]
    return score[:n_obj], keep[:n_obj].astype(bool), order[:k].copy(), clusters


def _cams(Ks, cams):
    Ks = _c(np.asarray(Ks, np.float32).reshape(-1, 4), np.float32)
    cams = _c(np.asarray(cams, np.float32).reshape(-1, 7), np.float32)
    return Ks, cams, Ks.shape[0]


def ransac_images(uv, xyz, img, Ks, cams, params=POSE1, seed=None):
    """orc_ransac with every correspondence in its own image."""
    if seed is not None:
        lib().srand(C.c_uint(seed))
    Ks, cams, n = _cams(Ks, cams)
    pose = np.zeros(7, np.float32)
    pp = PoseParams(**params)
    L = lib()
    L.orc_ransac_images.restype = C.c_int
    L.orc_ransac_images.argtypes = [_f32p, _f32p, _i32p, C.c_int, _f32p, _f32p, C.c_int, C.c_void_p, _f32p]
    ok = L.orc_ransac_images(_c(uv, np.float32).reshape(-1), _c(xyz, np.float32).reshape(-1), _c(img, np.int32),
                             len(uv), Ks.reshape(-1), cams.reshape(-1), n, C.addressof(pp), pose)
    return bool(ok), pose


def project_images(pose7, xyz, img, Ks, cams):
    """project() of every point through its own image's camera."""
    Ks, cams, n = _cams(Ks, cams)
    out = np.zeros((len(xyz), 2), np.float32)
    img = np.asarray(img, np.int32)
    for i in range(n):
        sel = img == i
        if sel.any():
            out[sel] = project(pose7, np.asarray(xyz, np.float32)[sel], Ks[i], cams[i])
    return out


def filter_images(uv, img, xyz, model_off, obj_model, obj_pose, Ks, cams, min_points, feature_distance, min_score):
    Ks, cams, n = _cams(Ks, cams)
    n_obj = len(obj_model)
    M = len(uv)
    score = np.zeros(max(n_obj, 1), np.float32)
    keep = np.zeros(max(n_obj, 1), np.uint8)
    order = np.zeros(max(n_obj, 1), np.int32)
    members = np.zeros(max(M, 1), np.int32)
    off = np.zeros(n_obj + 2, np.int32)
    L = lib()
    L.orc_filter_images.restype = C.c_int
    L.orc_filter_images.argtypes = [_f32p, _i32p, _f32p, _i32p, C.c_int, _i32p, _f32p, C.c_int, _f32p, _f32p, C.c_int,
                                    C.c_int, C.c_float, C.c_float, _f32p, C.c_void_p, _i32p, _i32p, _i32p]
    kept = L.orc_filter_images(_c(uv, np.float32).reshape(-1), _c(img, np.int32), _c(xyz, np.float32).reshape(-1),
                               _c(model_off, np.int32), len(model_off) - 1, _c(obj_model, np.int32),
                               _c(obj_pose, np.float32).reshape(-1), n_obj, Ks.reshape(-1), cams.reshape(-1), n,
                               min_points, feature_distance, min_score, score, keep.ctypes.data, order, members, off)
    clusters = [members[off[i]:off[i + 1]].copy() for i in range(kept)]
    return score[:n_obj], keep[:n_obj].astype(bool), order[:kept].copy(), clusters


def frame_rest_images(q_uv, q_img, idx1, d1, d2, model_of, db_xyz, n_models, Ks, cams, params=None, seed=None,
                      max_obj=4096):
    """CPU frame after the NN search for a frame with several images -> (models, poses, scores, counts[4])."""
    if seed is not None:
        lib().srand(C.c_uint(seed))
    fp = params or default_frame_params()
    Ks, cams, n = _cams(Ks, cams)
    om = np.zeros(max_obj, np.int32)
    op = np.zeros(max_obj * 7, np.float32)
    osc = np.zeros(max_obj, np.float32)
    counts = np.zeros(4, np.int32)
    q_uv = _c(q_uv, np.float32)
    L = lib()
    L.orc_frame_rest_images.restype = C.c_int
    L.orc_frame_rest_images.argtypes = [_f32p, _i32p, _i32p, _f32p, _f32p, C.c_int, C.c_float, _i32p, _f32p, C.c_int,
                                        _f32p, _f32p, C.c_int, C.c_void_p, _i32p, _f32p, _f32p, C.c_int, _i32p]
    k = L.orc_frame_rest_images(q_uv.reshape(-1), _c(q_img, np.int32), _c(idx1, np.int32), _c(d1, np.float32),
                                _c(d2, np.float32), q_uv.shape[0], fp.ratio, _c(model_of, np.int32),
                                _c(db_xyz, np.float32).reshape(-1), n_models, Ks.reshape(-1), cams.reshape(-1), n,
                                C.addressof(fp), om, op, osc, max_obj, counts)
    k = min(k, max_obj)
    return om[:k].copy(), op[:7 * k].reshape(k, 7).copy(), osc[:k].copy(), counts


def default_frame_params(run_stage2=True):
    """The shipped constants of moped2/libmoped/src/config.hpp:83-120."""
    return FrameParams(0.8, 200.0, 20.0, 7, 100, PoseParams(**POSE1), 5, 4096.0, 2.0,
                       PoseParams(**POSE2), 7, 4096.0, 3.0, int(run_stage2))


def frame_rest(q_uv, idx1, d1, d2, model_of, db_xyz, n_models, K, cam, params=None, n_threads=1,
               seed=None, max_obj=4096):
    """CPU frame after the NN search -> (objects structured array, counts[4])."""
    if seed is not None:
        lib().srand(C.c_uint(seed))
    fp = params or default_frame_params()
    om = np.zeros(max_obj, np.int32)
    op = np.zeros(max_obj * 7, np.float32)
    osc = np.zeros(max_obj, np.float32)
    counts = np.zeros(4, np.int32)
    q_uv = _c(q_uv, np.float32)
    n = lib().orc_frame_rest(q_uv.reshape(-1), _c(idx1, np.int32), _c(d1, np.float32), _c(d2, np.float32),
                             q_uv.shape[0], fp.ratio, _c(model_of, np.int32), _c(db_xyz, np.float32).reshape(-1),
                             n_models, _c(K, np.float32), _c(cam, np.float32), C.byref(fp), n_threads,
                             om, op, osc, max_obj, counts)
    n = min(n, max_obj)
    return om[:n].copy(), op[:7 * n].reshape(n, 7).copy(), osc[:n].copy(), counts


def frame_rest_inliers(q_uv, idx1, d1, d2, model_of, db_xyz, n_models, K, cam, params=None, n_threads=1,
                       seed=None, max_obj=4096):
    """frame_rest + the inlier set of every final object: (models, poses, scores, counts, inliers) with inliers[o] =
    rows of q_uv (testAllPoints of the object's final pose over its final cluster at the last POSE stage's
    ErrorThreshold, POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:166-180)."""
    if seed is not None:
        lib().srand(C.c_uint(seed))
    fp = params or default_frame_params()
    om = np.zeros(max_obj, np.int32)
    op = np.zeros(max_obj * 7, np.float32)
    osc = np.zeros(max_obj, np.float32)
    counts = np.zeros(4, np.int32)
    q_uv = _c(q_uv, np.float32)
    Q = q_uv.shape[0]
    off = np.zeros(max_obj + 1, np.int32)
    inl = np.zeros(max(Q, 1), np.int32)    # every correspondence belongs to at most one final cluster
    n = lib().orc_frame_rest_inliers(q_uv.reshape(-1), _c(idx1, np.int32), _c(d1, np.float32), _c(d2, np.float32),
                                     Q, fp.ratio, _c(model_of, np.int32), _c(db_xyz, np.float32).reshape(-1),
                                     n_models, _c(K, np.float32), _c(cam, np.float32), C.byref(fp), n_threads,
                                     om, op, osc, max_obj, counts, off, inl, inl.shape[0])
    n = min(n, max_obj)
    return (om[:n].copy(), op[:7 * n].reshape(n, 7).copy(), osc[:n].copy(), counts,
            [inl[off[o]:off[o + 1]].copy() for o in range(n)])


# ---- the reference's own libraries (only where oracle/_ref was built) ----------

_ref = {}


def ref_available(fast=False):
    return os.path.exists(os.path.join(HERE, "_ref", "libmoped_ref_fast.so" if fast else "libmoped_ref.so"))


def ref(fast=False):
    key = bool(fast)
    if key not in _ref:
        path = os.path.join(HERE, "_ref", "libmoped_ref_fast.so" if fast else "libmoped_ref.so")
        R = C.CDLL(path)
        R.ref_ann_build.argtypes = [_f32p, C.c_int, C.c_int]
        R.ref_ann_build.restype = C.c_void_p
        R.ref_ann_search2.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_float, _i32p, _f32p]
        R.ref_ann_free.argtypes = [C.c_void_p]
        R.ref_project.argtypes = [_f32p, _f32p, C.c_int, _f32p, _f32p, _f32p]
        R.ref_residuals.argtypes = [_f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p]
        R.ref_optimize_camera.argtypes = [_f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, C.c_int, _f32p]
        R.ref_optimize_camera.restype = C.c_int
        R.ref_residuals_depth.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p,
                                          C.c_float, _f32p]
        R.ref_optimize_camera_depth.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p,
                                                C.c_float, C.c_int, _f32p]
        R.ref_optimize_camera_depth.restype = C.c_int
        R.ref_sift.argtypes = [_u8p, C.c_int, C.c_int, _f32p, _f32p, C.c_int]
        R.ref_sift.restype = C.c_int
        _ref[key] = R
    return _ref[key]


class RefAnn:
    """ANN kd-tree over a DB, searched the way MATCH_ANN_CPU does."""

    def __init__(self, db, fast=False):
        self.R = ref(fast)
        db = _c(db, np.float32)
        self.h = self.R.ref_ann_build(db.reshape(-1), db.shape[0], db.shape[1])

    def search2(self, q, eps):
        q = _c(q, np.float32)
        idx = np.empty((q.shape[0], 2), np.int32)
        dist = np.empty((q.shape[0], 2), np.float32)
        self.R.ref_ann_search2(self.h, q.reshape(-1), q.shape[0], eps, idx.reshape(-1), dist.reshape(-1))
        return idx, dist

    def close(self):
        if self.h:
            self.R.ref_ann_free(self.h)
            self.h = None


def ref_project(pose7, xyz, K, cam):
    xyz = _c(xyz, np.float32)
    uv = np.empty((xyz.shape[0], 2), np.float32)
    ref().ref_project(_c(pose7, np.float32), xyz.reshape(-1), xyz.shape[0], _c(K, np.float32),
                      _c(cam, np.float32), uv.reshape(-1))
    return uv


def ref_residuals(pose7, uv, xyz, K, cam):
    n = uv.shape[0]
    hx = np.empty(2 * n, np.float32)
    ref().ref_residuals(_c(pose7, np.float32), _c(uv, np.float32).reshape(-1),
                        _c(xyz, np.float32).reshape(-1), n, _c(K, np.float32), _c(cam, np.float32), hx)
    return hx


def ref_residuals_depth(mode, pose7, uv, xyz, world, wgt, K, cam, alpha):
    n = uv.shape[0]
    err = np.empty((2 if mode == 1 else 3) * n, np.float32)
    ref().ref_residuals_depth(mode, _c(pose7, np.float32), _c(uv, np.float32).reshape(-1),
                              _c(xyz, np.float32).reshape(-1), _c(world, np.float32).reshape(-1),
                              _c(wgt, np.float32), n, _c(K, np.float32), _c(cam, np.float32), alpha, err)
    return err


def ref_optimize_camera_depth(mode, pose7, uv, xyz, world, wgt, K, cam, alpha, itmax):
    p = _c(pose7, np.float32).copy()
    info = np.zeros(10, np.float32)
    ret = ref().ref_optimize_camera_depth(mode, p, _c(uv, np.float32).reshape(-1), _c(xyz, np.float32).reshape(-1),
                                          _c(world, np.float32).reshape(-1), _c(wgt, np.float32), uv.shape[0],
                                          _c(K, np.float32), _c(cam, np.float32), alpha, itmax, info)
    return ret, p, info


def ref_optimize_camera(pose7, uv, xyz, K, cam, itmax):
    p = _c(pose7, np.float32).copy()
    info = np.zeros(10, np.float32)
    ret = ref().ref_optimize_camera(p, _c(uv, np.float32).reshape(-1), _c(xyz, np.float32).reshape(-1),
                                    uv.shape[0], _c(K, np.float32), _c(cam, np.float32), itmax, info)
    return ret, p, info


# ---------------------------------------------------------------------------- N3: model files
def parse_model_xml(path, desc_type="SIFT", dim=128):
    """CPU restatement (test infrastructure, small files) of how the reference reads a
    `.moped.xml` model: sXML's tokenizer (moped2/libmoped/include/sXML.hpp:53-118: element
    name, name="value" properties with its backslash rule, nested children, `<!-- -->`
    before an element) and Moped::addModel(sXML&) (src/moped.cpp:101-137: name = root
    property, LAST child called Points, every child of it is a point, p3d -> 3 floats,
    desc -> floats until the first bad token, filed by desc_type, bbox over all points).
    -> dict(name, xyz [n,3], desc [n,dim], bbox [6], n_bad_len)."""
    data = open(path, "rb").read().decode("latin-1")
    pos = 0
    n = len(data)

    def token():                      # sXML::getToken (:55-61)
        nonlocal pos
        b = pos
        while pos < n and not data[pos].isspace() and data[pos] not in ">=":
            pos += 1
        t = data[b:pos]
        while pos < n and data[pos].isspace():
            pos += 1
        return t

    def element():                    # sXML::process (:63-118) -> (name, props, children) or None at a close tag
        nonlocal pos
        while data[pos] != "<":
            pos += 1
        pos += 1
        name = token()
        while name == "!--":          # comment in front of an element (:73-83)
            end = data.index("-->", pos)
            pos = end + 2             # the reference leaves the stream ON the final '>' ...
            while data[pos] != "<":   # ... and then looks for the next '<'
                pos += 1
            pos += 1
            name = token()
        if name == "" or name[0] == "/" or name[-1] == "/":
            return (name, {}, [])
        props = {}
        while data[pos] != "/":       # properties (:85-103)
            pname = token()
            if pname == "" or data[pos] != "=":
                break
            while data[pos] != '"':
                pos += 1
            pos += 1
            val = []
            while data[pos] != '"':
                if data[pos] == "\\":
                    pos += 1
                    if data[pos] == "n":
                        val.append("\n")
                        pos += 1
                val.append(data[pos])
                pos += 1
            pos += 1
            while data[pos].isspace():
                pos += 1
            props[pname] = "".join(val)
        children = []
        while data[pos] == ">":       # children (:105-113)
            ch = element()
            if ch[0][:1] == "/":
                return (name, props, children)
            if ch[0] != "":
                children.append(ch)
        while data[pos] != ">":
            pos += 1
        return (name, props, children)

    def floats(text, cap=None):       # `while (jss >> f)`: blanks between, stop at the first bad token
        out = []
        for tok in text.split():
            try:
                if tok.lower().lstrip("+-")[:3] in ("nan", "inf"):
                    break
                out.append(np.float32(tok))
            except ValueError:
                break
            if cap is not None and len(out) == cap:
                break
        return out

    root = element()
    name = root[1].get("name", "")
    points = None
    for ch in root[2]:
        if ch[0] == "Points":
            points = ch
    lo = np.full(3, np.float32(10E10), np.float32)
    hi = np.full(3, np.float32(-10E10), np.float32)
    xyz, desc, bad = [], [], 0
    for pt in (points[2] if points else []):
        c = np.zeros(3, np.float32)
        v = floats(pt[1].get("p3d", ""), 3)
        c[:len(v)] = v
        lo, hi = np.minimum(lo, c), np.maximum(hi, c)
        if pt[1].get("desc_type", "") != desc_type:
            continue
        d = floats(pt[1].get("desc", ""))
        bad += len(d) != dim
        row = np.zeros(dim, np.float32)
        row[:min(len(d), dim)] = d[:dim]
        xyz.append(c)
        desc.append(row)
    return dict(name=name, xyz=np.array(xyz, np.float32).reshape(-1, 3), desc=np.array(desc, np.float32).reshape(-1, dim),
                bbox=np.concatenate([lo, hi]).astype(np.float32), n_bad_len=bad)


def ref_model_xml(path, desc_type="SIFT", dim=128, cap=1 << 16):
    """The same through the reference's own sXML.hpp + stream operators (oracle/_ref)."""
    R = ref()
    R.ref_model_xml.restype = C.c_int
    R.ref_model_xml.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                C.c_void_p, C.POINTER(C.c_int)]
    xyz = np.zeros((cap, 3), np.float32)
    desc = np.zeros((cap, dim), np.float32)
    name = C.create_string_buffer(256)
    bbox = np.zeros(6, np.float32)
    bad = C.c_int(0)
    k = R.ref_model_xml(path.encode(), desc_type.encode(), xyz.ctypes.data, desc.ctypes.data, cap, dim, name, 256,
                        bbox.ctypes.data, C.byref(bad))
    if k < 0:
        return None
    return dict(name=name.value.decode("latin-1"), xyz=xyz[:k].copy(), desc=desc[:k].copy(), bbox=bbox, n_bad_len=bad.value)


# ---------------------------------------------------------------------------- N2: SIFT
def sift(gray, double_size=True, cap=16384):
    """orc_sift -> (xy [n,2] = (col,row), scale_ori [n,2], desc [n,128]) in the reference's list order."""
    g = _c(gray, np.uint8)
    h, w = g.shape
    L = lib()
    L.orc_sift.restype = C.c_int
    L.orc_sift.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    xy = np.zeros((cap, 2), np.float32)
    so = np.zeros((cap, 2), np.float32)
    d = np.zeros((cap, 128), np.float32)
    n = L.orc_sift(g.ctypes.data, w, h, int(double_size), xy.ctypes.data, so.ctypes.data, d.ctypes.data, cap)
    n = min(n, cap)
    return xy[:n].copy(), so[:n].copy(), d[:n].copy()


def sift_image(gray, octave, kind, i, double_size=True):
    """One pyramid image of the oracle run (kind 0 Gaussian, 1 DoG)."""
    g = _c(gray, np.uint8)
    h, w = g.shape
    L = lib()
    L.orc_sift_image.restype = C.c_int
    L.orc_sift_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
    r, c = C.c_int(0), C.c_int(0)
    n = L.orc_sift_image(g.ctypes.data, w, h, int(double_size), octave, kind, i, None, C.byref(r), C.byref(c))
    if n == 0:
        return None
    out = np.zeros((r.value, c.value), np.float32)
    L.orc_sift_image(g.ctypes.data, w, h, int(double_size), octave, kind, i, out.ctypes.data, C.byref(r), C.byref(c))
    return out


def ref_sift(gray, cap=16384):
    """The reference's own libsiftfast build (oracle/_ref), called like FEAT_SIFT_CPU does."""
    g = _c(gray, np.uint8)
    h, w = g.shape
    R = ref()
    R.ref_sift2.restype = C.c_int
    R.ref_sift2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    xy = np.zeros((cap, 2), np.float32)
    so = np.zeros((cap, 2), np.float32)
    d = np.zeros((cap, 128), np.float32)
    n = min(R.ref_sift2(g.ctypes.data, w, h, xy.ctypes.data, so.ctypes.data, d.ctypes.data, cap), cap)
    return xy[:n].copy(), so[:n].copy(), d[:n].copy()


# ---- moped3d depth rules: DEPTHFILTER_CPU and MATCH_ADAPTIVE_FLANN_CPU's ratio ----
# Restated from the source text in numpy with the reference's float / double mix (Float = float,
# unsuffixed literals = double).  moped3d's step classes need OpenCV headers -> no reference build.

_f = np.float32


def _patch_of(uv, patch, pw, ph):
    """((int) location) / PatchSize (DEPTHFILTER_CPU.hpp:187), clamped to the patch grid."""
    px = np.clip(uv[:, 0].astype(np.int32) // patch, 0, pw - 1)
    py = np.clip(uv[:, 1].astype(np.int32) // patch, 0, ph - 1)
    return py * pw + px


def depth_patch_inv_size(depth_img, K, patch):
    """Per patch 1.0 / sizeMap (double): minDepthMap (DEPTHFILTER_CPU.hpp:146-166, std::min: NaN never
    wins), getArea at that depth (:50-61, with `y1 = min(.., width)` as written, :170)."""
    h, w = depth_img.shape[:2]
    pw, ph = -(-w // patch), -(-h // patch)
    K = np.asarray(K, _f)
    z = depth_img[:, :, 2].astype(_f)
    inv = np.empty(pw * ph, np.float64)

    def pt(u, v, d):
        return np.array([(_f(u) - K[2]) / K[0] * d, (_f(v) - K[3]) / K[1] * d, d], _f)

    def dist(a, b):
        r = _f(0)
        for x in range(3):
            d = _f(b[x] - a[x])
            r = _f(r + _f(d * d))
        return np.sqrt(r, dtype=_f)

    with np.errstate(all="ignore"):
        for py in range(ph):
            for px in range(pw):
                x0, y0, x1 = px * patch, py * patch, min((px + 1) * patch, w)
                blk = z[y0:min(y0 + patch, h), x0:x1].ravel()
                blk = blk[~np.isnan(blk)]
                m = _f(1e10)
                if len(blk) and blk.min() < m:
                    m = blk.min()
                y1 = min((py + 1) * patch, w)
                c0, c1, c2 = pt(x0, y0, m), pt(x0, y1, m), pt(x1, y0, m)
                area = _f(dist(c0, c2) * dist(c0, c1))
                inv[py * pw + px] = np.float64(1.0) / np.float64(area)
    return inv, pw, ph


def _density_replay(n, inv):
    c = _f(0)
    for _ in range(int(n)):
        c = _f(np.float64(c) + inv)      # countMap[..] += 1.0 / sizeMap[..] on a Float (:189)
    return c


def depthfilter_keep(depth_img, K, patch, density, uv, group_off=None):
    """DEPTHFILTER_CPU::process (:181-249): keep[i] for points uv, filtered per group (ToFilter = 1:
    one group = all features; ToFilter = 2: one group per model's matches)."""
    inv, pw, ph = depth_patch_inv_size(depth_img, K, patch)
    filt = _f(_f(_f(density) * _f(100)) * _f(100))            # Float filter = Density*100*100 (:132)
    n = len(uv)
    if group_off is None:
        group_off = [0, n]
    keep = np.zeros(n, bool)
    with np.errstate(all="ignore"):
        for g in range(len(group_off) - 1):
            a, b = int(group_off[g]), int(group_off[g + 1])
            if b <= a:
                continue
            p = _patch_of(uv[a:b], patch, pw, ph)
            cnt = np.bincount(p, minlength=pw * ph)
            val = np.array([_density_replay(cnt[i], inv[i]) for i in range(pw * ph)], _f).reshape(ph, pw)
            dil = val.copy()                                  # dilate (:76-113): 3x3 max of the copy
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    ys, yd = slice(max(dy, 0), ph + min(dy, 0)), slice(max(-dy, 0), ph + min(-dy, 0))
                    xs, xd = slice(max(dx, 0), pw + min(dx, 0)), slice(max(-dx, 0), pw + min(-dx, 0))
                    src = val[ys, xs]
                    upd = src > dil[yd, xd]
                    dil[yd, xd] = np.where(upd, src, dil[yd, xd])
            keep[a:b] = dil.ravel()[p] > filt
    return keep


def _ratio_at(depth, cp, max_depth):
    """MATCH_ADAPTIVE_FLANN_CPU::getRatio (moped3d/.../MATCH_ADAPTIVE_FLANN_CPU.hpp:193-215)."""
    depth = _f(depth)
    max_rd, min_rd, lo, hi = (_f(x) for x in cp)
    if depth > _f(max_depth):
        return _f(0)
    if depth < max_rd:
        progress = _f(depth / max_rd)
        return _f(lo + _f(progress * _f(hi - lo)))
    if depth < min_rd:
        return hi
    if depth < _f(min_rd * _f(2)):
        progress = _f(_f(_f(min_rd * _f(2)) - depth) / min_rd)
        return _f(progress * hi)
    return _f(0)


def adaptive_ratio(depth_img, fill_img, uv, model_of_nn, table, max_depth=4.0, default_depth=1.0, cauchy_scale=0.1):
    """getAdjustedRatio (:361-376) for every query given the model of its nearest neighbour ->
    (ratio float32 [n], reachable bool [n]: depth <= MaximumDepth, :457-460)."""
    h, w = depth_img.shape[:2]
    n = len(uv)
    ratio = np.zeros(n, _f)
    reach = np.zeros(n, bool)
    with np.errstate(all="ignore"):
        for i in range(n):
            x = min(max(int(uv[i, 0]), 0), w - 1)
            y = min(max(int(uv[i, 1]), 0), h - 1)
            depth = _f(depth_img[y, x, 2])
            reach[i] = not (depth > _f(max_depth))
            m = int(model_of_nn[i])
            if m < 0:
                continue
            fill = _f(fill_img[y, x]) if fill_img is not None else _f(0)
            wt = _f(fill / _f(cauchy_scale))
            weight = _f(np.float64(1.0) / (np.float64(1.0) + np.float64(_f(wt * wt))))
            put, dft = _ratio_at(depth, table[m], max_depth), _ratio_at(default_depth, table[m], max_depth)
            ratio[i] = _f(np.float64(_f(weight * put)) + (np.float64(1.0) - np.float64(weight)) * np.float64(dft))
    return ratio, reach


def adaptive_control_points(bbox_min, bbox_max, K, n_features, min_ratio=(0.6, 0.75), max_ratio=(0.65, 0.8),
                            dimension_peak=150.0, dimension_fade=50.0):
    """MATCH_ADAPTIVE_FLANN_CPU::Update's per-model control points (:100-177 with
    solveProjectionDepth :318-357, getAverageProjectedLength :262-312, getProjectedArea :238-256):
    -> (maxRatioDepth, minRatioDepth, ratioLow, ratioHigh).  Defaults = moped3d config.hpp:43."""
    K = [float(_f(k)) for k in K]
    rng = [float(_f(bbox_max[i]) - _f(bbox_min[i])) for i in range(3)]

    def projected_area(pts):
        us = [_f(_f(_f(K[0]) * p[0] + _f(K[2]) * p[2]) / p[2]) for p in pts]
        vs = [_f(_f(_f(K[1]) * p[1] + _f(K[3]) * p[2]) / p[2]) for p in pts]
        return _f(_f(max(us) - min(us)) * _f(max(vs) - min(vs)))

    def avg_len(depth):
        xr, yr, zr = (_f(r) for r in rng)
        mnx, mxx, mny, mxy, mnz, mxz = _f(-xr / 2), _f(xr / 2), _f(-yr / 2), _f(yr / 2), _f(-zr / 2), _f(zr / 2)
        zc, yc, xc = _f((mxx - mnx) * (mxy - mny)), _f((mxx - mnx) * (mxz - mnz)), _f((mxy - mny) * (mxz - mnz))
        d = _f(depth)
        if zc >= xc and zc >= yc:
            s = [(mnx, mny, d), (mnx, mxy, d), (mxx, mxy, d), (mxx, mny, d)]
        elif yc >= xc and yc >= zc:
            s = [(mnx, mnz, d), (mnx, mxz, d), (mxx, mxz, d), (mxx, mnz, d)]
        else:
            s = [(mny, mnz, d), (mny, mxz, d), (mxy, mxz, d), (mxy, mnz, d)]
        return np.sqrt(projected_area(s), dtype=_f)

    def solve(target, iters=100, tol=0.01):
        left, right, it = _f(0), _f(2), 0
        target = _f(target)
        while it < iters:
            it += 1
            if avg_len(right) > target:
                right = _f(right * 2)
            else:
                break
        max_err = _f(target * _f(tol))
        while it < iters:                      # the reference keeps counting with the same `iter`
            it += 1
            mid = _f(_f(left + right) / 2)
            length = avg_len(mid)
            if abs(_f(length - target)) < max_err:
                return mid
            if length > target:
                left = mid
            else:
                right = mid
        return _f(_f(left + right) / 2)

    d_peak, d_fade = solve(dimension_peak), solve(dimension_fade)
    adj = _f(1.0 / (1.0 + np.exp(-1.0 * float(_f((_f(1750) - _f(n_features)) / _f(250))))))   # canonicalSigmoid
    lo = _f(_f(min_ratio[0]) + adj * _f(_f(min_ratio[1]) - _f(min_ratio[0])))
    hi = _f(_f(max_ratio[0]) + adj * _f(_f(max_ratio[1]) - _f(max_ratio[0])))
    return np.array([d_peak, d_fade, lo, hi], _f)


def cluster_linkage(uv, model_xyz, world_xyz, depth_img, fill_img, cutoff=0.1, min_pts=7, use3d_filter=2,
                    linkage_type=1, sigma2d=-1.0, sigma3d=-1.0, want_k=False):
    """CLUSTER_LINKAGE_CPU for one model's matches (moped3d config.hpp:45 defaults) ->
    (list of member-index arrays in the reference's order[, K])."""
    n = len(uv)
    h, w = depth_img.shape[:2]
    members = np.zeros(max(n, 1), np.int32)
    off = np.zeros(n + 1, np.int32)
    K = np.zeros((n, n), np.float32) if want_k else None
    L = lib()
    L.orc_cluster_linkage.restype = C.c_int
    L.orc_cluster_linkage.argtypes = [_f32p, _f32p, _f32p, C.c_int, _f32p, C.c_int, C.c_int, C.c_void_p, C.c_float,
                                      C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _i32p, _i32p, C.c_void_p]
    fill = _c(fill_img, np.float32) if fill_img is not None else None
    ncl = L.orc_cluster_linkage(_c(uv, np.float32).reshape(-1), _c(model_xyz, np.float32).reshape(-1),
                                _c(world_xyz, np.float32).reshape(-1), n, _c(depth_img, np.float32).reshape(-1), w, h,
                                fill.ctypes.data if fill is not None else None, cutoff, min_pts, use3d_filter,
                                linkage_type, sigma2d, sigma3d, members, off,
                                K.ctypes.data if K is not None else None)
    cl = [members[off[c]:off[c + 1]].copy() for c in range(ncl)]
    return (cl, K) if want_k else cl


def depth_fill(depth_img, K, scale=8, bilinear=False):
    """DEPTH_FILL_EXACT_CPU (moped3d/libmoped/src/depthfill/DEPTH_FILL_EXACT_CPU.hpp:268-349; moped3d config.hpp:39
    ships (8, false)): depth_img [h, w, 4] (z < 0 = hole) -> (filled copy, distance map [h, w], scale used)."""
    h, w = depth_img.shape[:2]
    out = np.ascontiguousarray(depth_img, np.float32).copy()
    dist = np.zeros((h, w), np.float32)
    L = lib()
    L.orc_depth_fill.restype = C.c_int
    L.orc_depth_fill.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p]
    used = L.orc_depth_fill(out.reshape(-1), w, h, int(scale), 1 if bilinear else 0,
                            np.ascontiguousarray(K, np.float32), dist.reshape(-1))
    if used < 0:
        raise ValueError("orc_depth_fill: bad arguments")
    return out, dist, used
