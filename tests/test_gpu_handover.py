"""The six slots one call each on a frame that stays on the device (mh_step_*: the per-step plugins' hand-over,
include/moped_hip.h) -- what every call returns is what its slot writes into FrameData (src/util.hpp:68-110), and the
frame's final objects are bit for bit those of mh_frame_run_host (the six slots as one call) with the same constants
and seeds; the intermediate lists equal the upload-path entry points' and the oracle's."""
import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY


@pytest.fixture(scope="module")
def world():
    db = synth.make_db(12, 3000)
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    c.reserve(3000)
    yield db, c
    c.close()


def _stepped(c, fr, prm, seed):
    desc = fr.desc.copy()
    off, mq, pts = c.step_match(desc, fr.uv, K, CAM0, prm.ratio)
    cm, co, mem = c.step_cluster(prm.ms_radius, prm.ms_merge, prm.ms_min_pts, prm.ms_max_iter)
    o1 = c.step_pose(1, prm.pose1, seed)
    f1 = c.step_filter(1, prm.f1_min_points, prm.f1_feature_distance, prm.f1_min_score, len(o1))
    kept1 = o1[f1[2]]
    o2 = c.step_pose(2, prm.pose2, seed ^ 0x5DEECE66D)
    lst = np.concatenate([kept1, o2])
    f2 = c.step_filter(2, prm.f2_min_points, prm.f2_feature_distance, prm.f2_min_score, len(lst))
    return dict(desc=desc, off=off, mq=mq, pts=pts, cm=cm, co=co, mem=mem, o1=o1, f1=f1, o2=o2, lst=lst, f2=f2)


@pytest.mark.parametrize("seed,n_vis", [(0, 2), (1, 5), (2, 0), (3, 10)])
def test_six_step_calls_equal_the_frame_as_one_call(world, seed, n_vis):
    db, c = world
    fr = synth.make_frame(db, n_vis=n_vis, seed=seed)
    prm = capi.default_frame_params()
    want_desc = fr.desc.copy()
    want, wc = c.frame_run_host(want_desc, fr.uv, [K], [CAM0], prm, seed=seed + 40)
    r = _stepped(c, fr, prm, seed + 40)
    # MATCH: the descriptors normalised in place (MATCH_ANN_CPU.hpp:157), matches[m] in ascending query order
    assert np.array_equal(r["desc"].view(np.uint32), want_desc.view(np.uint32))
    idx, d1, d2 = orclib.match_2nn(orclib.normalize(db.desc), orclib.normalize(fr.desc))
    acc = np.nonzero((idx >= 0) & (d1 / d2 < np.float32(0.8)))[0]
    assert len(r["mq"]) == len(acc) == wc[0] == r["off"][-1]
    for m in range(db.n_models):
        q = r["mq"][r["off"][m]:r["off"][m + 1]]
        assert np.array_equal(q, acc[db.model_of[idx[acc]] == m])
        p = r["pts"][r["off"][m]:r["off"][m + 1]]
        assert np.array_equal(np.stack([p["x"], p["y"], p["z"]], 1), db.xyz[idx[q]]) and np.array_equal(p["u"], fr.uv[q, 0])
    # CLUSTER: the clusters of the upload-path entry point on the same lists, in (model, emission) order
    assert len(r["cm"]) == wc[1] and np.all(np.diff(r["cm"]) >= 0)
    for m in np.unique(r["cm"]):
        q = r["mq"][r["off"][m]:r["off"][m + 1]]
        clusters, label = c.meanshift(fr.uv[q], prm.ms_radius, prm.ms_merge, prm.ms_min_pts, prm.ms_max_iter)
        mine = [r["mem"][r["co"][k]:r["co"][k + 1]] for k in np.nonzero(r["cm"] == m)[0]]
        assert len(mine) == len(clusters)
        for members, ref in zip(mine, clusters):
            assert np.array_equal(members, ref)
    # the final objects: FILTER2's kept list, scores, bit for bit the one-call frame's
    score2, keep2, order2, co2, mem2 = r["f2"]
    final = r["lst"][order2]
    assert len(final) == len(want) == wc[3]
    assert np.array_equal(final["model"], want["model"])
    assert np.array_equal(final["pose"].view(np.uint32), want["pose"].view(np.uint32))
    assert np.array_equal(score2[order2].view(np.uint32), want["score"].view(np.uint32))
    assert np.array_equal(np.diff(co2), want["n_points"])
    assert int(keep2.sum()) == len(order2) and np.all(keep2[order2] == 1)
    if n_vis:
        assert set(fr.visible.tolist()) <= set(final["model"].tolist())
    # FILTER's clusters are index lists into matches[model] of points the kept object reprojects within FeatureDistance
    for k, i in enumerate(order2):
        m = int(r["lst"][i]["model"])
        members = mem2[co2[k]:co2[k + 1]]
        assert np.all(members >= 0) and np.all(members < r["off"][m + 1] - r["off"][m]) and np.all(np.diff(members) > 0)


def test_a_call_out_of_order_or_after_another_use_of_the_frame_arrays_is_refused(world):
    db, c = world
    fr = synth.make_frame(db, n_vis=2, seed=5)
    prm = capi.default_frame_params()
    c.step_match(fr.desc.copy(), fr.uv, K, CAM0)
    with pytest.raises(capi.MhError, match="not at the stage before"):
        c.step_pose(1, prm.pose1, 1)                      # CLUSTER has not run
    c.step_match(fr.desc.copy(), fr.uv, K, CAM0)
    c.step_cluster()
    with pytest.raises(capi.MhError, match="not at the stage before"):
        c.step_filter(1, 5, 4096.0, 2.0, 0)               # POSE has not run
    c.step_match(fr.desc.copy(), fr.uv, K, CAM0)
    c.step_cluster()
    objs = c.step_pose(1, prm.pose1, 1)
    with pytest.raises(capi.MhError, match="not the one the device holds"):
        c.step_filter(1, 5, 4096.0, 2.0, len(objs) + 1)   # the host's list is not the device's
    # a whole frame on the same context ends the hand-over
    c.step_match(fr.desc.copy(), fr.uv, K, CAM0)
    c.step_cluster()
    c.frame_run_host(fr.desc.copy(), fr.uv, [K], [CAM0], prm, seed=3)
    with pytest.raises(capi.MhError, match="not at the stage before"):
        c.step_pose(1, prm.pose1, 1)
    # ... and the next stepped frame is fine again
    r = _stepped(c, fr, prm, 9)
    assert len(r["f2"][2]) >= 2


def test_match_fetch_into_buffers_that_are_too_small_says_so_and_can_be_repeated(world):
    import ctypes as C
    db, c = world
    fr = synth.make_frame(db, n_vis=2, seed=6)
    desc = fr.desc.copy()
    cam = capi.make_cam(K, CAM0)
    uv = np.ascontiguousarray(fr.uv, np.float32)
    assert c.L.mh_step_match(c.h, desc.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p), len(desc), C.byref(cam),
                             C.c_float(0.8), 0) == 0
    off = np.zeros(db.n_models + 1, np.int32)
    mq, pts, n = np.zeros(4, np.int32), np.zeros(4, capi.CORR_DTYPE), C.c_int32(0)
    rc = c.L.mh_step_match_fetch(c.h, off.ctypes.data_as(C.c_void_p), mq.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p), 4,
                                 C.byref(n))
    assert rc == -3 and n.value > 4 and b"cap >= Q" in c.L.mh_last_error(c.h)      # MH_ERR_CAPACITY, the count is reported
    mq2, pts2 = np.zeros(n.value, np.int32), np.zeros(n.value, capi.CORR_DTYPE)
    rc = c.L.mh_step_match_fetch(c.h, off.ctypes.data_as(C.c_void_p), mq2.ctypes.data_as(C.c_void_p), pts2.ctypes.data_as(C.c_void_p),
                                 n.value, C.byref(n))
    assert rc == 0 and np.array_equal(mq2[:4], mq) and off[-1] == n.value
    cm, co, mem = c.step_cluster()                                                    # the resident frame went on being valid
    assert len(cm) >= 2
