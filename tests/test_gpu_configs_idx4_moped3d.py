"""BASELINE configs[4] at its own size: the moped3d Kinect path against a 50-model database
(250,000 descriptors), Q = 3000 keypoints, one GPU.

  * MATCH: every query's (index, d1, d2) bit-exact against the oracle's exact search over all
    250 k rows (the oracle needs ~20-40 s for this one search; it is shared by the tests below).
  * depth-constrained POSE (POSE_RANSAC_LM_DIFF_BACKPROJECTION_DEPTH_CPU / ..REPROJECTION_DEPTH_CPU,
    kinds 1 and 2) on mean-shift clusters, the depth map looked up on the device: accepted matches
    and clusters index-exact, every planted object found, each pose within 1 px of the oracle's
    pose on the oracle's inliers and within 1 cm (kind 1) of the planted one.
  * the same with moped3d's shipped front end on the device (DEPTHFILTER x 2, depth-adaptive ratio,
    DEPTHMAP_PROP, CLUSTER_LINKAGE): match lists index-exact for ALL 50 models, the cluster table
    equal to the oracle's linkage clusters model by model, planted objects found."""
import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY
N_MODELS, PPM, Q, N_VIS = 50, 5000, 3000, 5


def _mean_reproj(pose, uv, xyz):
    return float(np.sqrt(((orclib.project(pose, xyz, K, CAM0) - uv) ** 2).sum(1)).mean())


@pytest.fixture(scope="module")
def cfg4():
    import torch
    from moped_amd.pipeline import FramePipeline, ShardedDB
    db = synth.make_db(N_MODELS, PPM)
    # seed 402: no two planted keypoints of different objects truncate to the same depth-map pixel (such a
    # collision hands one of them a depth 0.1-0.2 m off; the reference's back-projection objective is not
    # robust to that and its own pose -- the oracle's -- then lands 5-10 px away, seed to seed)
    fr = synth.make_frame(db, n_vis=N_VIS, seed=402, Q=Q)
    img, fill = synth.depth_image(db, fr, seed=402, fill_max=0.05)
    dev = torch.device("cuda:0")
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=Q)
    dbn, qn = orclib.normalize(db.desc), orclib.normalize(fr.desc)
    idx, d1, d2 = orclib.match_2nn(dbn, qn)            # the one full-size oracle search
    d_img, d_fill = torch.from_numpy(img).to(dev), torch.from_numpy(fill).to(dev)
    yield dict(db=db, fr=fr, img=img, fill=fill, pipe=pipe, c=pipe.ctxs[0], torch=torch, dev=dev, dbn=dbn, qn=qn,
               idx=idx, d1=d1, d2=d2, d_img=d_img, d_fill=d_fill)
    pipe.ctxs[0].frame_set_depth_rules(off=True)
    pipe.ctxs[0].frame_set_cluster_linkage(None)
    pipe.ctxs[0].frame_set_depth_image(0, 0, 0, 0, 0)
    pipe.close()


def test_config4_match_bit_exact_on_every_query(cfg4):
    s = cfg4
    torch, dev, c = s["torch"], s["dev"], s["c"]
    tq = torch.from_numpy(s["qn"]).to(dev)
    qnorm = torch.from_numpy(orclib.row_norms(s["qn"])).to(dev)
    out = [torch.empty(Q, dtype=t, device=dev) for t in (torch.int32, torch.float32, torch.float32)]
    c.match_local_dev(tq.data_ptr(), qnorm.data_ptr(), Q, *[o.data_ptr() for o in out])
    c.synchronize()
    gi, g1, g2 = [o.cpu().numpy() for o in out]
    assert np.array_equal(gi, s["idx"])
    assert np.array_equal(g1.view(np.uint32), s["d1"].view(np.uint32))
    assert np.array_equal(g2.view(np.uint32), s["d2"].view(np.uint32))
    planted = np.nonzero(s["fr"].src_point >= 0)[0]
    assert (gi[planted] == s["fr"].src_point[planted]).mean() > 0.99


@pytest.mark.parametrize("kind,scale", [(capi.DEPTH_BACKPROJECTION, 0.1), (capi.DEPTH_REPROJECTION, 25.0)])
def test_config4_depth_pose_on_mean_shift_clusters(cfg4, kind, scale):
    s = cfg4
    torch, dev, c, db, fr = s["torch"], s["dev"], s["c"], s["db"], s["fr"]
    c.frame_set_depth_rules(off=True)
    c.frame_set_cluster_linkage(None)
    c.frame_set_depth_image(s["d_img"].data_ptr(), s["d_fill"].data_ptr(), 640, 480, kind, 0.5, scale)
    s["pipe"].enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=17)
    objs, counts = s["pipe"].fetch(0)
    got_q, got_m = c.frame_fetch_matches()
    # accepted matches (fixed ratio) and mean-shift clusters: index-exact vs the oracle
    out_q, off = orclib.match_accept(s["idx"], s["d1"], s["d2"], 0.8, db.model_of, db.n_models)
    assert np.array_equal(got_q, out_q) and counts[0] == len(out_q)
    n_clusters = 0
    big = {}
    for m in range(db.n_models):
        qs = out_q[off[m]:off[m + 1]]
        if len(qs) == 0:
            continue
        clusters, _ = orclib.meanshift(fr.uv[qs])
        n_clusters += len(clusters)
        if m in fr.visible and clusters:
            big[m] = qs[max(clusters, key=len)]
    assert counts[1] == n_clusters
    # every planted object, once, at its planted pose
    assert sorted(objs["model"].tolist()) == sorted(fr.visible.tolist())
    world, wgt = orclib.depthmap_prop(s["img"], s["fill"], fr.uv, scale)
    for o in objs:
        m = int(o["model"])
        j = list(fr.visible).index(m)
        qs = big[m]
        uv, xyz = fr.uv[qs], db.xyz[s["idx"][qs]]
        ok, op = orclib.ransac_depth(kind, uv, xyz, world[qs], wgt[qs], K, CAM0, 0.5, orclib.POSE1_3D, seed=3)
        assert ok
        _, oinl = orclib.test_all_points(op, uv, xyz, K, CAM0, 8.0)
        # the north-star bar: within 1 px (mean reprojection) of the reference pose on the reference's inliers
        assert _mean_reproj(o["pose"], uv[oinl], xyz[oinl]) <= _mean_reproj(op, uv[oinl], xyz[oinl]) + 1.0
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
        rows = rows[db.model_of[fr.src_point[rows]] == m]
        # absolute: the depth-aware objectives trade pixels for depth by design (the oracle itself sits at
        # 0.4-2.1 px on these objects from seed to seed; the reprojection+depth class also shifts t, test_gpu_depth.py)
        assert _mean_reproj(o["pose"], fr.uv[rows], db.xyz[fr.src_point[rows]]) < 1.5
        assert np.linalg.norm(o["pose"][4:] - fr.poses[j][4:]) < (0.01 if kind == 1 else 0.02)


def _oracle_lists(s, feature_density, match_density, table, ratio=0.8):
    """The match lists moped3d's rules leave (tests/test_gpu_depth_rules.py, at this size)."""
    db, fr, idx, d1, d2 = s["db"], s["fr"], s["idx"], s["d1"], s["d2"]
    ok = idx >= 0
    if feature_density >= 0:
        ok &= orclib.depthfilter_keep(s["img"], K, 64, feature_density, fr.uv)
    model = np.where(idx >= 0, db.model_of[np.maximum(idx, 0)], -1)
    with np.errstate(all="ignore"):
        q = (d1 / d2).astype(np.float32)
    if table is not None:
        r, reach = orclib.adaptive_ratio(s["img"], s["fill"], fr.uv, model, table)
        ok &= reach & (q < r)
    else:
        ok &= q < np.float32(ratio)
    qs = np.nonzero(ok)[0]
    qs = qs[np.lexsort((qs, model[qs]))]
    if match_density >= 0:
        off = np.searchsorted(model[qs], np.arange(db.n_models + 1))
        qs = qs[orclib.depthfilter_keep(s["img"], K, 64, match_density, fr.uv[qs], off)]
    return qs.astype(np.int32), model[qs].astype(np.int32)


def test_config4_moped3d_front_end_at_full_size(cfg4):
    s = cfg4
    torch, dev, c, db, fr = s["torch"], s["dev"], s["c"], s["db"], s["fr"]
    table = np.stack([orclib.adaptive_control_points(db.xyz[db.model_of == m].min(0), db.xyz[db.model_of == m].max(0),
                                                     K, int((db.model_of == m).sum())) for m in range(db.n_models)])
    c.frame_set_depth_image(s["d_img"].data_ptr(), s["d_fill"].data_ptr(), 640, 480, capi.DEPTH_BACKPROJECTION, 0.5, 0.1)
    c.frame_set_depth_rules(K, 64, 0.02, 0.004, table)
    c.frame_set_cluster_linkage(capi.default_linkage_params())
    s["pipe"].enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=23)
    objs, counts = s["pipe"].fetch(0)
    got_q, got_m = c.frame_fetch_matches()
    want_q, want_m = _oracle_lists(s, 0.02, 0.004, table)
    assert np.array_equal(got_q, want_q) and np.array_equal(got_m, want_m)       # all 50 models' lists
    plain_q, _ = _oracle_lists(s, -1, -1, None)
    assert 0 < len(want_q) and not np.array_equal(want_q, plain_q)               # the rules bite
    # CLUSTER_LINKAGE model by model on the oracle
    n_clusters = 0
    for m in range(db.n_models):
        qs = want_q[want_m == m]
        if len(qs) == 0:
            continue
        world, _ = orclib.depthmap_prop(s["img"], s["fill"], fr.uv[qs], 0.1)
        n_clusters += len(orclib.cluster_linkage(fr.uv[qs], db.xyz[s["idx"][qs]], world, s["img"], s["fill"]))
    assert counts[0] == len(want_q) and counts[1] == n_clusters
    # planted objects come out at their planted poses (the density filters may drop one that sits on thin depth)
    assert len(objs) >= N_VIS - 1 and set(objs["model"].tolist()) <= set(fr.visible.tolist())
    for o in objs:
        j = list(fr.visible).index(int(o["model"]))
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
        rows = rows[db.model_of[fr.src_point[rows]] == o["model"]]
        assert _mean_reproj(o["pose"], fr.uv[rows], db.xyz[fr.src_point[rows]]) < 1.0
        assert np.linalg.norm(o["pose"][4:] - fr.poses[j][4:]) < 0.01
