"""Edge cases the reference's callers can produce: empty inputs, single rows,
capacity limits, repeated model updates.  None of them may fault or hang the GPU."""
import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY


def test_empty_and_tiny_calls(ctx, sift):
    base, _, _ = sift
    d = orclib.normalize(base[:10])
    # empty query set / empty DB: silent, like MATCH_ANN_CPU::process (:140,143)
    ctx.db_upload(d, np.zeros(10, np.int32), np.zeros((10, 3), np.float32), 1)
    acc, raw, d1, d2 = ctx.match(np.zeros((0, 128), np.float32))
    assert len(acc) == 0
    ctx.db_upload(np.zeros((0, 128), np.float32), np.zeros(0, np.int32), np.zeros((0, 3), np.float32), 0)
    acc, raw, d1, d2 = ctx.match(d[:3])
    assert np.all(raw == -1) and np.all(acc == -1)
    # one row: best found, no second best -> d2 = inf -> ratio 0 -> accepted (the reference skips N <= 1
    # in Update(), :102; the plugin keeps that rule, the C ABI reports what the arithmetic gives)
    ctx.db_upload(d[:1], np.zeros(1, np.int32), np.zeros((1, 3), np.float32), 1)
    acc, raw, d1, d2 = ctx.match(d[:3])
    assert np.all(raw == 0) and np.all(np.isinf(d2))
    cl, label = ctx.meanshift(np.zeros((0, 2), np.float32))
    assert cl == []
    out = ctx.pose_ransac(capi.pack_corr(np.zeros((0, 2)), np.zeros((0, 3))), np.array([0], np.int32), K, CAM0,
                          capi.make_pose_params())
    assert len(out) == 0
    s, k, o, c = ctx.filter(capi.pack_corr(np.zeros((0, 2)), np.zeros((0, 3))), np.array([0, 0], np.int32),
                            np.zeros(0, np.int32), np.zeros((0, 7), np.float32), K, CAM0, 5, 4096.0, 2.0)
    assert len(o) == 0


def test_meanshift_capacity_is_an_error_not_a_fault(ctx):
    pts = np.random.default_rng(0).uniform(0, 640, size=(2049, 2)).astype(np.float32)
    with pytest.raises(capi.MhError):
        ctx.meanshift(pts)
    cl, _ = ctx.meanshift(pts[:2048], 30.0, 5.0, 40, 3)   # the largest supported problem runs
    want, _ = orclib.meanshift(pts[:2048], 30.0, 5.0, 40, 3)
    assert [c.tolist() for c in cl] == [c.tolist() for c in want]


def test_pose_cluster_larger_than_lds_cache_is_flagged_not_faulting(ctx):
    rng = np.random.default_rng(1)
    n = 2500   # > POSE_MAX_PTS (2048): truncated to the first 2048 correspondences
    xyz = ((rng.random((n, 3)) - 0.5) * [0.1, 0.1, 0.2]).astype(np.float32)
    pose = np.concatenate([synth.random_quat(rng), [0.0, 0.0, 0.8]]).astype(np.float32)
    uv = (orclib.project(pose, xyz, K, CAM0) + rng.uniform(-.5, .5, (n, 2))).astype(np.float32)
    out = ctx.pose_ransac(capi.pack_corr(uv, xyz), np.array([0, n], np.int32), K, CAM0,
                          capi.make_pose_params(256, 1, 5, 6, 10.0, 5, 5))
    assert len(out) == 1
    assert np.sqrt(((orclib.project(out[0]["pose"], xyz, K, CAM0) - uv) ** 2).sum(1)).mean() < 1.0


def test_db_reupload_between_frames(ctx, sift):
    """modelsUpdated() -> Update(): a smaller DB after a larger one, then larger again."""
    base, _, _ = sift
    q = orclib.normalize(base[:200])
    for n_models, ppm in ((3, 700), (1, 130), (4, 900)):
        db = synth.make_db(n_models, ppm, seed=n_models)
        dbn = orclib.normalize(db.desc)
        ctx.db_upload(dbn, db.model_of, db.xyz, db.n_models)
        acc, raw, d1, d2 = ctx.match(q)
        oi, o1, o2 = orclib.match_2nn(dbn, q)
        assert np.array_equal(raw, oi) and np.array_equal(d1, o1) and np.array_equal(d2, o2)


def test_frame_with_many_models_few_matches(ctx):
    """200 models, almost all without a match: per-model workgroups must exit cleanly."""
    import torch
    from moped_amd.pipeline import FramePipeline, ShardedDB
    db = synth.make_db(200, 60)
    fr = synth.make_frame(db, n_vis=1, seed=2, Q=700, pts_per_obj=40)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=700)
    dev = torch.device("cuda:0")
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=1)
    objs, counts = pipe.fetch(0)
    dbn, qn = orclib.normalize(db.desc), orclib.normalize(fr.desc)
    idx, d1, d2 = orclib.match_2nn(dbn, qn)
    om, op, osc, oc = orclib.frame_rest(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0, n_threads=2, seed=1)
    assert counts[0] == oc[0] and counts[1] == oc[1]
    assert sorted(objs["model"].tolist()) == sorted(om.tolist())
    pipe.close()


def test_growing_query_count_reallocates(ctx, sift):
    import torch
    from moped_amd.pipeline import FramePipeline, ShardedDB
    db = synth.make_db(3, 500)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=256)
    dev = torch.device("cuda:0")
    for Q in (100, 5000, 300):
        fr = synth.make_frame(db, n_vis=1, seed=Q, Q=Q, pts_per_obj=60)
        pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=1)
        objs, counts = pipe.fetch(0)
        assert fr.visible[0] in objs["model"].tolist()
    pipe.close()


def test_frame_capacities_are_errors_not_faults():
    """A frame whose clusters / objects exceed what mh_reserve set aside: fetch reports MH_ERR_CAPACITY (sticky flags in the
    result header), nothing faults, and the context keeps working with larger capacities afterwards."""
    import torch
    from moped_amd import synth
    db = synth.make_db(8, 2000)
    fr = synth.make_frame(db, n_vis=6, seed=4, Q=3000, pts_per_obj=150)
    dev = torch.device("cuda:0")
    q = torch.from_numpy(fr.desc).to(dev)
    uv = torch.from_numpy(fr.uv).to(dev)
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    prm = capi.default_frame_params()
    c.reserve(3000, max_clusters=2, max_objects=4)          # six planted objects: > 2 clusters, > 4 object slots
    qa, qb = q.clone(), q.clone()                            # MATCH normalises in place
    torch.cuda.synchronize()                                 # (torch's stream made the copies; the context runs on its own)
    c.frame_enqueue(qa.data_ptr(), uv.data_ptr(), 3000, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, 5)
    with pytest.raises(capi.MhError, match="capacity"):
        c.frame_fetch()
    c.reserve(3000, max_clusters=1024, max_objects=4096)
    c.frame_enqueue(qb.data_ptr(), uv.data_ptr(), 3000, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, 5)
    objs, counts = c.frame_fetch()
    assert set(objs["model"].tolist()) == set(fr.visible.tolist())
    c.close()
