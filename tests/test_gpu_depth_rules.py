"""moped3d's rules on which features / matches reach CLUSTER, applied on the
device inside the frame: DEPTHFILTER (features), MATCH_ADAPTIVE_FLANN's depth-adaptive ratio,
DEPTHFILTER2 (each model's matches).  The device's accepted match lists must be exactly the
ones the oracle's restatement (orclib.depthfilter_keep / adaptive_ratio) selects from the
oracle's exact 2-NN search -- index-exact, like the fixed-ratio MATCH."""
import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY


@pytest.fixture(scope="module")
def scene():
    import torch
    from moped_amd.pipeline import FramePipeline, ShardedDB
    db = synth.make_db(6, 1500, seed=5)
    fr = synth.make_frame(db, n_vis=2, seed=21, Q=1600, pts_per_obj=140)
    img, fill = synth.depth_image(db, fr, seed=21, fill_max=0.3)
    img[300:, :200, 2] = 5.0                      # a corner beyond MaximumDepth (4 m)
    img[:40, :, 2] = np.nan                       # and a band of NaN depths
    dev = torch.device("cuda:0")
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=1600)
    c = pipe.ctxs[0]
    d_img, d_fill = torch.from_numpy(img).to(dev), torch.from_numpy(fill).to(dev)
    c.frame_set_depth_image(d_img.data_ptr(), d_fill.data_ptr(), 640, 480, capi.DEPTH_BACKPROJECTION, 0.5, 0.1)
    dbn, qn = orclib.normalize(db.desc), orclib.normalize(fr.desc)
    idx, d1, d2 = orclib.match_2nn(dbn, qn)
    table = np.stack([orclib.adaptive_control_points(db.xyz[db.model_of == m].min(0), db.xyz[db.model_of == m].max(0),
                                                     K, int((db.model_of == m).sum())) for m in range(db.n_models)])
    yield dict(db=db, fr=fr, img=img, fill=fill, pipe=pipe, c=c, torch=torch, dev=dev, idx=idx, d1=d1, d2=d2,
               table=table, keepalive=(d_img, d_fill))
    c.frame_set_depth_rules(off=True)
    c.frame_set_depth_image(0, 0, 0, 0, 0)
    pipe.close()


def _oracle_lists(s, feature_density, match_density, table, ratio=0.8):
    db, fr, idx, d1, d2 = s["db"], s["fr"], s["idx"], s["d1"], s["d2"]
    n = len(idx)
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
    order = np.lexsort((qs, model[qs]))                        # matches[model] lists, ascending query
    qs = qs[order]
    if match_density >= 0:
        off = np.searchsorted(model[qs], np.arange(db.n_models + 1))
        qs = qs[orclib.depthfilter_keep(s["img"], K, 64, match_density, fr.uv[qs], off)]
    return qs.astype(np.int32), model[qs].astype(np.int32)


@pytest.mark.parametrize("feature_density,match_density,adaptive", [
    (-1, -1, True), (0.02, -1, False), (-1, 0.004, False), (0.02, 0.004, True), (0.05, 0.01, True)])
def test_depth_rules_select_the_oracles_matches(scene, feature_density, match_density, adaptive):
    s = scene
    c, torch, dev, fr = s["c"], s["torch"], s["dev"], s["fr"]
    table = s["table"] if adaptive else None
    c.frame_set_depth_rules(K, 64, feature_density, match_density, table)
    s["pipe"].enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=3)
    objs, counts = s["pipe"].fetch(0)
    got_q, got_m = c.frame_fetch_matches()
    want_q, want_m = _oracle_lists(s, feature_density, match_density, table)
    assert np.array_equal(got_q, want_q) and np.array_equal(got_m, want_m)
    assert counts[0] == len(want_q)
    # the rules bite (and differ from the plain ratio test) without emptying the frame
    plain_q, _ = _oracle_lists(s, -1, -1, None)
    assert 0 < len(want_q) and not np.array_equal(want_q, plain_q)
    # planted objects survive the rules (one of the two sits partly on the NaN band / beyond MaximumDepth)
    assert len(objs) >= 1 and set(objs["model"].tolist()) <= set(fr.visible.tolist())


def test_depth_rules_off_is_the_plain_frame(scene):
    s = scene
    c, torch, dev, fr = s["c"], s["torch"], s["dev"], s["fr"]
    c.frame_set_depth_rules(off=True)
    s["pipe"].enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=3)
    s["pipe"].fetch(0)
    got_q, got_m = c.frame_fetch_matches()
    want_q, want_m = _oracle_lists(s, -1, -1, None)
    assert np.array_equal(got_q, want_q) and np.array_equal(got_m, want_m)
    # counts per (model, patch) are back to zero: a second run with the match filter gives the same lists
    for _ in range(2):
        c.frame_set_depth_rules(K, 64, -1, 0.004, None)
        s["pipe"].enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=3)
        s["pipe"].fetch(0)
        a = c.frame_fetch_matches()
        b = _oracle_lists(s, -1, 0.004, None)
        assert np.array_equal(a[0], b[0])


def test_batch_with_a_depth_map_per_frame_equals_single_frames():
    """mh_frame_set_depth_image_batch: the moped3d front end (depth rules, DEPTHMAP_PROP, linkage) on a batch of two
    frames, each with its own depth and distance maps -- objects and counts bit for bit the single frames'."""
    import torch
    from moped_amd import moped3d
    dev = torch.device("cuda:0")
    db = synth.make_db(6, 1500, seed=2)
    frs = [synth.make_frame(db, n_vis=2, seed=30 + i, Q=1500, pts_per_obj=130) for i in range(2)]
    maps = []
    for i, f in enumerate(frs):
        img, fill = synth.depth_image(db, f, seed=i, fill_max=0.3)
        maps.append((torch.from_numpy(img).to(dev), torch.from_numpy(fill).to(dev)))
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    Q = 1500
    c.reserve(2 * Q)
    table = moped3d.ratio_table(db.xyz, db.model_of, db.n_models, synth.K_DEFAULT)
    c.frame_set_depth_rules(synth.K_DEFAULT, 64, 0.05, 0.01, table)
    c.frame_set_cluster_linkage(capi.default_linkage_params())
    prm = capi.default_frame_params()
    alone = []
    for i, f in enumerate(frs):
        c.frame_set_depth_image(maps[i][0].data_ptr(), maps[i][1].data_ptr(), 640, 480, capi.DEPTH_BACKPROJECTION, 0.5, 0.1)
        qd, uv = torch.from_numpy(f.desc).to(dev), torch.from_numpy(f.uv).to(dev)
        c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, 5 + i)
        alone.append(c.frame_fetch())
    qd = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
    uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
    with pytest.raises(capi.MhError):     # one depth map, two frames
        c.frame_enqueue_batch(qd.data_ptr(), uv.data_ptr(), Q, 2, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, [5, 6])
    c.frame_set_depth_image_batch([m[0].data_ptr() for m in maps], [m[1].data_ptr() for m in maps], 640, 480,
                                  capi.DEPTH_BACKPROJECTION, 0.5, 0.1)
    qd = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
    c.frame_enqueue_batch(qd.data_ptr(), uv.data_ptr(), Q, 2, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, [5, 6])
    for f in range(2):
        objs, counts = c.frame_fetch_slot(f)
        a, ac = alone[f]
        assert np.array_equal(counts, ac) and len(a) >= 1
        assert np.array_equal(objs["model"], a["model"]) and np.array_equal(objs["pose"].view(np.uint32), a["pose"].view(np.uint32))
    c.close()


def test_merged_batch_of_five_frames_with_their_own_depth_maps_twice():
    """Round 4: the frames of mh_frame_enqueue_batch share their launches with the moped3d front end on too -- depth
    patches, DEPTHFILTER x 2, DEPTHMAP_PROP inside group_kernel and the linkage clusterer take frame f's depth map from
    a table.  Five frames with 0..3 visible objects, their maps and fill-distance maps, twice over the same arenas and
    rule buffers: objects and counts bit for bit those of the frames alone."""
    import torch
    from moped_amd import moped3d
    dev = torch.device("cuda:0")
    db = synth.make_db(6, 1500, seed=2)
    n_vis = (2, 0, 3, 1, 2)
    B, Q = len(n_vis), 1500
    frs = [synth.make_frame(db, n_vis=n, seed=60 + i, Q=Q, pts_per_obj=130) for i, n in enumerate(n_vis)]
    maps = []
    for i, f in enumerate(frs):
        img, fill = synth.depth_image(db, f, seed=10 + i, fill_max=0.3)
        maps.append((torch.from_numpy(img).to(dev), torch.from_numpy(fill).to(dev)))
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    c.reserve(B * Q)
    table = moped3d.ratio_table(db.xyz, db.model_of, db.n_models, synth.K_DEFAULT)
    c.frame_set_depth_rules(synth.K_DEFAULT, 64, 0.05, 0.01, table)
    c.frame_set_cluster_linkage(capi.default_linkage_params())
    prm = capi.default_frame_params()
    alone = []
    for i, f in enumerate(frs):
        c.frame_set_depth_image(maps[i][0].data_ptr(), maps[i][1].data_ptr(), 640, 480, capi.DEPTH_BACKPROJECTION, 0.5, 0.1)
        qd, uv = torch.from_numpy(f.desc).to(dev), torch.from_numpy(f.uv).to(dev)
        c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, 5 + i)
        alone.append(c.frame_fetch())
    assert sum(len(a[0]) for a in alone) >= 6
    uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
    c.frame_set_depth_image_batch([m[0].data_ptr() for m in maps], [m[1].data_ptr() for m in maps], 640, 480,
                                  capi.DEPTH_BACKPROJECTION, 0.5, 0.1)
    for rep in range(2):
        qd = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
        c.frame_enqueue_batch(qd.data_ptr(), uv.data_ptr(), Q, B, synth.K_DEFAULT, synth.CAM_IDENTITY, prm,
                              [5 + i for i in range(B)])
        for f in range(B):
            objs, counts = c.frame_fetch_slot(f)
            a, ac = alone[f]
            assert np.array_equal(counts, ac), (rep, f, counts, ac)
            assert np.array_equal(objs["model"], a["model"]), (rep, f)
            assert np.array_equal(objs["pose"].view(np.uint32), a["pose"].view(np.uint32)), (rep, f)
            assert np.array_equal(objs["score"].view(np.uint32), a["score"].view(np.uint32)), (rep, f)
    c.close()
