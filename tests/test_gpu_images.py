"""Frames with several images (cameras) on the GPU against the oracle's restatement of what the reference does
with FrameData::images: CLUSTER per (model, image) (CLUSTER_MEAN_SHIFT_CPU.hpp:189-195), POSE with every
correspondence in its own image (…REPROJECTION_CPU.hpp:76-98, 213-237), FILTER keyed by (coord2D, image)
(FILTER_PROJECTION_CPU.hpp:89-141) -- step by step through the host-pointer entry points and as a device-resident
frame (mh_frame_set_images)."""
import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu
K = synth.K_DEFAULT


@pytest.fixture(scope="module")
def scene():
    db = synth.make_db(6, 1500, seed=8)
    cams = [synth.camera_pose(0.0), synth.camera_pose(-0.12, (0.10, 0.0, 0.0))]
    fr = synth.make_frame_images(db, cams, n_vis=2, seed=5, q_per_image=900, pts_per_obj=130)
    dbn, qn = orclib.normalize(db.desc), orclib.normalize(fr.desc)
    idx, d1, d2 = orclib.match_2nn(dbn, qn)
    out_q, off = orclib.match_accept(idx, d1, d2, 0.8, db.model_of, db.n_models)
    return dict(db=db, fr=fr, idx=idx, d1=d1, d2=d2, out_q=out_q, off=off, uv=fr.uv[out_q], img=fr.image[out_q],
                xyz=db.xyz[idx[out_q]])


def _err(pose, uv, xyz, img, fr):
    return float(np.sqrt(((orclib.project_images(pose, xyz, img, fr.Ks, fr.cams) - uv) ** 2).sum(1)).mean())


def test_filter_images_matches_oracle(ctx, scene):
    s = scene
    fr, off = s["fr"], s["off"]
    rng = np.random.default_rng(3)
    obj_model, obj_pose = [], []
    for j, m in enumerate(fr.visible):                      # the planted pose + perturbed copies + a wrong-model object
        for k in range(4):
            p = fr.poses[j].copy()
            p[4:] += rng.normal(0, 0.002 * k, 3).astype(np.float32)
            obj_model.append(int(m))
            obj_pose.append(p)
    obj_model.append(int((fr.visible[0] + 1) % s["db"].n_models))
    obj_pose.append(fr.poses[0])
    order_in = np.argsort(np.array(obj_model), kind="stable")      # (model, list) order is the caller's business
    obj_model = np.array(obj_model, np.int32)[order_in]
    obj_pose = np.array(obj_pose, np.float32)[order_in]
    corr = capi.pack_corr(s["uv"], s["xyz"])
    for mp, fd, ms in ((5, 4096.0, 2.0), (7, 4096.0, 3.0), (3, 25.0, 0.5)):
        g = ctx.filter_images(corr, s["img"], off, obj_model, obj_pose, fr.Ks, fr.cams, mp, fd, ms)
        o = orclib.filter_images(s["uv"], s["img"], s["xyz"], off, obj_model, obj_pose, fr.Ks, fr.cams, mp, fd, ms)
        assert np.array_equal(g[0].view(np.uint32), o[0].view(np.uint32))          # scores bit for bit
        assert np.array_equal(g[1], o[1]) and np.array_equal(g[2], o[2])
        assert len(g[3]) == len(o[3]) and all(np.array_equal(a, b) for a, b in zip(g[3], o[3]))
        assert g[1].any()
    # the image is part of the ownership key: squeezing everything into image 0 changes the result
    g0 = ctx.filter_images(corr, np.zeros_like(s["img"]), off, obj_model, obj_pose, fr.Ks, fr.cams, 5, 4096.0, 2.0)
    g1 = ctx.filter_images(corr, s["img"], off, obj_model, obj_pose, fr.Ks, fr.cams, 5, 4096.0, 2.0)
    assert not np.array_equal(g0[0], g1[0])


def test_pose_ransac_images_mixed_clusters(ctx, scene):
    """POSE2-like clusters: all matches of a visible model, from both images."""
    s = scene
    fr, off = s["fr"], s["off"]
    prm = capi.make_pose_params(1024, 4, 6, 8, 5.0, 10, 10)
    clusters, coff = [], [0]
    for m in fr.visible:
        sl = slice(off[m], off[m + 1])
        clusters.append((s["uv"][sl], s["xyz"][sl], s["img"][sl]))
        coff.append(coff[-1] + (off[m + 1] - off[m]))
    uv = np.concatenate([c[0] for c in clusters])
    xyz = np.concatenate([c[1] for c in clusters])
    img = np.concatenate([c[2] for c in clusters])
    out = ctx.pose_ransac_images(capi.pack_corr(uv, xyz), img, np.array(coff, np.int32), fr.Ks, fr.cams, prm, seed=4)
    assert sorted(set(out["cluster"].tolist())) == list(range(len(clusters)))
    for c, (cuv, cxyz, cimg) in enumerate(clusters):
        assert set(cimg.tolist()) == {0, 1}
        ok, op = orclib.ransac_images(cuv, cxyz, cimg, fr.Ks, fr.cams, orclib.POSE2, seed=1)
        assert ok
        d = orclib.project_images(op, cxyz, cimg, fr.Ks, fr.cams) - cuv
        oinl = (d ** 2).sum(1) < 5.0
        for o in out[out["cluster"] == c]:
            assert _err(o["pose"], cuv[oinl], cxyz[oinl], cimg[oinl], fr) <= _err(op, cuv[oinl], cxyz[oinl], cimg[oinl], fr) + 1.0
            assert np.linalg.norm(o["pose"][4:] - fr.poses[c][4:]) < 0.01


def test_frame_with_two_cameras_matches_oracle(scene):
    import torch
    from moped_amd.pipeline import FramePipeline, ShardedDB
    s = scene
    db, fr = s["db"], s["fr"]
    dev = torch.device("cuda:0")
    Q = len(fr.uv)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=Q)
    c = pipe.ctxs[0]
    q_img = torch.from_numpy(fr.image).to(dev)
    c.frame_set_images(q_img.data_ptr(), fr.Ks, fr.cams)
    res = []
    for rep in range(2):
        pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=9)
        res.append(pipe.fetch(0))
    objs, counts = res[0]
    assert np.array_equal(res[1][0]["pose"], objs["pose"])                       # deterministic
    om, op, osc, oc = orclib.frame_rest_images(fr.uv, fr.image, s["idx"], s["d1"], s["d2"], db.model_of, db.xyz,
                                               db.n_models, fr.Ks, fr.cams, seed=2)
    assert counts[0] == oc[0]                      # accepted matches
    assert counts[1] == oc[1]                      # clusters: per (model, image), the oracle's count
    assert sorted(objs["model"].tolist()) == sorted(om.tolist()) == sorted(fr.visible.tolist())
    for m, p, sc in zip(om, op, osc):
        g = objs[objs["model"] == m][0]
        j = list(fr.visible).index(m)
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
        rows = rows[db.model_of[fr.src_point[rows]] == m]
        xyz, uv, img = db.xyz[fr.src_point[rows]], fr.uv[rows], fr.image[rows]
        e_o, e_g = _err(p, uv, xyz, img, fr), _err(g["pose"], uv, xyz, img, fr)
        assert e_g <= e_o + 1.0 and e_g < 1.0
        assert np.linalg.norm(g["pose"][4:] - fr.poses[j][4:]) < 0.005
        assert abs(g["score"] - sc) <= 0.05 * sc               # FILTER2's score over BOTH images' matches
    # one image again: the plain frame (image 1's keypoints now read as image 0's: different clusters)
    c.frame_set_images(0)
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=9)
    _, counts1 = pipe.fetch(0)
    assert counts1[0] == counts[0] and counts1[1] != counts[1]
    pipe.close()


def test_batch_of_two_camera_frames_equals_the_frames_alone(scene):
    """Frames with several images through mh_frame_enqueue_batch (round 3): the per-query image index lies frame
    after frame like the queries; per-image clustering and per-match cameras apply to every frame of the batch
    (CLUSTER_MEAN_SHIFT_CPU.hpp:189-195, POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:213-237) -- every frame's objects
    are bit for bit what mh_frame_enqueue gives it alone."""
    import torch
    s = scene
    db, fr = s["db"], s["fr"]
    dev = torch.device("cuda:0")
    Q = len(fr.uv)
    # a second and third frame: the same rig, the image index of half of image 1's keypoints flipped / all on image 0
    rng = np.random.default_rng(4)
    img2 = fr.image.copy()
    flip = np.nonzero(fr.image == 1)[0]
    img2[rng.choice(flip, len(flip) // 2, replace=False)] = 0
    imgs = [fr.image, img2, np.zeros_like(fr.image)]
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    c.reserve(3 * Q)
    prm = capi.default_frame_params()
    alone = []
    for f, im in enumerate(imgs):
        q_img = torch.from_numpy(np.ascontiguousarray(im)).to(dev)
        c.frame_set_images(q_img.data_ptr(), fr.Ks, fr.cams)
        qd, uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
        c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, fr.Ks[0], fr.cams[0], prm, 30 + f)
        alone.append(c.frame_fetch())
    assert alone[0][1][1] != alone[2][1][1]          # the image index matters: other clusters with one image
    q_img = torch.from_numpy(np.concatenate(imgs)).to(dev)
    c.frame_set_images(q_img.data_ptr(), fr.Ks, fr.cams)
    qd = torch.from_numpy(np.concatenate([fr.desc] * 3)).to(dev)
    uv = torch.from_numpy(np.concatenate([fr.uv] * 3)).to(dev)
    c.frame_enqueue_batch(qd.data_ptr(), uv.data_ptr(), Q, 3, fr.Ks[0], fr.cams[0], prm, [30, 31, 32])
    for f in range(3):
        o, cnt = c.frame_fetch_slot(f)
        a, ac = alone[f]
        assert np.array_equal(cnt, ac), (f, cnt, ac)
        assert len(o) == len(a) >= 1 and np.array_equal(o["model"], a["model"])
        assert np.array_equal(o["pose"].view(np.uint32), a["pose"].view(np.uint32))
        assert np.array_equal(o["score"].view(np.uint32), a["score"].view(np.uint32))
    c.frame_set_images(0)
    c.close()


def test_batches_take_the_image_index_frame_after_frame_and_still_refuse_one_depth_map(scene):
    """A depth map belongs to ONE frame: mh_frame_enqueue_batch says so (mh_frame_set_depth_image_batch hands in one
    per frame) instead of applying the first frame's map to all."""
    import torch
    s = scene
    db, fr = s["db"], s["fr"]
    dev = torch.device("cuda:0")
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    Q = len(fr.uv)
    c.reserve(2 * Q)
    qd = torch.from_numpy(np.concatenate([fr.desc, fr.desc])).to(dev)
    uv = torch.from_numpy(np.concatenate([fr.uv, fr.uv])).to(dev)
    depth = torch.zeros(480 * 640 * 4, dtype=torch.float32, device=dev)
    c.frame_set_depth_image(depth.data_ptr(), 0, 640, 480, 1)
    with pytest.raises(capi.MhError):
        c.frame_enqueue_batch(qd.data_ptr(), uv.data_ptr(), Q, 2, fr.Ks[0], fr.cams[0], capi.default_frame_params(), [1, 2])
    c.frame_set_depth_image(0, 0, 640, 480, 0)
    c.frame_enqueue_batch(qd.data_ptr(), uv.data_ptr(), Q, 2, fr.Ks[0], fr.cams[0], capi.default_frame_params(), [1, 2])
    o0, _ = c.frame_fetch_slot(0)
    o1, _ = c.frame_fetch_slot(1)
    assert len(o0) == len(o1)   # the same frame twice (different seeds): the same detections
    c.close()
