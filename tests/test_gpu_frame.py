"""End-to-end frame parity on the GPU: MATCH -> CLUSTER -> POSE -> FILTER -> POSE2 ->
FILTER2 through mh_frame_enqueue vs the CPU oracle pipeline on the same frames."""
import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY


def _mean_reproj(pose, uv, xyz):
    p = orclib.project(pose, xyz, K, CAM0)
    return float(np.sqrt(((p - uv) ** 2).sum(1)).mean())


@pytest.fixture(scope="module")
def world():
    import torch
    from moped_amd.pipeline import FramePipeline, ShardedDB
    db = synth.make_db(20, 5000)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=2, max_queries=3000)
    dbn = orclib.normalize(db.desc)
    yield db, dbn, pipe, torch
    pipe.close()


@pytest.mark.parametrize("seed,n_vis", [(0, 2), (1, 2), (2, 5), (3, 10), (4, 1)])
def test_frame_matches_oracle_pipeline(world, seed, n_vis):
    db, dbn, pipe, torch = world
    fr = synth.make_frame(db, n_vis=n_vis, seed=seed)
    dev = torch.device("cuda:0")
    q_desc = torch.from_numpy(fr.desc).to(dev)
    q_uv = torch.from_numpy(fr.uv).to(dev)
    slot = seed % 2
    pipe.enqueue(slot, q_desc, q_uv, seed=seed + 11)
    objs, counts = pipe.fetch(slot)

    qn = orclib.normalize(fr.desc)
    # A1 in place, bit exact
    assert np.array_equal(q_desc.cpu().numpy().view(np.uint32), qn.view(np.uint32))
    idx, d1, d2 = orclib.match_2nn(dbn, qn)
    om, op, osc, oc, oinl = orclib.frame_rest_inliers(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0,
                                                      n_threads=4, seed=seed)
    # index-exact stages: accepted matches and mean-shift clusters
    assert counts[0] == oc[0]
    assert counts[1] == oc[1]
    # same set of detected models (planted objects; FILTER2 removes duplicates)
    assert sorted(objs["model"].tolist()) == sorted(om.tolist())
    assert set(om.tolist()) == set(fr.visible.tolist())
    for m, p, sc, inl in zip(om, op, osc, oinl):
        g = objs[objs["model"] == m][0]
        # the north-star bar as it is stated: mean reprojection error within 1 px of the oracle pose's, over the ORACLE'S
        # inlier set (testAllPoints of its final pose over its final cluster, ...REPROJECTION_CPU.hpp:166-180)
        assert len(inl) >= 7 and np.all(db.model_of[idx[inl]] == m)
        xyz_i, uv_i = db.xyz[idx[inl]], fr.uv[inl]
        ei_o, ei_g = _mean_reproj(p, uv_i, xyz_i), _mean_reproj(g["pose"], uv_i, xyz_i)
        assert ei_g <= ei_o + 1.0, (m, ei_g, ei_o)
        assert ei_g < 1.0
        # and over the generator's own ground truth (the planted, non-outlier points)
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
        rows = rows[db.model_of[fr.src_point[rows]] == m]
        xyz, uv = db.xyz[fr.src_point[rows]], fr.uv[rows]
        e_o, e_g = _mean_reproj(p, uv, xyz), _mean_reproj(g["pose"], uv, xyz)
        assert e_g <= e_o + 1.0, (m, e_g, e_o)      # the north-star bar: within 1 px of the reference pose
        assert e_g < 1.0                            # and close to the planted pose in absolute terms
        assert abs(g["score"] - sc) <= 0.05 * sc    # FILTER2 score of the same object, same points
        j = list(fr.visible).index(m)
        assert np.allclose(g["pose"][4:], fr.poses[j][4:], atol=3e-3)


def test_frame_no_objects_on_clutter(world):
    db, dbn, pipe, torch = world
    fr = synth.make_frame(db, n_vis=0, seed=77)
    dev = torch.device("cuda:0")
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=5)
    objs, counts = pipe.fetch(0)
    assert len(objs) == 0


def test_frame_deterministic_for_seed(world):
    db, dbn, pipe, torch = world
    fr = synth.make_frame(db, n_vis=3, seed=9)
    dev = torch.device("cuda:0")
    res = []
    for slot in (0, 1, 0):
        pipe.enqueue(slot, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=123)
        objs, _ = pipe.fetch(slot)
        res.append(objs)
    assert np.array_equal(res[0]["model"], res[1]["model"]) and np.array_equal(res[0]["model"], res[2]["model"])
    assert np.array_equal(res[0]["pose"].view(np.uint32), res[1]["pose"].view(np.uint32))
    assert np.array_equal(res[0]["pose"].view(np.uint32), res[2]["pose"].view(np.uint32))


def test_frame_timing_api(world):
    db, dbn, pipe, torch = world
    c = pipe.ctxs[0]
    c.enable_timing(True)
    fr = synth.make_frame(db, n_vis=2, seed=1)
    dev = torch.device("cuda:0")
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=1)
    pipe.fetch(0)
    t = c.timing()
    c.enable_timing(False)
    assert t["total_ms"] > 0 and t["match_ms"] > 0
    assert abs(sum(t[k] for k in t if k != "total_ms") - t["total_ms"]) < 0.5 * t["total_ms"] + 0.5


def test_batch_of_frames_equals_the_frames_alone(world):
    """mh_frame_enqueue_batch: one MATCH launch sequence over the queries of B frames, CLUSTER..FILTER2 frame by
    frame -- every frame's objects and counts are bit for bit what mh_frame_enqueue gives it alone."""
    db, dbn, pipe, torch = world
    dev = torch.device("cuda:0")
    frs = [synth.make_frame(db, n_vis=n, seed=40 + i) for i, n in enumerate((2, 5, 1))]
    alone = []
    for i, fr in enumerate(frs):
        pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=70 + i)
        alone.append(pipe.fetch(0))
    qd = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
    uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
    c = pipe.ctxs[1]
    c.reserve(3 * 3000)
    pipe.enqueue_batch(1, qd, uv, 3, [70, 71, 72])
    for f, (objs, counts) in enumerate(pipe.fetch_batch(1, 3)):
        a, ac = alone[f]
        assert np.array_equal(counts, ac) and len(objs) == len(a) >= 1
        assert np.array_equal(objs["model"], a["model"])
        assert np.array_equal(objs["pose"].view(np.uint32), a["pose"].view(np.uint32))
        assert np.array_equal(objs["score"].view(np.uint32), a["score"].view(np.uint32))
    # the batch normalised all three frames' descriptors in place, like the single frames
    assert np.array_equal(qd[:3000].cpu().numpy().view(np.uint32), orclib.normalize(frs[0].desc).view(np.uint32))


def test_randomised_frames_against_the_oracle_pipeline():
    """tests/tools/frame_stress.py, 18 scenes: random DBs, query counts, visible objects, points per object and outlier
    shares -- matches and clusters exact, model sets equal, poses within 1 px of the oracle's, the same frames through
    mh_frame_enqueue_batch bit-identical (600 scenes: profiles/r02_frame_stress.txt)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "frame_stress.py"), "18", "5"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_pose_as_two_launches_gives_the_bits_of_the_one_launch_form(world):
    """mh_pose_set_split: the hypotheses of every task in pose_kernel and the refines in pose_refine_kernel (one wavefront
    per task; the default of the frame paths since round 4) against the workgroup that found a task's winner refining it
    itself -- frames alone and the same frames as one batch, 0 to 10 visible objects: the same objects bit for bit (the
    refine is the same code on the same points, in LDS or, for clusters past the refine's cache, in global scratch)."""
    db, dbn, pipe, torch = world
    from moped_amd.pipeline import FramePipeline, ShardedDB
    dev = torch.device("cuda:0")
    frs = [synth.make_frame(db, n_vis=n, seed=300 + i) for i, n in enumerate((2, 5, 10, 0, 1, 3, 2, 7))]
    big = synth.make_frame(db, n_vis=1, seed=77, pts_per_obj=420, outlier_frac=0.0)     # one cluster past the 320-point cache
    B = len(frs)
    p2 = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=2, max_queries=B * 3000, batch=B)
    qd = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
    uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
    res = {}
    for split in (0, 1):
        for c in p2.ctxs:
            c.pose_set_split(split)
        out = []
        for i, fr in enumerate(frs + [big]):
            p2.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=50 + i)
            out.append(p2.fetch(0))
        qd.copy_(torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev))
        torch.cuda.synchronize()   # the copy runs on torch's stream, the frames on the library's own (non-blocking) streams
        p2.enqueue_batch(1, qd, uv, B, [50 + i for i in range(B)])
        out += p2.fetch_batch(1, B)
        res[split] = out
    n_obj = 0
    for (o0, c0), (o1, c1) in zip(res[0], res[1]):
        assert np.array_equal(c0, c1) and np.array_equal(o0["model"], o1["model"])
        assert np.array_equal(o0["pose"].view(np.uint32), o1["pose"].view(np.uint32))
        assert np.array_equal(o0["score"].view(np.uint32), o1["score"].view(np.uint32))
        n_obj += len(o1)
    assert n_obj >= 2 * 30 and len(res[1][B][0]) == 1     # the 420-point object is found (its cluster does not fit the cache)
    p2.close()


def test_frame_with_more_object_slots_than_filter_keeps_in_lds():
    """72 visible objects: >= 288 (cluster, replica) object slots after POSE, more than the 256 whose model / list / score
    the fused FILTER tail holds in LDS (csrc/filter_dev.h, FL_SLOTS) -- the slots past them take the path that reads the
    arrays.  Same objects as the oracle pipeline, scores within its 5%, every planted object found."""
    import torch
    from moped_amd.pipeline import FramePipeline, ShardedDB
    db = synth.make_db(80, 400, seed=5)
    fr = synth.make_frame(db, n_vis=72, seed=9, Q=6000, pts_per_obj=60)
    Q = len(fr.desc)
    dev = torch.device("cuda:0")
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=Q)
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=21)
    objs, counts = pipe.fetch(0)
    pipe.close()
    assert counts[2] > 256          # object slots in use after POSE
    idx, d1, d2 = orclib.match_2nn(orclib.normalize(db.desc), orclib.normalize(fr.desc))
    om, op, osc, oc, _ = orclib.frame_rest_inliers(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0,
                                                   n_threads=4, seed=9)
    assert counts[0] == oc[0] and counts[1] == oc[1]
    assert sorted(objs["model"].tolist()) == sorted(om.tolist())
    assert set(fr.visible.tolist()) <= set(objs["model"].tolist())
    for m, sc in zip(om, osc):
        g = objs[objs["model"] == m][0]
        assert abs(g["score"] - sc) <= 0.05 * sc, (m, g["score"], sc)


@pytest.mark.parametrize("seed", [0, 1, 3, 4])
def test_pose_survivor_stays_when_pose2_finds_nothing(world, seed):
    """Eight clean matches per object: POSE finds it (8 > MinNPtsObject 6), FILTER keeps it, POSE2 cannot succeed (it needs
    MORE than 8 inliers) -- the reference's list still holds the POSE object (POSE2 appends, ...REPROJECTION_CPU.hpp:299)
    and FILTER2 scores every object (FILTER_PROJECTION_CPU.hpp:96): 8 >= MinPoints 7, score ~7.8 >= 3, it stays.
    The fused FILTER2 tail must score those kept slots too (it erased them): fused == stand-alone FILTER
    launches (stage timing keeps the steps apart) == oracle, alone and in a batch."""
    db, dbn, pipe, torch = world
    dev = torch.device("cuda:0")
    fr = synth.make_frame(db, n_vis=2, seed=seed, pts_per_obj=8, outlier_frac=0.0, pix_noise=0.2)
    idx, d1, d2 = orclib.match_2nn(dbn, orclib.normalize(fr.desc))
    om, op, osc, oc, _ = orclib.frame_rest_inliers(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0,
                                                   n_threads=1, seed=seed)
    assert len(om) >= 1 and oc[3] == len(om)
    res = []
    c = pipe.ctxs[0]
    for timing in (False, True):
        c.enable_timing(timing)
        pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=seed + 5)
        res.append(pipe.fetch(0))
    c.enable_timing(False)
    (fo, fc), (uo, uc) = res
    assert np.array_equal(fc, uc) and fc[0] == oc[0] and fc[1] == oc[1]
    assert sorted(fo["model"].tolist()) == sorted(uo["model"].tolist()) == sorted(om.tolist())
    assert np.array_equal(fo["pose"].view(np.uint32), uo["pose"].view(np.uint32))
    assert np.array_equal(fo["score"].view(np.uint32), uo["score"].view(np.uint32))
    for m, sc in zip(om, osc):
        g = fo[fo["model"] == m][0]
        assert abs(g["score"] - sc) <= 0.05 * sc and g["n_points"] >= 7
    # the same frame twice in a merged batch (the refine launch closes the frames)
    qd = torch.cat([torch.from_numpy(fr.desc)] * 2).to(dev)
    uv = torch.cat([torch.from_numpy(fr.uv)] * 2).to(dev)
    pipe.ctxs[1].reserve(2 * 3000)
    pipe.enqueue_batch(1, qd, uv, 2, [seed + 5, seed + 5])
    for objs, counts in pipe.fetch_batch(1, 2):
        assert np.array_equal(counts, fc) and np.array_equal(objs["model"], fo["model"])
        assert np.array_equal(objs["pose"].view(np.uint32), fo["pose"].view(np.uint32))
        assert np.array_equal(objs["score"].view(np.uint32), fo["score"].view(np.uint32))
