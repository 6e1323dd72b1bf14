"""GPU parity tests, per step, through the C ABI (libmoped_hip.so) against the CPU
oracle on the same seeded inputs.  Integer/index results must be bit-exact."""
import os
import zlib

import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu

K = synth.K_DEFAULT
CAM0 = synth.CAM_IDENTITY


# ---------------------------------------------------------------------------- A1
def test_normalize_bit_exact(ctx, sift):
    base, _, _ = sift
    rng = np.random.default_rng(1)
    raw = (base[:777] * rng.uniform(0.2, 300.0, size=(777, 1))).astype(np.float32)
    got = ctx.normalize(raw)
    want = orclib.normalize(raw)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_normalize_ragged_sizes(ctx, sift):
    base, _, _ = sift
    for n in (1, 63, 64, 65, 130):
        got = ctx.normalize(base[:n] * 7.0)
        want = orclib.normalize(base[:n] * 7.0)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---------------------------------------------------------------------------- A3
def _match_case(ctx, db, q, ratio=0.8):
    dbn = orclib.normalize(db.desc)
    qn = orclib.normalize(q)
    ctx.db_upload(dbn, db.model_of, db.xyz, db.n_models)
    acc, raw, d1, d2 = ctx.match(qn, ratio)
    oi, od1, od2 = orclib.match_2nn(dbn, qn)
    assert np.array_equal(raw, oi)
    assert np.array_equal(d1.view(np.uint32), od1.view(np.uint32))
    assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
    want_acc = np.where(od1 / od2 < np.float32(ratio), oi, -1).astype(np.int32)
    assert np.array_equal(acc, want_acc)
    return acc, raw, d1, d2


@pytest.mark.parametrize("n_models,ppm", [(1, 1000), (2, 5000), (20, 5000)])
def test_match_bit_exact_vs_oracle(ctx, sift, n_models, ppm):
    base, _, _ = sift
    db = synth.make_db(n_models, ppm)
    _match_case(ctx, db, base)


def test_normalize_match_equals_the_two_calls(ctx, sift):
    """mh_normalize_match (what the MATCH plugins call): the normalised descriptors and the results of mh_normalize
    followed by mh_match, bit for bit."""
    base, _, _ = sift
    db = synth.make_db(6, 2000)
    ctx.db_upload(ctx.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    q = np.ascontiguousarray(base[:1777] * np.float32(37.5))
    qn = ctx.normalize(q)
    want = ctx.match(qn, 0.8)
    got_q, *got = ctx.normalize_match(q, 0.8)
    assert np.array_equal(got_q.view(np.uint32), qn.view(np.uint32))
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("tag,n_models,ppm", [("1k", 1, 1000), ("10k", 2, 5000), ("100k", 20, 5000)])
def test_match_vs_reference_ann_golden(ctx, sift, tag, n_models, ppm):
    """Against the reference's own ANN kd-tree at eps=0 (tests/golden): identical
    accepted set and indices; raw NN may differ only on exact near-ties."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"match_ann_{tag}.npz"))
    base, _, _ = sift
    db = synth.make_db(n_models, ppm)
    acc, raw, d1, d2 = _match_case(ctx, db, base)
    gi, gd = g["idx_eps0"], g["dist_eps0"]
    g_acc = np.where(gd[:, 0] / gd[:, 1] < np.float32(0.8), gi[:, 0], -1)
    borderline = np.abs(gd[:, 0] / gd[:, 1] - 0.8) < 1e-6
    assert borderline.sum() == 0
    assert np.array_equal(acc, g_acc)
    differ = raw != gi[:, 0]
    # any raw-NN disagreement must be a numerical tie between the two candidates
    assert np.all(np.abs(gd[differ, 0] - gd[differ, 1]) <= 1e-5 * gd[differ, 1])
    assert np.allclose(d1, gd[:, 0], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("Q,N", [(1, 1), (1, 2), (3, 127), (129, 128), (200, 129), (5, 1000), (1025, 257)])
def test_match_ragged_shapes(ctx, sift, Q, N):
    base, _, _ = sift
    db = synth.make_db(1, N, seed=7)
    rng = np.random.default_rng(Q * 1000 + N)
    q = base[rng.integers(0, base.shape[0], Q)]
    _match_case(ctx, db, q)


def test_match_duplicates_and_ties(ctx, sift):
    """Exact duplicates in the DB: equal best/second best -> ratio 1 -> rejected;
    the raw index is the lowest row (canonical tie rule)."""
    base, _, _ = sift
    d = orclib.normalize(base[:300])
    dbd = np.concatenate([d, d[:50]])
    ctx.db_upload(dbd, np.zeros(len(dbd), np.int32), np.zeros((len(dbd), 3), np.float32), 1)
    acc, raw, d1, d2 = ctx.match(d[:50], 0.8)
    assert np.array_equal(raw, np.arange(50))
    assert np.all(acc == -1)
    oi, od1, od2 = orclib.match_2nn(dbd, d[:50])
    assert np.array_equal(raw, oi) and np.array_equal(d1, od1) and np.array_equal(d2, od2)


def test_match_sharded_merge_equals_global(ctx, sift):
    """Model-sharded DB: per-shard local top-2 + merge == search over the whole DB."""
    import torch
    base, _, _ = sift
    db = synth.make_db(8, 700, seed=3)
    dbn = orclib.normalize(db.desc)
    qn = orclib.normalize(base[:1500])
    Q = qn.shape[0]
    oi, od1, od2 = orclib.match_2nn(dbn, qn)
    dev = torch.device("cuda:0")
    tq = torch.from_numpy(qn).to(dev)
    qnorm = torch.from_numpy(orclib.row_norms(qn)).to(dev)
    S = 4
    idx_s = torch.empty((S, Q), dtype=torch.int32, device=dev)
    d1_s = torch.empty((S, Q), dtype=torch.float32, device=dev)
    d2_s = torch.empty((S, Q), dtype=torch.float32, device=dev)
    rows = db.n // S
    for s in range(S):
        lo, hi = s * rows, (s + 1) * rows if s < S - 1 else db.n
        ctx.db_upload(dbn[lo:hi], db.model_of[lo:hi], db.xyz[lo:hi], db.n_models, index_base=lo)
        ctx.match_local_dev(tq.data_ptr(), qnorm.data_ptr(), Q, idx_s[s].data_ptr(),
                            d1_s[s].data_ptr(), d2_s[s].data_ptr())
        ctx.synchronize()
    gi = torch.empty(Q, dtype=torch.int32, device=dev)
    g1 = torch.empty(Q, dtype=torch.float32, device=dev)
    g2 = torch.empty(Q, dtype=torch.float32, device=dev)
    ctx.match_merge_dev(idx_s.data_ptr(), d1_s.data_ptr(), d2_s.data_ptr(), S, Q,
                        gi.data_ptr(), g1.data_ptr(), g2.data_ptr())
    ctx.synchronize()
    assert np.array_equal(gi.cpu().numpy(), oi)
    assert np.array_equal(g1.cpu().numpy(), od1)
    assert np.array_equal(g2.cpu().numpy(), od2)
    # and the CPU merge restatement agrees
    mi, m1, m2 = orclib.match_merge(idx_s.cpu().numpy(), d1_s.cpu().numpy(), d2_s.cpu().numpy())
    assert np.array_equal(mi, oi) and np.array_equal(m1, od1) and np.array_equal(m2, od2)


# ---------------------------------------------------------------------------- A6
def _ms_points(rng, kind, n):
    if kind == "blobs":
        c = rng.uniform([50, 50], [590, 430], size=(4, 2))
        p = c[rng.integers(0, 4, n)] + rng.normal(0, 25, size=(n, 2))
    elif kind == "uniform":
        p = rng.uniform([0, 0], [640, 480], size=(n, 2))
    elif kind == "chain":  # points strung out so merges chain across iterations
        t = np.sort(rng.uniform(0, 1, n))
        p = np.stack([40 + 560 * t, 240 + 30 * np.sin(9 * t)], 1) + rng.normal(0, 3, (n, 2))
    elif kind == "tight":
        p = rng.normal([320, 240], 6, size=(n, 2))
    elif kind == "grid":
        g = np.stack(np.meshgrid(np.arange(8) * 18.0, np.arange(8) * 18.0), -1).reshape(-1, 2)
        p = g[rng.permutation(len(g))[:n]] + 100
    else:
        raise ValueError(kind)
    return p.astype(np.float32)


@pytest.mark.parametrize("kind", ["blobs", "uniform", "chain", "tight", "grid"])
@pytest.mark.parametrize("n", [1, 6, 7, 40, 64, 65, 150, 400])
def test_meanshift_partition_and_order_exact(ctx, kind, n):
    rng = np.random.default_rng(zlib.crc32(f"{kind}-{n}".encode()))
    if kind == "grid":
        n = min(n, 64)
    pts = _ms_points(rng, kind, n)
    for (radius, merge, min_pts) in ((200.0, 20.0, 7), (60.0, 20.0, 3), (150.0, 35.0, 1)):
        want, _ = orclib.meanshift(pts, radius, merge, min_pts, 100)
        got, label = ctx.meanshift(pts, radius, merge, min_pts, 100)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)  # same members, same (splice) order


def test_meanshift_adversarial_orderings(ctx):
    """Same point set, different input orders: the reference's result depends on
    order; the kernel must follow the oracle in every order."""
    rng = np.random.default_rng(99)
    base = _ms_points(rng, "blobs", 120)
    for trial in range(6):
        pts = base[rng.permutation(len(base))]
        want, _ = orclib.meanshift(pts, 200.0, 20.0, 7, 100)
        got, _ = ctx.meanshift(pts, 200.0, 20.0, 7, 100)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)


def test_meanshift_3d_and_iteration_cap(ctx):
    rng = np.random.default_rng(5)
    pts = rng.normal(0, 0.05, size=(90, 3)).astype(np.float32)
    pts[45:] += 0.4
    want, _ = orclib.meanshift(pts, 0.2, 0.05, 5, 100)
    got, _ = ctx.meanshift(pts, 0.2, 0.05, 5, 100)
    assert [list(a) for a in got] == [list(b) for b in want]
    p2 = _ms_points(rng, "chain", 200)
    for cap in (1, 2, 3):
        want, _ = orclib.meanshift(p2, 200.0, 20.0, 1, cap)
        got, _ = ctx.meanshift(p2, 200.0, 20.0, 1, cap)
        assert [list(a) for a in got] == [list(b) for b in want]


def test_meanshift_batch_equals_per_problem_calls(ctx):
    """mh_meanshift_batch (the per-model loop of CLUSTER_MEAN_SHIFT_CPU::process in one launch):
    every problem, empty ones included, gets the oracle's clusters in the oracle's order."""
    rng = np.random.default_rng(2024)
    problems = [_ms_points(rng, kind, n) for kind, n in
                (("blobs", 150), ("uniform", 0), ("chain", 90), ("uniform", 5), ("blobs", 400), ("grid", 49),
                 ("uniform", 0))]
    got = ctx.meanshift_batch(problems, 200.0, 20.0, 7, 100)
    assert len(got) == len(problems)
    for pts, (clusters, label) in zip(problems, got):
        want = orclib.meanshift(pts, 200.0, 20.0, 7, 100)[0] if len(pts) else []
        assert [list(a) for a in clusters] == [list(b) for b in want]
        want_label = np.full(len(pts), -1, np.int32)
        for c, members in enumerate(want):
            want_label[members] = c
        assert np.array_equal(label, want_label)
    assert ctx.meanshift_batch([], 200.0, 20.0, 7, 100) == []


# ---------------------------------------------------------------------------- A12
def _planted(rng, n, cam=CAM0, noise=0.5, outliers=0.0):
    xyz = ((rng.random((n, 3)) - 0.5) * [0.1, 0.1, 0.2]).astype(np.float32)
    q = synth.random_quat(rng)
    t = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(0.5, 1.0)])
    pose = np.concatenate([q, t]).astype(np.float32)
    uv = orclib.project(pose, xyz, K, cam)
    uv = (uv + rng.uniform(-noise, noise, uv.shape)).astype(np.float32)
    bad = rng.random(n) < outliers
    uv[bad] = rng.uniform([0, 0], [640, 480], size=(int(bad.sum()), 2)).astype(np.float32)
    return pose, uv, xyz, bad


def test_project_test_bit_exact(ctx):
    rng = np.random.default_rng(11)
    cam = np.concatenate([synth.random_quat(rng), [0.05, -0.02, 0.1]]).astype(np.float32)
    for c in (CAM0, cam):
        pose, uv, xyz, _ = _planted(rng, 500, c, noise=3.0)
        # include points behind / at the camera plane
        xyz[:5] = [[0, 0, -5], [0, 0, -0.8], [0.01, 0.01, -0.6], [0, 0, -0.75], [0.3, 0.3, -0.7]]
        corr = capi.pack_corr(uv, xyz)
        for thr in (10.0, 5.0, 4096.0):
            cnt, inl, e2 = ctx.project_test(pose, corr, K, c, thr)
            ocnt, oinl = orclib.test_all_points(pose, uv, xyz, K, c, thr)
            assert cnt == ocnt
            assert np.array_equal(inl, oinl)
        ouv = orclib.project(pose, xyz, K, c)
        d = ouv - uv
        want_e2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
        fin = np.isfinite(want_e2)
        assert np.array_equal(e2[fin].view(np.uint32), want_e2[fin].view(np.uint32))


# ---------------------------------------------------------------------------- A8-A13
def _mean_reproj(pose, uv, xyz, cam):
    p = orclib.project(pose, xyz, K, cam)
    return float(np.sqrt(((p - uv) ** 2).sum(1)).mean())


@pytest.mark.parametrize("seed", range(6))
def test_pose_ransac_matches_oracle_pose(ctx, seed):
    """Final-pose parity: for every cluster the oracle's RANSAC+LM
    solves, the HIP RANSAC reports a pose whose mean reprojection error over the
    ORACLE's inlier set is within 1 px of the oracle pose's."""
    rng = np.random.default_rng(100 + seed)
    clusters, off = [], [0]
    truth = []
    for c in range(4):
        n = int(rng.integers(12, 160))
        pose, uv, xyz, bad = _planted(rng, n, CAM0, noise=0.5, outliers=0.25)
        clusters.append((uv, xyz))
        truth.append((pose, bad))
        off.append(off[-1] + n)
    uv_all = np.concatenate([c[0] for c in clusters])
    xyz_all = np.concatenate([c[1] for c in clusters])
    prm = capi.make_pose_params(1024, 4, 5, 6, 10.0, 10, 10)
    out = ctx.pose_ransac(capi.pack_corr(uv_all, xyz_all), np.array(off, np.int32), K, CAM0, prm, seed=seed + 1)
    by_cluster = {}
    for o in out:
        by_cluster.setdefault(int(o["cluster"]), []).append(o)
    for c, (uv, xyz) in enumerate(clusters):
        ok, opose = orclib.ransac(uv, xyz, K, CAM0, orclib.POSE1, seed=seed)
        if not ok:
            continue
        _, oinl = orclib.test_all_points(opose, uv, xyz, K, CAM0, 10.0)
        assert c in by_cluster, f"cluster {c}: oracle found an object, HIP path did not"
        e_or = _mean_reproj(opose, uv[oinl], xyz[oinl], CAM0)
        best = min(_mean_reproj(o["pose"], uv[oinl], xyz[oinl], CAM0) for o in by_cluster[c])
        assert best <= e_or + 1.0, (best, e_or)
        # every replica that reports an object must itself be a good pose
        for o in by_cluster[c]:
            assert _mean_reproj(o["pose"], uv[oinl], xyz[oinl], CAM0) <= e_or + 1.0
            assert o["n_inliers"] > 6


def test_pose_ransac_rejects_small_and_degenerate(ctx):
    rng = np.random.default_rng(7)
    prm = capi.make_pose_params(1024, 4, 5, 6, 10.0, 10, 10)
    # fewer distinct image points than NPtsAlign -> randSample fails -> no object
    pose, uv, xyz, _ = _planted(rng, 4, CAM0)
    out = ctx.pose_ransac(capi.pack_corr(uv, xyz), np.array([0, 4], np.int32), K, CAM0, prm)
    assert len(out) == 0
    # 30 correspondences that all share 3 image points
    pose, uv, xyz, _ = _planted(rng, 30, CAM0)
    uv[:] = uv[np.arange(30) % 3]
    out = ctx.pose_ransac(capi.pack_corr(uv, xyz), np.array([0, 30], np.int32), K, CAM0, prm)
    assert len(out) == 0
    # pure clutter: no pose may gather > MinNPtsObject inliers
    uv = rng.uniform([0, 0], [640, 480], size=(40, 2)).astype(np.float32)
    xyz = ((rng.random((40, 3)) - 0.5) * [0.1, 0.1, 0.2]).astype(np.float32)
    out = ctx.pose_ransac(capi.pack_corr(uv, xyz), np.array([0, 40], np.int32), K, CAM0, prm)
    for o in out:
        cnt, _ = orclib.test_all_points(o["pose"], uv, xyz, K, CAM0, 10.0)
        assert cnt > 6  # whatever is reported really has the inliers it claims


def test_pose_ransac_nonidentity_camera(ctx):
    rng = np.random.default_rng(21)
    cam = np.concatenate([synth.random_quat(rng), [0.05, -0.02, 0.1]]).astype(np.float32)
    xyz = ((rng.random((80, 3)) - 0.5) * [0.1, 0.1, 0.2]).astype(np.float32)
    # pose expressed in the world frame; project through the camera pose
    q = synth.random_quat(rng)
    pose_cam = np.concatenate([q, [0.03, -0.04, 0.7]]).astype(np.float32)
    uv_id = orclib.project(pose_cam, xyz, K, CAM0)
    # world pose = cam o pose_cam
    Rc, R = synth.quat_to_R(cam[:4]), synth.quat_to_R(pose_cam[:4])
    Rw, tw = Rc @ R, Rc @ pose_cam[4:] + cam[4:]
    uv = (uv_id + rng.uniform(-0.5, 0.5, uv_id.shape)).astype(np.float32)
    prm = capi.make_pose_params(1024, 2, 5, 6, 10.0, 10, 10)
    out = ctx.pose_ransac(capi.pack_corr(uv, xyz), np.array([0, 80], np.int32), K, cam, prm)
    assert len(out) >= 1
    for o in out:
        assert _mean_reproj(o["pose"], uv, xyz, cam) < 1.0
        assert np.allclose(o["pose"][4:], tw, atol=5e-3)


# ---------------------------------------------------------------------------- N1
def test_filter_matches_oracle(ctx):
    rng = np.random.default_rng(31)
    n_models = 5
    uvs, xyzs, off = [], [], [0]
    objs_m, objs_p = [], []
    for m in range(n_models):
        n = int(rng.integers(0, 120)) if m != 2 else 0
        if n:
            pose, uv, xyz, _ = _planted(rng, n, CAM0, noise=0.7, outliers=0.3)
            uvs.append(uv)
            xyzs.append(xyz)
            for r in range(int(rng.integers(1, 5))):   # duplicate / perturbed hypotheses
                p = pose.copy()
                p[4:] += rng.normal(0, 0.002 * r, 3).astype(np.float32)
                objs_m.append(m)
                objs_p.append(p)
            if rng.random() < 0.5:                      # a wrong pose for the same model
                objs_m.append(m)
                objs_p.append(np.concatenate([synth.random_quat(rng), [0, 0, 0.8]]).astype(np.float32))
        off.append(off[-1] + n)
    uv = np.concatenate(uvs)
    xyz = np.concatenate(xyzs)
    uv[10] = uv[3]  # two keypoints at the same image location (bestPoints map key collision)
    order = np.argsort(np.array(objs_m), kind="stable")
    obj_m = np.array(objs_m, np.int32)[order]
    obj_p = np.array(objs_p, np.float32)[order]
    for (mp, fd, ms) in ((5, 4096.0, 2.0), (7, 4096.0, 3.0), (1, 25.0, 0.0)):
        s_o, k_o, ord_o, cl_o = orclib.filter_projection(uv, xyz, np.array(off, np.int32), obj_m, obj_p, K, CAM0, mp, fd, ms)
        s_g, k_g, ord_g, cl_g = ctx.filter(capi.pack_corr(uv, xyz), np.array(off, np.int32), obj_m, obj_p, K, CAM0, mp, fd, ms)
        assert np.array_equal(k_g, k_o)
        assert np.array_equal(ord_g, ord_o)
        assert np.array_equal(s_g.view(np.uint32), s_o.view(np.uint32))
        assert len(cl_g) == len(cl_o)
        for a, b in zip(cl_g, cl_o):
            assert np.array_equal(a, b)


def test_meanshift_stress_all_walk_variants_and_real_layouts(ctx):
    """The list walk has register variants for up to 128 / 256 / 384 / 512 / 768 / 1024 canopies and an LDS
    variant beyond; closeness graphs that are far from transitive (smooth fields of means: real keypoint
    layouts, strings of points) take the outward-member and relaxation paths, deep merge forests the
    list-order fallback of the fold.  Everything must stay exact: same clusters, same member order."""
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sift_ref_frames.npz"))
    real = np.concatenate([gold["xy0"], gold["xy3"] + np.float32(0.37)])
    rng = np.random.default_rng(2024)
    cases = []
    for n in (100, 129, 257, 385, 513, 600, 769, 1025, 1100, 1500, 2048):
        kinds = ("uniform", "blobs") if n > 1100 else ("uniform", "blobs", "chain", "real")
        for kind in kinds:
            if kind == "real":
                pts = real[np.sort(rng.choice(len(real), min(n, len(real)), replace=False))]
            else:
                pts = _ms_points(rng, kind, n)
            cases.append((f"{kind}-{n}", np.ascontiguousarray(pts, np.float32), (200.0, 20.0, 7)))
    # strings of points 15 px apart with a radius that sees only the point itself: closeness is a chain,
    # merges cascade along it (deep forests, many iterations)
    for n in (40, 90, 300):
        line = np.stack([20 + 15.0 * np.arange(n) % 600, 100 + 40.0 * (15 * np.arange(n) // 600)], 1).astype(np.float32)
        cases.append((f"line-{n}", line, (10.0, 20.0, 2)))
        cases.append((f"line-shuffled-{n}", line[rng.permutation(n)], (10.0, 20.0, 2)))
    for name, pts, (radius, merge, min_pts) in cases:
        want, _ = orclib.meanshift(pts, radius, merge, min_pts, 100)
        got, _ = ctx.meanshift(pts, radius, merge, min_pts, 100)
        assert len(got) == len(want), name
        for a, b in zip(got, want):
            assert np.array_equal(a, b), name
