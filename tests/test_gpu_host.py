"""The C++ host side: the STEP plugin classes (moped_amd/host/*.hpp) wired into a
MopedPipeline exactly like src/config.hpp does, driven by moped_hip_test, must give
the same frame result as the oracle pipeline."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orclib
from moped_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "moped_amd", "host")
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY


def test_plugin_headers_are_gnu98_clean():
    subprocess.check_call(["make", "-s", "-C", HOST, "check98"])


@pytest.mark.gpu
def test_step_plugins_through_pipeline(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_scene
    subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_test"])
    db = synth.make_db(6, 2000)
    fr = synth.make_frame(db, n_vis=2, seed=5, Q=1500, pts_per_obj=120)
    scene = str(tmp_path / "scene.bin")
    dump_scene.dump(scene, db, fr)
    out = subprocess.check_output([os.path.join(HOST, "moped_hip_test"), scene, "3"], text=True)
    objs, counts = [], None
    for line in out.splitlines():
        w = line.split()
        if w[0] == "OBJ":
            objs.append((int(w[1].replace("model", "")), np.array([float(x) for x in w[5:9] + w[2:5]], np.float32), float(w[9])))
        if w[0] == "MATCHES":
            counts = (int(w[1]), int(w[3]))
    dbn, qn = orclib.normalize(db.desc), orclib.normalize(fr.desc)
    idx, d1, d2 = orclib.match_2nn(dbn, qn)
    om, op, osc, oc, oinl = orclib.frame_rest_inliers(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0, n_threads=2, seed=1)
    assert counts == (int(oc[0]), int(oc[1]))          # matches and clusters: index-exact stages
    assert sorted(m for m, _, _ in objs) == sorted(om.tolist())
    for m, pose, score in objs:
        j = list(om).index(m)
        # the pose bar over the oracle's inlier set (north_star), then over the planted points
        inl = oinl[j]
        ei_g = np.sqrt(((orclib.project(pose, db.xyz[idx[inl]], K, CAM0) - fr.uv[inl]) ** 2).sum(1)).mean()
        ei_o = np.sqrt(((orclib.project(op[j], db.xyz[idx[inl]], K, CAM0) - fr.uv[inl]) ** 2).sum(1)).mean()
        assert len(inl) >= 7 and ei_g <= ei_o + 1.0
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
        rows = rows[db.model_of[fr.src_point[rows]] == m]
        xyz, uv = db.xyz[fr.src_point[rows]], fr.uv[rows]
        e_g = np.sqrt(((orclib.project(pose, xyz, K, CAM0) - uv) ** 2).sum(1)).mean()
        e_o = np.sqrt(((orclib.project(op[j], xyz, K, CAM0) - uv) ** 2).sum(1)).mean()
        assert e_g <= e_o + 1.0
        assert abs(score - osc[j]) <= 0.05 * osc[j]


@pytest.mark.gpu
def test_feat_sift_plugin_through_pipeline(tmp_path):
    """FEAT_SIFT_HIP in the "SIFT" slot (config.hpp:69): the features it appends to
    detectedFeatures are the C ABI's, in the same order."""
    from moped_amd import capi
    subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_test"])
    gold = np.load(os.path.join(ROOT, "tests", "golden", "sift_ref_frames.npz"))
    gray = gold["gray0"]
    pgm = tmp_path / "frame.pgm"
    with open(pgm, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (gray.shape[1], gray.shape[0]))
        f.write(gray.tobytes())
    out = subprocess.check_output([os.path.join(HOST, "moped_hip_test"), "--sift", str(pgm)], text=True)
    rows = [l.split() for l in out.splitlines()]
    n = int(rows[0][1])
    kp = np.array([[float(x) for x in r[2:5]] for r in rows[1:]], np.float64)
    assert rows[0][0] == "KEYPOINTS" and n == len(kp) and all(r[1] == "0" for r in rows[1:])
    c = capi.Context(0)
    xy, so, desc = c.sift(gray)
    c.close()
    assert n == len(xy) and abs(n - len(gold["xy0"])) <= 2
    assert np.abs(kp[:, :2] - xy).max() < 1e-5
    assert np.abs(kp[:, 2] - (desc.astype(np.float64) * np.arange(1, 129)).sum(1)).max() < 1e-3


@pytest.mark.gpu
def test_step_plugins_two_cameras_among_maps(tmp_path):
    """FrameData::images = [camera 0, depth map, distance map, camera 1] (a moped3d frame lists its maps as Images too,
    moped3d/libmoped/src/depthfill/DEPTH_NO_FILL_CPU.hpp:88-96): CLUSTER per image, POSE / POSE2 with every
    correspondence in its own image, FILTER / FILTER2 keyed by (coord2D, image) -- against the oracle's frame."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_scene
    subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_test"])
    db = synth.make_db(6, 1500, seed=8)
    cams = [synth.camera_pose(0.0), synth.camera_pose(-0.12, (0.10, 0.0, 0.0))]
    fr = synth.make_frame_images(db, cams, n_vis=2, seed=5, q_per_image=900, pts_per_obj=130)
    junk = (1, [1.0, 1.0, 0.0, 0.0], CAM0)
    images = [(0, fr.Ks[0], fr.cams[0]), junk, junk, (0, fr.Ks[1], fr.cams[1])]
    scene = str(tmp_path / "scene2.bin")
    dump_scene.dump_images(scene, db, fr, images, np.where(fr.image == 0, 0, 3))
    out = subprocess.check_output([os.path.join(HOST, "moped_hip_test"), "--images", scene, "2"], text=True)
    objs, counts = [], None
    for line in out.splitlines():
        w = line.split()
        if w[0] == "OBJ":
            objs.append((int(w[1].replace("model", "")), np.array([float(x) for x in w[5:9] + w[2:5]], np.float32), float(w[9])))
        if w[0] == "MATCHES":
            counts = (int(w[1]), int(w[3]))
    dbn, qn = orclib.normalize(db.desc), orclib.normalize(fr.desc)
    idx, d1, d2 = orclib.match_2nn(dbn, qn)
    om, op, osc, oc = orclib.frame_rest_images(fr.uv, fr.image, idx, d1, d2, db.model_of, db.xyz, db.n_models, fr.Ks,
                                               fr.cams, seed=2)
    assert counts == (int(oc[0]), int(oc[1]))
    assert sorted(m for m, _, _ in objs) == sorted(om.tolist()) == sorted(fr.visible.tolist())
    for m, pose, score in objs:
        j = list(om).index(m)
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
        rows = rows[db.model_of[fr.src_point[rows]] == m]
        xyz, uv, img = db.xyz[fr.src_point[rows]], fr.uv[rows], fr.image[rows]
        e_g = np.sqrt(((orclib.project_images(pose, xyz, img, fr.Ks, fr.cams) - uv) ** 2).sum(1)).mean()
        e_o = np.sqrt(((orclib.project_images(op[j], xyz, img, fr.Ks, fr.cams) - uv) ** 2).sum(1)).mean()
        assert e_g <= e_o + 1.0 and e_g < 1.0
        assert score > 0 and abs(score - osc[j]) <= 0.05 * osc[j]      # scored over both cameras' matches


@pytest.mark.gpu
def test_depthfill_plugin_through_pipeline(tmp_path):
    """moped3d's DEPTHFILL slot: DEPTH_FILL_EXACT_HIP(8, false) as config.hpp:39 wires the CPU step, on a frame that
    lists its camera image and its depth map; the depth map is filled in place and the "<name>.distance" map appended,
    both bit-identical to the oracle's restatement of DEPTH_FILL_EXACT_CPU."""
    subprocess.check_call(["make", "-s", "-C", HOST, "moped3d_depthfill_test"])
    rng = np.random.default_rng(4)
    h, w = 480, 640
    Kd = np.array([525.0, 525.0, 319.5, 239.5], np.float32)
    z = rng.uniform(0.6, 3.0, size=(h, w)).astype(np.float32)
    for _ in range(20):
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(8, 60)
        yy, xx = np.ogrid[:h, :w]
        z[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = -1.0
    d = np.zeros((h, w, 4), np.float32)
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    d[..., 2] = z
    d[..., 0] = (u - Kd[2]) / Kd[0] * z
    d[..., 1] = (v - Kd[3]) / Kd[1] * z
    d[..., 3] = np.sqrt((d[..., :3] ** 2).sum(-1))
    src, dst = str(tmp_path / "depth_in.bin"), str(tmp_path / "depth_out.bin")
    with open(src, "wb") as f:
        f.write(np.array([w, h], np.int32).tobytes() + Kd.tobytes() + d.tobytes())
    for scale, bilinear in ((8, 0), (16, 1)):
        subprocess.check_call([os.path.join(HOST, "moped3d_depthfill_test"), src, dst, str(scale), str(bilinear)])
        raw = np.fromfile(dst, np.float32)
        got, got_dist = raw[:h * w * 4].reshape(h, w, 4), raw[h * w * 4:].reshape(h, w)
        want, want_dist, _ = orclib.depth_fill(d, Kd, scale, bool(bilinear))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert np.array_equal(got_dist.view(np.uint32), want_dist.view(np.uint32))


def _harness_objects(args):
    out = subprocess.check_output([os.path.join(HOST, "moped_hip_test")] + args, text=True)
    objs = []
    for line in out.splitlines():
        w = line.split()
        if w[0] == "OBJ":
            objs.append((int(w[1].replace("model", "")), np.array([float(x) for x in w[5:9] + w[2:5]], np.float64), float(w[9])))
    return objs, out


@pytest.mark.gpu
def test_frame_resident_plugin_is_the_device_resident_frame(tmp_path):
    """FRAME_RESIDENT_HIP -- MATCH_SIFT .. FILTER2 as ONE step of the pipeline (mh_frame_run_host) -- gives the objects
    of the device-resident frame with the plugin's seed, bit for bit at the C ABI and to the harness's six printed
    digits through the MopedPipeline; the step-by-step plugins find the same models."""
    from moped_amd import capi
    import torch
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_scene
    subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_test"])
    db = synth.make_db(6, 2000)
    fr = synth.make_frame(db, n_vis=2, seed=5, Q=1500, pts_per_obj=120)
    scene = str(tmp_path / "scene.bin")
    dump_scene.dump(scene, db, fr)
    got, out = _harness_objects(["--resident", scene, "1"])
    steps, _ = _harness_objects([scene, "1"])
    assert "TIME MATCH_SIFT" in out and "TIME POSE" not in out          # one step ran
    seed = 1 * 2654435761 + 0                                            # frameCounter * 2654435761 + _alg, first frame
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    prm = capi.default_frame_params()
    desc = fr.desc.copy()
    objs, counts = c.frame_run_host(desc, fr.uv, [K], [CAM0], prm, seed)
    assert np.array_equal(desc.view(np.uint32), c.normalize(fr.desc).view(np.uint32))    # written back normalised
    dev = torch.device("cuda:0")
    c.reserve(len(fr.uv))
    qd, uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
    c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), len(fr.uv), K, CAM0, prm, seed)
    want, wcounts = c.frame_fetch()
    c.close()
    assert np.array_equal(counts, wcounts) and len(objs) == len(want) == 2
    assert np.array_equal(objs["model"], want["model"])
    assert np.array_equal(objs["pose"].view(np.uint32), want["pose"].view(np.uint32))
    assert np.array_equal(objs["score"].view(np.uint32), want["score"].view(np.uint32))
    assert [m for m, _, _ in got] == objs["model"].tolist()
    for (m, pose, score), o in zip(got, objs):
        assert np.allclose(pose, o["pose"].astype(np.float64), atol=2e-6) and abs(score - float(o["score"])) < 1e-3
    assert sorted(m for m, _, _ in steps) == sorted(objs["model"].tolist())


@pytest.mark.gpu
def test_frame_resident_plugin_two_cameras_among_maps(tmp_path):
    """The same step on a frame whose features lie in two cameras listed among maps (FrameData::images = [camera, depth
    map, distance map, camera]): the planted objects, poses under a pixel over both cameras' matches."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_scene
    subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_test"])
    db = synth.make_db(6, 1500, seed=8)
    cams = [synth.camera_pose(0.0), synth.camera_pose(-0.12, (0.10, 0.0, 0.0))]
    fr = synth.make_frame_images(db, cams, n_vis=2, seed=5, q_per_image=900, pts_per_obj=130)
    junk = (1, [1.0, 1.0, 0.0, 0.0], CAM0)
    images = [(0, fr.Ks[0], fr.cams[0]), junk, junk, (0, fr.Ks[1], fr.cams[1])]
    scene = str(tmp_path / "scene2.bin")
    dump_scene.dump_images(scene, db, fr, images, np.where(fr.image == 0, 0, 3))
    got, _ = _harness_objects(["--resident", "--images", scene, "2"])
    assert sorted(m for m, _, _ in got) == sorted(fr.visible.tolist())
    for m, pose, score in got:
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
        rows = rows[db.model_of[fr.src_point[rows]] == m]
        xyz, uv, img = db.xyz[fr.src_point[rows]], fr.uv[rows], fr.image[rows]
        e = np.sqrt(((orclib.project_images(pose.astype(np.float32), xyz, img, fr.Ks, fr.cams) - uv) ** 2).sum(1)).mean()
        assert e < 1.0 and score > 0


@pytest.mark.gpu
def test_frame_resident_plugin_publishes_every_constant_and_fills_matches_on_request(tmp_path):
    """getConfig carries all 23 constants of the six steps it stands for (+ DescriptorSize, FillMatches) under the
    reference's names (GET_CONFIG lists: MATCH_ANN_CPU.hpp:122-125, CLUSTER_MEAN_SHIFT_CPU.hpp:168-171,
    POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:248-252, FILTER_PROJECTION_CPU.hpp:68-70), and with FillMatches set through
    setConfig a step behind it finds frameData.matches filled like MATCH_ANN_CPU::process leaves it (:165-176) -- the same
    number of matches the six-step pipeline reports, the same objects."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_scene
    subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_test"])
    db = synth.make_db(6, 1500, seed=3)
    fr = synth.make_frame(db, n_vis=2, seed=11, Q=1200, pts_per_obj=120)
    scene = str(tmp_path / "scene.bin")
    dump_scene.dump(scene, db, fr)
    outs = {}
    for name, args in (("steps", [scene, "1"]), ("resident", ["--resident", scene, "1"]),
                       ("filled", ["--resident", "--fill-matches", scene, "1"])):
        out = subprocess.check_output([os.path.join(HOST, "moped_hip_test"), *args], text=True)
        kv = {l.split()[0]: l.split()[1:] for l in out.splitlines() if l.split() and l.split()[0] in ("CONFIG_KEYS", "MATCHES")}
        objs = sorted(l.split()[1] for l in out.splitlines() if l.startswith("OBJ "))
        outs[name] = (int(kv["CONFIG_KEYS"][0]), int(kv["MATCHES"][0]), objs)
    assert outs["resident"][0] == 24 and outs["filled"][0] == 24      # the 23 constructor constants + FillMatches
    assert outs["resident"][1] == 0                                   # sized, left empty (the default)
    assert outs["filled"][1] == outs["steps"][1] > 100                # what MATCH_BRUTE_HIP put there step by step
    assert outs["filled"][2] == outs["resident"][2] and len(outs["filled"][2]) == 2


@pytest.mark.gpu
def test_run_host_in_halves_from_page_locked_memory_equals_the_one_call():
    """mh_host_alloc + mh_frame_run_host_begin / mh_frame_wait_descriptors / mh_frame_fetch (what FRAME_RESIDENT_HIP calls
    since the end of round 4): the descriptors come back normalised as soon as the wait returns, the objects are those
    of mh_frame_run_host -- bit for bit, three frames in a row through the same page-locked block."""
    from moped_amd import capi
    db = synth.make_db(6, 2000)
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    prm = capi.default_frame_params()
    Q = 1500
    block = c.host_alloc(Q * (128 + 2) * 4)
    desc = block[:Q * 128 * 4].view(np.float32).reshape(Q, 128)
    uv = block[Q * 128 * 4:].view(np.float32).reshape(Q, 2)
    for i in range(3):
        fr = synth.make_frame(db, n_vis=2, seed=40 + i, Q=Q, pts_per_obj=120)
        plain = fr.desc.copy()
        want, wcounts = c.frame_run_host(plain, fr.uv, [K], [CAM0], prm, 77 + i)
        desc[:] = fr.desc
        uv[:] = fr.uv
        c.frame_run_host_begin(desc, uv, [K], [CAM0], prm, 77 + i)
        c.frame_wait_descriptors()
        assert np.array_equal(desc.view(np.uint32), plain.view(np.uint32))     # normalised, before the frame is fetched
        objs, counts = c.frame_fetch()
        assert len(want) == 2 and np.array_equal(counts, wcounts)
        assert np.array_equal(objs["model"], want["model"])
        assert np.array_equal(objs["pose"].view(np.uint32), want["pose"].view(np.uint32))
        assert np.array_equal(objs["score"].view(np.uint32), want["score"].view(np.uint32))
    c.host_free(block)
    c.close()


@pytest.mark.gpu
def test_six_hip_steps_hand_the_frame_over_on_the_device(tmp_path):
    """MATCH_BRUTE_HIP -> CLUSTER_MEAN_SHIFT_HIP -> POSE -> FILTER -> POSE2 -> FILTER2 wired as six steps
    (src/config.hpp:83-120): every step finds FrameData as the HIP step before left it and runs on the device-resident
    frame (mh_step_*, HipHandover) -- six hand-overs per frame, and the frame's objects are bit for bit those of the six
    slots as ONE step (FRAME_RESIDENT_HIP -> mh_frame_run_host): same kernels on the same lists with the same random
    streams.  With MH_STEP_HANDOVER=0 every step takes its upload path: same models, poses within the bar."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_scene
    subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_test"])
    db = synth.make_db(10, 3000)
    fr = synth.make_frame(db, n_vis=4, seed=12, Q=3000)
    scene = str(tmp_path / "scene.bin")
    dump_scene.dump(scene, db, fr)
    exe = os.path.join(HOST, "moped_hip_test")

    def run(args, env=None):
        out = subprocess.check_output([exe] + args, text=True, env=dict(os.environ, **(env or {})))
        objs = [l for l in out.splitlines() if l.startswith("OBJ ")]
        head = {l.split()[0]: l.split()[1:] for l in out.splitlines() if not l.startswith(("OBJ", "TIME"))}
        return objs, head
    stepped, h1 = run([scene, "3"])
    resident, _ = run(["--resident", scene, "3"])
    upload, h0 = run([scene, "3"], {"MH_STEP_HANDOVER": "0"})
    assert h1["HANDOVER_STEPS"][0] == "18" and h0["HANDOVER_STEPS"][0] == "0"      # 6 steps x 3 frames / none
    assert len(stepped) >= 4 and stepped == resident                                 # printed to 1e-6: the same objects
    assert h1["MATCHES"] == h0["MATCHES"]                                            # matches, clusters, objects after POSE counts
    assert sorted(l.split()[1] for l in upload) == sorted(l.split()[1] for l in stepped)
    for a in upload:
        b = [l for l in stepped if l.split()[1] == a.split()[1]][0]
        pa, pb = np.array(a.split()[2:9], float), np.array(b.split()[2:9], float)
        assert np.abs(pa[:3] - pb[:3]).max() < 3e-3 and min(np.abs(pa[3:] - pb[3:]).max(), np.abs(pa[3:] + pb[3:]).max()) < 2e-2
