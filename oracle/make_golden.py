#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own libraries (oracle/_ref).

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference for
the bag file and a built oracle/_ref); the .npz outputs are data -- inputs and the
reference's outputs -- and are committed so the GPU box can check against them
without the reference.

  sift_frames.npz   real SIFT descriptors of the 5 frames in
                    moped2/test_data/timing.bag, extracted with the reference's
                    libsiftfast the way FEAT_SIFT_CPU.hpp:78-112 calls it;
                    descriptors quantised x512 to uint8 (values are defined as u8/512)
  match_ann_*.npz   ANN 1.1.1 kd-tree 2-NN at eps=0 (exact) and eps=5 (shipped
                    default, config.hpp:83) for fixture queries vs seeded DBs
  pose_depth_ref.npz  the two moped3d depth residual models (A14): residual tables and
                    slevmar_dif end states
  sift_ref_frames.npz   two bundled frames (gray) + the reference libsiftfast's keypoints for
                    them, full precision (N2; `OMP_NUM_THREADS=1 make_golden.py siftref`)
  models/*.moped.xml + model_xml_ref.npz   small model files and what the reference's
                    sXML.hpp + stream operators read out of them (N3)
  pose_ref.npz      project() / lmFuncQuat residual tables and slevmar_dif end
                    states for seeded 5/6-point and inlier-set problems
"""
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import orclib  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
BAG = "/root/reference/moped2/test_data/timing.bag"


def make_sift_fixture():
    from PIL import Image
    bag = open(BAG, "rb").read()
    R = orclib.ref()
    descs, xys, frames = [], [], []
    pos, f = 0, 0
    while True:
        pos = bag.find(b"\xff\xd8\xff", pos)
        if pos < 0:
            break
        img = Image.open(io.BytesIO(bag[pos:]))
        img.load()
        g = np.ascontiguousarray(np.array(img.convert("L")), dtype=np.uint8)
        h, w = g.shape
        xy = np.zeros((8192, 2), np.float32)
        d = np.zeros((8192, 128), np.float32)
        n = R.ref_sift(g.reshape(-1), w, h, xy.reshape(-1), d.reshape(-1), 8192)
        descs.append(d[:n])
        xys.append(xy[:n])
        frames.append(np.full(n, f, np.int32))
        print(f"frame {f}: {w}x{h} -> {n} keypoints")
        pos += 3
        f += 1
    d = np.concatenate(descs)
    u8 = np.clip(np.rint(d * 512.0), 0, 255).astype(np.uint8)
    np.savez_compressed(os.path.join(GOLD, "sift_frames.npz"), desc_u8=u8,
                        xy=np.concatenate(xys), frame=np.concatenate(frames))
    print("sift_frames.npz:", u8.shape)


def make_match_golden():
    from moped_amd import synth
    base, _, _ = synth.load_sift_fixture()
    q = orclib.normalize(base)  # MATCH_ANN_CPU.hpp:157
    for tag, n_models, ppm in (("1k", 1, 1000), ("10k", 2, 5000), ("100k", 20, 5000)):
        db = synth.make_db(n_models, ppm)
        dbn = orclib.normalize(db.desc)  # MATCH_ANN_CPU.hpp:94
        ann = orclib.RefAnn(dbn)
        idx0, dist0 = ann.search2(q, 0.0)
        idx5, dist5 = ann.search2(q, 5.0)
        ann.close()
        np.savez_compressed(os.path.join(GOLD, f"match_ann_{tag}.npz"),
                            n_models=n_models, pts_per_model=ppm,
                            idx_eps0=idx0, dist_eps0=dist0, idx_eps5=idx5, dist_eps5=dist5)
        acc0 = (dist0[:, 0] / dist0[:, 1] < 0.8).sum()
        acc5 = (dist5[:, 0] / dist5[:, 1] < 0.8).sum()
        print(f"match_ann_{tag}: N={db.n} accepted eps0={acc0} eps5={acc5}")


def make_pose_golden():
    from moped_amd import synth
    rng = np.random.default_rng(20240607)
    K = synth.K_DEFAULT
    cams = [synth.CAM_IDENTITY,
            np.concatenate([synth.random_quat(rng), [0.05, -0.02, 0.1]]).astype(np.float32)]
    out = {}
    cases = []
    for ci in range(24):
        cam = cams[ci % 2]
        n = [5, 6, 12, 40][ci % 4]
        xyz = ((rng.random((n, 3)) - 0.5) * [0.1, 0.1, 0.2]).astype(np.float32)
        q = synth.random_quat(rng)
        t = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(0.5, 1.0)])
        pose = np.concatenate([q, t]).astype(np.float32)
        uv = orclib.ref_project(pose, xyz, K, cam)
        uv = (uv + rng.uniform(-0.5, 0.5, uv.shape)).astype(np.float32)
        # start: perturbed truth (refine-like) for even cases, the reference's
        # initPose-style random quaternion + (0,0,0.5) for odd ones
        if ci % 2 == 0:
            start = pose + rng.normal(0, 0.02, 7).astype(np.float32)
        else:
            start = np.array([rng.integers(0, 256) / 256., rng.integers(0, 256) / 256.,
                              rng.integers(0, 256) / 256., rng.integers(0, 256) / 256.,
                              0, 0, 0.5], np.float32)
        itmax = 200 if ci % 4 < 2 else 500
        hx = orclib.ref_residuals(start, uv, xyz, K, cam)
        ret, p_end, info = orclib.ref_optimize_camera(start, uv, xyz, K, cam, itmax)
        uv_end = orclib.ref_project(p_end, xyz, K, cam)
        cases.append(dict(cam=cam, xyz=xyz, uv=uv, pose_true=pose, start=start, itmax=itmax,
                          hx_start=hx, ret=ret, pose_end=p_end, info=info, uv_end=uv_end))
    for i, c in enumerate(cases):
        for k, v in c.items():
            out[f"c{i}_{k}"] = np.asarray(v)
    out["n_cases"] = len(cases)
    out["K"] = K
    # behind-the-camera / near-plane table for project() and the residual branch
    pose = np.array([0, 0, 0, 1, 0, 0, 0.0], np.float32)
    xyz = np.array([[0, 0, 1], [0.1, -0.1, 0.5], [0, 0, 0.0005], [0, 0, -0.3], [0.2, 0.1, 0.001],
                    [0.05, 0.05, 0.00099]], np.float32)
    out["edge_xyz"] = xyz
    out["edge_uv"] = orclib.ref_project(pose, xyz, K, synth.CAM_IDENTITY)
    out["edge_hx"] = orclib.ref_residuals(pose, np.zeros((6, 2), np.float32), xyz, K, synth.CAM_IDENTITY)
    np.savez_compressed(os.path.join(GOLD, "pose_ref.npz"), **out)
    print("pose_ref.npz:", len(cases), "cases")


def make_pose_depth_golden():
    """moped3d residual tables and slevmar_dif end states for both depth classes."""
    from moped_amd import synth
    rng = np.random.default_rng(77)
    K = synth.K_DEFAULT
    cam = synth.CAM_IDENTITY
    out = {"K": K}
    i = 0
    for mode in (1, 2):
        for n in (5, 6, 20, 60):
            xyz = ((rng.random((n, 3)) - 0.5) * [0.1, 0.1, 0.2]).astype(np.float32)
            pose = np.concatenate([synth.random_quat(rng), [rng.uniform(-.1, .1), rng.uniform(-.1, .1),
                                                            rng.uniform(0.5, 1.0)]]).astype(np.float32)
            uv = (orclib.ref_project(pose, xyz, K, cam) + rng.uniform(-0.5, 0.5, (n, 2))).astype(np.float32)
            R = synth.quat_to_R(pose[:4])
            world = xyz.astype(np.float64) @ R.T + pose[4:].astype(np.float64)
            world *= 1 + rng.normal(0, 0.0035, (n, 1)) * world[:, 2:3]      # Kinect-like depth noise ~ z^2
            world = world.astype(np.float32)
            fill = rng.uniform(0, 0.08 if mode == 1 else 20.0, n)
            wgt = orclib.cauchy_weight(fill, 0.1 if mode == 1 else 25.0)
            start = (pose + rng.normal(0, 0.02, 7)).astype(np.float32)
            err = orclib.ref_residuals_depth(mode, start, uv, xyz, world, wgt, K, cam, 0.5)
            ret, p_end, info = orclib.ref_optimize_camera_depth(mode, start, uv, xyz, world, wgt, K, cam, 0.5, 100)
            for k, v in dict(mode=mode, xyz=xyz, uv=uv, world=world, wgt=wgt, pose_true=pose, start=start,
                             err_start=err, ret=ret, pose_end=p_end, info=info).items():
                out[f"c{i}_{k}"] = np.asarray(v)
            i += 1
    # a point behind the camera exercises the penalty branch
    out["n_cases"] = i
    np.savez_compressed(os.path.join(GOLD, "pose_depth_ref.npz"), **out)
    print("pose_depth_ref.npz:", i, "cases")


def make_sift_ref_golden():
    """Two of the bundled frames (gray, uint8) and the reference's own libsiftfast output for them
    (oracle/_ref ref_sift2, ONE OpenMP thread: with more the list order depends on timing)."""
    assert os.environ.get("OMP_NUM_THREADS") == "1", "run with OMP_NUM_THREADS=1"
    from PIL import Image
    bag = open(BAG, "rb").read()
    out = {}
    pos, f = 0, 0
    while True:
        pos = bag.find(b"\xff\xd8\xff", pos)
        if pos < 0:
            break
        if f in (0, 3):
            img = Image.open(io.BytesIO(bag[pos:]))
            img.load()
            g = np.ascontiguousarray(np.array(img.convert("L")), dtype=np.uint8)
            xy, so, d = orclib.ref_sift(g)
            out[f"gray{f}"], out[f"xy{f}"], out[f"scale_ori{f}"], out[f"desc{f}"] = g, xy, so, d
            print(f"frame {f}: {len(xy)} keypoints")
        pos += 3
        f += 1
    out["frames"] = np.array([0, 3])
    np.savez_compressed(os.path.join(GOLD, "sift_ref_frames.npz"), **out)


def make_model_golden():
    """Small `.moped.xml` fixtures (layout of moped2/modeling/sfm_export_xml.m) and what the
    reference's own sXML.hpp + stream operators read out of them (oracle/_ref ref_model_xml)."""
    from moped_amd import synth
    d = os.path.join(GOLD, "models")
    os.makedirs(d, exist_ok=True)
    base, _, _ = synth.load_sift_fixture()
    rng = np.random.default_rng(4242)
    out = {}
    names = []
    for i, (nm, n, full) in enumerate((("tazo_box", 40, False), ("rice tuscan", 25, True), ("odwalla", 30, False))):
        xyz = ((rng.random((n, 3)) - 0.5) * [0.1, 0.1, 0.2]).astype(np.float32)
        desc = base[rng.integers(0, base.shape[0], n)]
        path = os.path.join(d, f"{nm.replace(' ', '_')}.moped.xml")
        synth.write_model_xml(path, nm, xyz, desc, full_export=full, seed=i)
        names.append(os.path.basename(path))
    # hand-made corner cases: comment before an element, escapes, attribute order, a point of
    # another descriptor type, a short descriptor, '+' signs / exponents, a self-closed point,
    # a second <Points> that replaces the first
    quirks = os.path.join(d, "quirks.moped.xml")
    d128 = " ".join("%.6f" % (0.001 * k) for k in range(128))
    with open(quirks, "w") as f:
        f.write('<Model version="x" name="quirk \\"quoted\\" model">\n'
                '  <!-- a comment -->\n  <Openrave><name>q</name></Openrave>\n'
                '  <Points>\n    <Point p3d="9 9 9" desc_type="SIFT" desc="' + d128 + '"/>\n  </Points>\n'
                '  <Points>\n'
                '    <Point desc="' + d128 + '" desc_type="SIFT" p3d="+1.5e-2 -2.5E-1 .125"/>\n'
                '    <Point p3d="0.1 0.2 0.3" desc_type="SURF" desc="1 2 3"></Point>\n'
                '    <!-- between points -->\n'
                '    <Point p3d="-0.4 0.5" desc_type="SIFT" desc="' + d128 + '">\n'
                '      <Observation camera_id="1" desc_type="SIFT" loc="1 2 3 4" desc="' + d128 + '"/>\n'
                '    </Point>\n'
                '    <Point p3d="0.7 0.8 0.9 1.0" desc_type="SIFT" desc="0.5 0.25 oops 0.125"/>\n'
                '  </Points>\n</Model>\n')
    names.append("quirks.moped.xml")
    for nm in names:
        r = orclib.ref_model_xml(os.path.join(d, nm))
        assert r is not None, nm
        key = nm.split(".")[0]
        for k, v in r.items():
            out[f"{key}_{k}"] = np.asarray(v)
        print(f"{nm}: name={r['name']!r} points={len(r['xyz'])} bad_len={r['n_bad_len']}")
    out["files"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "model_xml_ref.npz"), **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1:] or ["sift", "match", "pose", "depth", "models"]
    if "sift" in what:
        make_sift_fixture()
    if "match" in what:
        make_match_golden()
    if "pose" in what:
        make_pose_golden()
    if "depth" in what:
        make_pose_depth_golden()
    if "models" in what:
        make_model_golden()
    if "siftref" in what:
        make_sift_ref_golden()
