"""Seeded synthetic workloads for the MATCH -> CLUSTER -> POSE path.

Everything is float32/int32 numpy on the host.  Object models are absent from the
reference tree (moped2/download_models.sh fetches them), so model databases are
synthesised: SIFT-like descriptors are derived from the real SIFT descriptors of
the five frames bundled in moped2/test_data/timing.bag (tests/golden/sift_frames.npz,
extracted once with the reference's own libsiftfast -- oracle/make_golden.py).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_FIXTURE = os.path.join(_HERE, "..", "tests", "golden", "sift_frames.npz")

# 640x480, K = (fx, fy, cx, cy), identity camera pose (moped2/moped_test.cpp:187-188)
K_DEFAULT = np.array([800.0, 800.0, 320.0, 240.0], dtype=np.float32)
CAM_IDENTITY = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float32)
IMG_W, IMG_H = 640, 480


def load_sift_fixture():
    """Real SIFT descriptors (x512 quantised to u8) + keypoint xy + frame id."""
    z = np.load(_FIXTURE)
    desc = z["desc_u8"].astype(np.float32) / np.float32(512.0)
    return desc, z["xy"].astype(np.float32), z["frame"].astype(np.int32)


def l2_normalize(d: np.ndarray) -> np.ndarray:
    """Plain numpy normalisation used only to *generate* data (the pipeline
    re-normalises with the reference's arithmetic, MATCH_ANN_CPU.hpp:54-57)."""
    n = np.sqrt((d.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
    n[n == 0] = 1.0
    return (d / n).astype(np.float32)


@dataclass
class ModelDB:
    desc: np.ndarray      # [N,128] float32, near unit norm
    xyz: np.ndarray       # [N,3]  float32 model-frame coordinates (metres)
    model_of: np.ndarray  # [N]    int32
    n_models: int

    @property
    def n(self) -> int:
        return int(self.desc.shape[0])


def make_db(n_models: int, pts_per_model: int = 5000, seed: int = 0xC0FFEE,
            base: np.ndarray | None = None) -> ModelDB:
    """SIFT-like model database: base descriptor from the real fixture +
    N(0, 0.03^2) per dim, clamped at 0, L2-normalised; 3-D points uniform in a
    0.10 x 0.10 x 0.20 m box."""
    if base is None:
        base, _, _ = load_sift_fixture()
    descs, xyzs, owners = [], [], []
    for m in range(n_models):
        rng = np.random.default_rng([seed, m])
        pick = rng.integers(0, base.shape[0], size=pts_per_model)
        d = base[pick] + rng.normal(0.0, 0.03, size=(pts_per_model, 128)).astype(np.float32)
        d = l2_normalize(np.maximum(d, 0.0).astype(np.float32))
        x = (rng.random((pts_per_model, 3)) - 0.5) * np.array([0.10, 0.10, 0.20])
        descs.append(d)
        xyzs.append(x.astype(np.float32))
        owners.append(np.full(pts_per_model, m, dtype=np.int32))
    return ModelDB(np.ascontiguousarray(np.concatenate(descs)),
                   np.ascontiguousarray(np.concatenate(xyzs)),
                   np.concatenate(owners), n_models)


def random_quat(rng) -> np.ndarray:
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return q.astype(np.float32)  # (x, y, z, w)


def quat_to_R(q) -> np.ndarray:
    x, y, z, w = [float(v) for v in q]
    return np.array([
        [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y],
        [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
        [2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def project_np(pose7, xyz, K=K_DEFAULT):
    """float64 pinhole projection with an identity camera (generation only)."""
    R = quat_to_R(pose7[:4])
    p = xyz.astype(np.float64) @ R.T + np.asarray(pose7[4:7], dtype=np.float64)
    u = p[:, 0] / p[:, 2] * float(K[0]) + float(K[2])
    v = p[:, 1] / p[:, 2] * float(K[1]) + float(K[3])
    return np.stack([u, v], axis=1), p[:, 2]


@dataclass
class Frame:
    desc: np.ndarray       # [Q,128] float32 query descriptors (un-normalised is fine)
    uv: np.ndarray         # [Q,2]   float32 pixel coordinates
    visible: np.ndarray    # [n_vis] int32 model ids planted in this frame
    poses: np.ndarray      # [n_vis,7] float32 planted poses (qx,qy,qz,qw,tx,ty,tz)
    src_point: np.ndarray  # [Q] int32 DB row a planted feature came from, -1 = clutter
    is_outlier: np.ndarray # [Q] bool planted feature with a wrong 2-D location


def make_frame(db: ModelDB, n_vis: int = 2, seed: int = 0, Q: int = 3000,
               pts_per_obj: int = 150, outlier_frac: float = 0.2,
               pix_noise: float = 0.5, K=K_DEFAULT,
               base: np.ndarray | None = None) -> Frame:
    """One 640x480 frame: n_vis planted objects x pts_per_obj features (+/-0.5 px
    uniform noise, descriptor = model descriptor + N(0, 0.01^2) renormalised,
    outlier_frac of them at a wrong pixel), the rest clutter (real fixture
    descriptors at uniform pixel positions), shuffled."""
    if base is None:
        base, _, _ = load_sift_fixture()
    rng = np.random.default_rng([0xF4A3E, seed])
    n_vis = min(n_vis, db.n_models)
    visible = rng.choice(db.n_models, size=n_vis, replace=False).astype(np.int32)
    descs, uvs, srcs, outl, poses = [], [], [], [], []
    for m in visible:
        rows = np.nonzero(db.model_of == m)[0]
        for _ in range(100):
            q = random_quat(rng)
            z = rng.uniform(0.5, 1.0)
            u0 = rng.uniform(120, IMG_W - 120)
            v0 = rng.uniform(100, IMG_H - 100)
            t = np.array([(u0 - K[2]) / K[0] * z, (v0 - K[3]) / K[1] * z, z])
            pose = np.concatenate([q, t]).astype(np.float32)
            uv_all, zc = project_np(pose, db.xyz[rows], K)
            ok = (zc > 0.05) & (uv_all[:, 0] >= 0) & (uv_all[:, 0] < IMG_W) & \
                 (uv_all[:, 1] >= 0) & (uv_all[:, 1] < IMG_H)
            if ok.sum() >= pts_per_obj:
                break
        sel = rng.choice(np.nonzero(ok)[0], size=pts_per_obj, replace=False)
        uv = uv_all[sel] + rng.uniform(-pix_noise, pix_noise, size=(pts_per_obj, 2))
        bad = rng.random(pts_per_obj) < outlier_frac
        uv[bad] = rng.uniform([0, 0], [IMG_W, IMG_H], size=(int(bad.sum()), 2))
        d = db.desc[rows[sel]] + rng.normal(0, 0.01, size=(pts_per_obj, 128)).astype(np.float32)
        descs.append(l2_normalize(np.maximum(d, 0).astype(np.float32)))
        uvs.append(uv.astype(np.float32))
        srcs.append(rows[sel].astype(np.int32))
        outl.append(bad)
        poses.append(pose)
    n_planted = n_vis * pts_per_obj
    n_clutter = max(Q - n_planted, 0)
    pick = rng.integers(0, base.shape[0], size=n_clutter)
    descs.append(base[pick])
    uvs.append(rng.uniform([0, 0], [IMG_W, IMG_H], size=(n_clutter, 2)).astype(np.float32))
    srcs.append(np.full(n_clutter, -1, dtype=np.int32))
    outl.append(np.zeros(n_clutter, dtype=bool))
    desc = np.concatenate(descs)
    uv = np.concatenate(uvs)
    src = np.concatenate(srcs)
    bad = np.concatenate(outl)
    perm = rng.permutation(desc.shape[0])
    return Frame(np.ascontiguousarray(desc[perm], dtype=np.float32),
                 np.ascontiguousarray(uv[perm], dtype=np.float32),
                 visible, np.asarray(poses, dtype=np.float32).reshape(-1, 7),
                 src[perm], bad[perm])


def camera_pose(rot_y: float = 0.0, t=(0.0, 0.0, 0.0)) -> np.ndarray:
    """cameraPose (qx,qy,qz,qw,tx,ty,tz) of a camera rotated by rot_y about the world's y axis at position t
    (Image::cameraPose maps camera coordinates to world coordinates, include/moped.hpp:226-241)."""
    return np.array([0.0, np.sin(rot_y / 2), 0.0, np.cos(rot_y / 2), t[0], t[1], t[2]], np.float32)


def project_cam_np(pose7, xyz, K, cam):
    """float64 projection of model points at world pose `pose7` through the camera with cameraPose `cam`."""
    R = quat_to_R(pose7[:4])
    pw = xyz.astype(np.float64) @ R.T + np.asarray(pose7[4:7], np.float64)
    Rc = quat_to_R(cam[:4])
    pc = (pw - np.asarray(cam[4:7], np.float64)) @ Rc          # TM^-1: Rc^T (p - tc)
    u = pc[:, 0] / pc[:, 2] * float(K[0]) + float(K[2])
    v = pc[:, 1] / pc[:, 2] * float(K[1]) + float(K[3])
    return np.stack([u, v], axis=1), pc[:, 2]


@dataclass
class FrameImages:
    desc: np.ndarray        # [Q,128]
    uv: np.ndarray          # [Q,2]
    image: np.ndarray       # [Q] int32 image (camera) of every feature
    visible: np.ndarray
    poses: np.ndarray       # [n_vis,7] planted WORLD poses
    src_point: np.ndarray   # [Q] DB row of a planted feature, -1 = clutter
    is_outlier: np.ndarray
    Ks: np.ndarray          # [n_images,4]
    cams: np.ndarray        # [n_images,7]


def make_frame_images(db: ModelDB, cams, n_vis: int = 2, seed: int = 0, q_per_image: int = 1200,
                      pts_per_obj: int = 120, outlier_frac: float = 0.2, pix_noise: float = 0.5, K=K_DEFAULT,
                      base: np.ndarray | None = None) -> FrameImages:
    """One frame seen by several cameras (FrameData::images): every planted object is rendered into every image
    (its own random subset of the model's points), clutter per image; the features of image 0 come first, then
    image 1, ..., each image's features shuffled among themselves (as FEAT extracts image by image)."""
    if base is None:
        base, _, _ = load_sift_fixture()
    cams = np.asarray(cams, np.float32).reshape(-1, 7)
    rng = np.random.default_rng([0x1A6E5, seed])
    visible = rng.choice(db.n_models, size=min(n_vis, db.n_models), replace=False).astype(np.int32)
    poses = []
    per_image = [([], [], [], []) for _ in cams]     # desc, uv, src, outlier
    for m in visible:
        rows = np.nonzero(db.model_of == m)[0]
        for _ in range(200):
            q = random_quat(rng)
            z = rng.uniform(0.6, 0.9)
            u0, v0 = rng.uniform(200, IMG_W - 200), rng.uniform(160, IMG_H - 160)
            pose = np.concatenate([q, [(u0 - K[2]) / K[0] * z, (v0 - K[3]) / K[1] * z, z]]).astype(np.float32)
            oks = []
            for c in cams:
                uv_all, zc = project_cam_np(pose, db.xyz[rows], K, c)
                oks.append((uv_all, (zc > 0.05) & (uv_all[:, 0] >= 0) & (uv_all[:, 0] < IMG_W) &
                            (uv_all[:, 1] >= 0) & (uv_all[:, 1] < IMG_H)))
            if all(ok.sum() >= pts_per_obj for _, ok in oks):
                break
        poses.append(pose)
        for ci, (uv_all, ok) in enumerate(oks):
            sel = rng.choice(np.nonzero(ok)[0], size=pts_per_obj, replace=False)
            uv = uv_all[sel] + rng.uniform(-pix_noise, pix_noise, size=(pts_per_obj, 2))
            bad = rng.random(pts_per_obj) < outlier_frac
            uv[bad] = rng.uniform([0, 0], [IMG_W, IMG_H], size=(int(bad.sum()), 2))
            d = db.desc[rows[sel]] + rng.normal(0, 0.01, size=(pts_per_obj, 128)).astype(np.float32)
            per_image[ci][0].append(l2_normalize(np.maximum(d, 0).astype(np.float32)))
            per_image[ci][1].append(uv.astype(np.float32))
            per_image[ci][2].append(rows[sel].astype(np.int32))
            per_image[ci][3].append(bad)
    descs, uvs, imgs, srcs, outl = [], [], [], [], []
    for ci in range(len(cams)):
        n_planted = sum(len(x) for x in per_image[ci][2])
        n_clutter = max(q_per_image - n_planted, 0)
        pick = rng.integers(0, base.shape[0], size=n_clutter)
        d = np.concatenate(per_image[ci][0] + [base[pick]])
        uv = np.concatenate(per_image[ci][1] + [rng.uniform([0, 0], [IMG_W, IMG_H], size=(n_clutter, 2)).astype(np.float32)])
        src = np.concatenate(per_image[ci][2] + [np.full(n_clutter, -1, np.int32)])
        bad = np.concatenate(per_image[ci][3] + [np.zeros(n_clutter, bool)])
        perm = rng.permutation(len(d))
        descs.append(d[perm]); uvs.append(uv[perm]); srcs.append(src[perm]); outl.append(bad[perm])
        imgs.append(np.full(len(d), ci, np.int32))
    return FrameImages(np.ascontiguousarray(np.concatenate(descs), np.float32),
                       np.ascontiguousarray(np.concatenate(uvs), np.float32), np.concatenate(imgs), visible,
                       np.asarray(poses, np.float32).reshape(-1, 7), np.concatenate(srcs), np.concatenate(outl),
                       np.tile(np.asarray(K, np.float32), (len(cams), 1)), cams)


def frame_depth(db: ModelDB, frame: Frame, seed: int = 0, K=K_DEFAULT, fill_max: float = 0.02):
    """Per-query depth attributes for the moped3d (Kinect) configuration: the camera-frame
    point the depth map holds at each keypoint (planted inliers: the true point with
    sigma = 0.0035 z^2 depth noise along the ray; everything else: a point on the pixel's
    ray at a random depth) and a fill distance in [0, fill_max] m (0 = measured pixel).
    Returns (world [Q,3] float32, fill [Q] float32)."""
    rng = np.random.default_rng([0xD3B7, seed])
    Q = frame.uv.shape[0]
    z = rng.uniform(0.5, 1.5, Q)
    world = np.stack([(frame.uv[:, 0] - K[2]) / K[0] * z, (frame.uv[:, 1] - K[3]) / K[1] * z, z], 1)
    for j, m in enumerate(frame.visible):
        rows = np.nonzero((frame.src_point >= 0) & ~frame.is_outlier)[0]
        rows = rows[db.model_of[frame.src_point[rows]] == m]
        R = quat_to_R(frame.poses[j][:4])
        p = db.xyz[frame.src_point[rows]].astype(np.float64) @ R.T + frame.poses[j][4:].astype(np.float64)
        p *= 1 + rng.normal(0, 0.0035, (len(rows), 1)) * p[:, 2:3]
        world[rows] = p
    fill = rng.uniform(0, fill_max, Q) * (rng.random(Q) < 0.3)
    return world.astype(np.float32), fill.astype(np.float32)


def write_model_xml(path, name: str, xyz: np.ndarray, desc: np.ndarray, full_export: bool = False,
                    desc_type: str = "SIFT", seed: int = 0):
    """A `.moped.xml` model file in the layout moped2/modeling/sfm_export_xml.m writes
    (:62-131): `%.6f` decimals separated by blanks, one <Point> per model point, with
    <Observation> children when full_export.  Returns the float32 arrays a loader must
    produce (the decimals rounded to float32)."""
    rng = np.random.default_rng([0xA11CE, seed])
    n = xyz.shape[0]
    out = [f'<Model name="{name}" version="Bundler v0.3">\n', "  <Openrave>\n", f"    <name>{name}</name>\n",
           f"    <xml>{name}.kinbody.xml</xml>\n",
           "    <transf>" + "".join("%.6f " % v for v in np.eye(3).reshape(-1).tolist() + [0, 0, 0]) + "</transf>\n",
           "  </Openrave>\n", "  <Points>\n"]
    for i in range(n):
        out.append('    <Point p3d="%.6f %.6f %.6f" nviews="%d" avg_err="%.6f" color="%d %d %d" desc_type="%s" desc="'
                   % (xyz[i, 0], xyz[i, 1], xyz[i, 2], 3 + i % 5, 0.25 + 0.01 * (i % 7), i % 256, (3 * i) % 256,
                      (7 * i) % 256, desc_type))
        out.append("".join("%.6f " % v for v in desc[i]))
        out.append('">\n')
        if full_export:
            for j in range(2):
                out.append('      <Observation camera_id="%d" desc_type="SIFT" loc="' % (j + 1))
                out.append("".join("%.6f " % v for v in rng.uniform(0, 640, 4)))
                out.append('" desc="')
                out.append("".join("%.6f " % v for v in np.abs(desc[i] + rng.normal(0, 0.01, desc.shape[1]))))
                out.append('"/>\n')
        out.append("</Point>\n")
    out.append("  </Points>\n")
    out.append("</Model>\n")
    text = "".join(out)
    with open(path, "w") as f:
        f.write(text)
    r = lambda a: np.array([[np.float32("%.6f" % v) for v in row] for row in a], np.float32).reshape(a.shape)
    return r(xyz), r(desc)


def depth_image(db: ModelDB, frame: Frame, seed: int = 0, K=K_DEFAULT, fill_max: float = 0.02):
    """The moped3d depth map of a frame (moped3d/moped3d.cpp:279-333): [480,640,4] float32 =
    camera-frame (x, y, z, norm) per pixel -- a background surface at 1.5 m, the planted
    objects' points at the pixels their keypoints truncate to (sigma = 0.0035 z^2 depth noise
    along the ray), 5 % invalid pixels (norm = -1) -- and the per-pixel fill distance
    [480,640] (0 on measured pixels, up to fill_max on 30 % "filled" ones)."""
    rng = np.random.default_rng([0xD1A6, seed])
    v, u = np.mgrid[0:IMG_H, 0:IMG_W].astype(np.float64)
    z = 1.5 + 0.05 * np.sin(u / 40.0) * np.cos(v / 55.0)
    img = np.stack([(u + 0.5 - K[2]) / K[0] * z, (v + 0.5 - K[3]) / K[1] * z, z, np.zeros_like(z)], -1)
    world, _ = frame_depth(db, frame, seed=seed, K=K, fill_max=fill_max)
    rows = np.nonzero((frame.src_point >= 0) & ~frame.is_outlier)[0]
    ix = np.clip(frame.uv[rows, 0].astype(np.int32), 0, IMG_W - 1)
    iy = np.clip(frame.uv[rows, 1].astype(np.int32), 0, IMG_H - 1)
    img[iy, ix, :3] = world[rows]
    img[..., 3] = np.sqrt((img[..., :3] ** 2).sum(-1))
    invalid = rng.random((IMG_H, IMG_W)) < 0.05
    invalid[iy, ix] = False
    img[invalid, 3] = -1.0
    fill = rng.uniform(0, fill_max, (IMG_H, IMG_W)) * (rng.random((IMG_H, IMG_W)) < 0.3)
    return np.ascontiguousarray(img, np.float32), np.ascontiguousarray(fill, np.float32)


def textured_image(seed: int = 0, height: int = 480, width: int = 640, fine: float = 0.42):
    """A 640x480 gray image that yields about 3 000 SIFT keypoints (with the first octave doubled, as FEAT_SIFT_CPU
    extracts: moped2/libmoped/src/feat/FEAT_SIFT_CPU.hpp:78-112) -- the keypoint count BASELINE.json's metric names;
    the reference's bundled frames give ~590.  Band-limited noise over four octaves, the finest band at weight `fine`
    (1.0: ~8 500 keypoints, 0.25: ~2 000).  uint8 [height, width]."""
    from scipy import ndimage
    rng = np.random.default_rng([0x5157, seed])
    img = np.zeros((height, width))
    for k, a in enumerate((fine, 1.0, 1.0, 1.0)):
        s = 1.2 * 2 ** k
        img += a * s * ndimage.gaussian_filter(rng.normal(size=(height, width)), s)
    img = (img - img.min()) / (img.max() - img.min())
    return np.round(255 * img).astype(np.uint8)
