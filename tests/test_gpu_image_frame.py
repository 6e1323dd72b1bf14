"""Image in, objects out (mh_frame_enqueue_image): FEAT on the device feeds MATCH..FILTER2
without a host round trip, the keypoint count stays on the device.  Must give exactly what
extracting the features first (mh_sift_extract) and enqueueing them with their host-known
count (mh_frame_enqueue) gives -- same kernels, same arithmetic, only the launch sizing differs.

Scene: the "planar model from frame 0" of SURVEY 8(c)(viii): the keypoints of the reference's
first bundled frame, back-projected onto a plane 0.8 m in front of the camera, are the model;
every bundled frame (the camera barely moves in timing.bag) must then show it."""
import os

import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sift_ref_frames.npz"))
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY
CAP = 2048


@pytest.fixture(scope="module")
def scene():
    import torch
    c = capi.Context(0)
    xy, _, desc = c.sift(GOLD["gray0"])
    z = np.float32(0.8)
    xyz = np.stack([(xy[:, 0] - K[2]) / K[0] * z, (xy[:, 1] - K[3]) / K[1] * z, np.full(len(xy), z)], 1).astype(np.float32)
    rng = np.random.default_rng(7)
    clutter = np.abs(rng.normal(size=(500, 128))).astype(np.float32)       # a second model nobody sees
    db_desc = orclib.normalize(np.concatenate([desc, clutter]))
    db_xyz = np.concatenate([xyz, rng.uniform(-0.1, 0.1, (500, 3)).astype(np.float32)])
    model_of = np.concatenate([np.zeros(len(xy), np.int32), np.ones(500, np.int32)])
    c.db_upload(db_desc, model_of, db_xyz, 2)
    c.reserve(CAP)
    yield c, torch, db_desc, db_xyz, model_of
    c.close()


@pytest.mark.parametrize("f", [int(x) for x in GOLD["frames"]])
def test_image_frame_equals_features_then_frame(scene, f):
    c, torch, db_desc, db_xyz, model_of = scene
    dev = torch.device("cuda:0")
    gray = GOLD[f"gray{f}"]
    h, w = gray.shape
    prm = capi.default_frame_params()

    # two steps through the host: FEAT, then the frame with Q known
    xy, _, desc = c.sift(gray)
    n = len(xy)
    q_desc, q_uv = torch.from_numpy(desc).to(dev), torch.from_numpy(xy).to(dev)
    torch.cuda.synchronize()
    c.frame_enqueue(q_desc.data_ptr(), q_uv.data_ptr(), n, K, CAM0, prm, seed=5 + f)
    want, want_counts = c.frame_fetch()

    # one call, image resident on the device (twice: the second launch is sized by the first's count)
    g = torch.from_numpy(gray).to(dev)
    torch.cuda.synchronize()
    for _ in range(2):
        c.frame_enqueue_image(g.data_ptr(), w, h, True, CAP, K, CAM0, prm, seed=5 + f)
        got, got_counts = c.frame_fetch()
        assert c.frame_keypoints() == n
        assert np.array_equal(got_counts, want_counts)
        assert got.tobytes() == want.tobytes()          # bit-identical objects

    # and the planted plane is found where it is: identity pose up to the camera's small motion
    assert len(got) >= 1 and got_counts[0] > 50
    best = got[np.argmax(got["score"])]
    assert best["model"] == 0
    assert np.abs(best["pose"][4:7]).max() < 0.05 and abs(abs(best["pose"][3]) - 1) < 0.01   # model frame = camera frame of frame 0

    # the device-side features are the normalised descriptors of the two-step path
    d_ptr, u_ptr, n_ptr = c.frame_features_dev()
    from moped_amd.pipeline import _DevMem
    dd = torch.as_tensor(_DevMem(d_ptr, (n, 128), "<f4"), device=dev).cpu().numpy()
    assert np.array_equal(dd.view(np.uint32), q_desc.cpu().numpy().view(np.uint32))


def test_image_frame_capacity_and_empty(scene):
    c, torch, *_ = scene
    dev = torch.device("cuda:0")
    prm = capi.default_frame_params()
    flat = torch.full((64, 80), 128, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    c.frame_enqueue_image(flat.data_ptr(), 80, 64, True, CAP, K, CAM0, prm, seed=1)
    objs, counts = c.frame_fetch()
    assert len(objs) == 0 and c.frame_keypoints() == 0 and counts[0] == 0
    # fewer slots than keypoints: the first `cap` of the list are used, nothing overflows
    gray = GOLD["gray0"]
    g = torch.from_numpy(gray).to(dev)
    torch.cuda.synchronize()
    c.frame_enqueue_image(g.data_ptr(), gray.shape[1], gray.shape[0], True, 256, K, CAM0, prm, seed=1)
    objs, counts = c.frame_fetch()
    assert c.frame_keypoints() == 256


def test_batch_of_images_equals_the_images_alone(scene):
    """mh_frame_enqueue_image_batch: FEAT image by image, ONE MATCH launch sequence over all keypoints (every image at
    a stride of the capacity, the rows past its count zeroed before and masked after the search), CLUSTER..FILTER2 image
    after image: every image's objects and counts bit for bit those of mh_frame_enqueue_image on it alone."""
    c, torch, db_desc, db_xyz, model_of = scene
    dev = torch.device("cuda:0")
    frames = [int(x) for x in GOLD["frames"]]
    imgs = [torch.from_numpy(GOLD[f"gray{f}"]).to(dev) for f in frames]
    blank = torch.zeros_like(imgs[0])                      # an image without keypoints in the middle of the batch
    h, w = GOLD["gray0"].shape
    prm = capi.default_frame_params()
    alone = []
    pool = imgs + [blank]
    for i, g in enumerate(pool):
        c.frame_enqueue_image(g.data_ptr(), w, h, True, CAP, K, CAM0, prm, seed=40 + i)
        alone.append(c.frame_fetch())
    order = [0, len(pool) - 1] + list(range(1, len(pool) - 1)) + [0]   # (the blank image second, the first one again last)
    c.reserve(len(order) * CAP)
    batch = [pool[j] for j in order]
    for _ in range(2):
        c.frame_enqueue_image_batch([g.data_ptr() for g in batch], w, h, True, CAP, K, CAM0, prm, [40 + j for j in order])
        for slot, j in enumerate(order):
            got, got_counts = c.frame_fetch_slot(slot)
            want, want_counts = alone[j]
            assert np.array_equal(got_counts, want_counts)
            assert got.tobytes() == want.tobytes()
    assert len(alone[-1][0]) == 0 and len(alone[0][0]) >= 1
    # single images still work on the context afterwards
    c.frame_enqueue_image(imgs[0].data_ptr(), w, h, True, CAP, K, CAM0, prm, seed=40)
    again, _ = c.frame_fetch()
    assert again.tobytes() == alone[0][0].tobytes()


def test_bench_image_leg_finds_the_planted_model_in_every_frame():
    """bench.py's `image_to_objects` leg (what the driver's line reports for FEAT in front of the path): a small run of it
    -- two slots, batches of 16 images -- must find the planar model planted from frame 0's keypoints in every frame of
    the last batch, as its best object."""
    import argparse
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    r = bench.image_to_objects_leg(argparse.Namespace(models=5), frames=64, slots=2, batch=16)
    assert r["value"] > 0 and r["frames"] == 64 and r["keypoints_per_image"] > 300
    assert r["frames_of_the_last_batch_whose_best_object_is_the_planted_model"] == 16
    assert all(n >= 1 for n in r["objects_per_frame_last_batch"]) and 0 < r["sift_alone_ms"] < 5
