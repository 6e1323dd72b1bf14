"""FEAT (SIFT) alone and image -> objects at the metric's keypoint count: the bundled frame (~590 keypoints) and
synth.textured_image (~3 200), D frames in flight.  usage: sift_size_probe.py [frames=1200]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from moped_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
only = sys.argv[2] if len(sys.argv) > 2 else ""      # "textured" / "bundled": that image only, SIFT alone at depth 1 (for rocprofv3)
gold = np.load(os.path.join(ROOT, "tests", "golden", "sift_ref_frames.npz"))
dev = torch.device("cuda:0")
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY
db = synth.make_db(20, 5000)
for name, gray in (("bundled frame 0", gold["gray0"]), ("textured image", synth.textured_image(0))):
    if only and not name.startswith(only):
        continue
    h, w = gray.shape
    img = torch.from_numpy(np.ascontiguousarray(gray)).to(dev)
    cap = 4096
    for D in ((1,) if only else (1, 4, 16)):
        ctxs, streams, bufs = [], [], []
        for i in range(D):
            c = capi.Context(0); s = torch.cuda.Stream(device=dev); c.set_stream(s.cuda_stream)
            ctxs.append(c); streams.append(s)
            bufs.append((torch.empty((cap, 128), device=dev), torch.empty((cap, 2), device=dev), torch.zeros(1, dtype=torch.int32, device=dev)))
        def go(k):
            for i in range(k):
                d, xy, cnt = bufs[i % D]
                ctxs[i % D].sift_dev(img.data_ptr(), w, h, True, d.data_ptr(), xy.data_ptr(), 0, cap, cnt.data_ptr())
        go(4 * D); torch.cuda.synchronize()
        t0 = time.perf_counter(); go(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"SIFT alone, {name}: {D:2d} in flight: {1e3 * dt / n:.3f} ms per image ({n / dt:.0f} images/s); keypoints {int(bufs[0][2].item())}", flush=True)
        for c in ctxs: c.close()
    if only:
        continue
    # image -> objects: the image's own keypoints as a 21st planar model (as scripts/image_frame_bench.py does)
    c0 = capi.Context(0)
    xy, _, desc = c0.sift(gray)
    n_all = len(xy)
    if n_all > 1000:   # one planar object in the middle of the image, not the whole image (a model of 3 000 matches in one cluster
        keep = (np.abs(xy[:, 0] - w / 2) < 110) & (np.abs(xy[:, 1] - h / 2) < 90)   # is past what CLUSTER / POSE reserve per model)
        xy, desc = xy[keep], desc[keep]
    z = np.float32(0.8)
    xyz = np.stack([(xy[:, 0] - K[2]) / K[0] * z, (xy[:, 1] - K[3]) / K[1] * z, np.full(len(xy), z)], 1).astype(np.float32)
    all_desc = c0.normalize(np.concatenate([db.desc, desc]))
    all_xyz = np.concatenate([db.xyz, xyz]); model_of = np.concatenate([db.model_of, np.full(len(xy), 20, np.int32)])
    c0.close()
    D = 16
    ctxs, streams = [], []
    for i in range(D):
        c = capi.Context(0); s = torch.cuda.Stream(device=dev); c.set_stream(s.cuda_stream)
        if i == 0: c.db_upload(all_desc, model_of, all_xyz, 21)
        else: c.db_share(ctxs[0])
        c.reserve(cap)
        ctxs.append(c); streams.append(s)
    prm = capi.default_frame_params()
    def run(k):
        for i in range(k):
            ctxs[i % D].frame_enqueue_image(img.data_ptr(), w, h, True, cap, K, CAM0, prm, i + 1)
    run(2 * D); torch.cuda.synchronize()
    m = max(200, n // 2)
    t0 = time.perf_counter(); run(m); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    objs, counts = ctxs[(m - 1) % D].frame_fetch()
    print(f"image -> objects, {name}: {m / dt:.0f} frames/s ({1e3 * dt / m:.3f} ms per frame), 16 in flight, keypoints {n_all} ({len(xy)} of them the planted planar model's), counts {counts.tolist()}, objects {len(objs)}", flush=True)
    for c in ctxs: c.close()
