"""Image in, objects out: frames/s of FEAT(SIFT) -> MATCH -> ... -> FILTER2 on the reference's bundled
640x480 frames (~590 keypoints each) against the planar model of frame 0 + a synthetic N-model DB.
usage: image_frame_bench.py [models=20] [depth=4] [frames=400] [images per MATCH launch sequence=1]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # one hardware queue per frame slot (the default of 4 caps the overlap at 4 kernels)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from moped_amd import capi, synth
models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 400
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
gold = np.load(os.path.join(ROOT, "tests", "golden", "sift_ref_frames.npz"))
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY
dev = torch.device("cuda:0")
db = synth.make_db(models, 5000)
c0 = capi.Context(0)
xy, _, desc = c0.sift(gold["gray0"])
z = np.float32(0.8)
xyz = np.stack([(xy[:, 0] - K[2]) / K[0] * z, (xy[:, 1] - K[3]) / K[1] * z, np.full(len(xy), z)], 1).astype(np.float32)
all_desc = c0.normalize(np.concatenate([db.desc, desc]))
all_xyz = np.concatenate([db.xyz, xyz])
model_of = np.concatenate([db.model_of, np.full(len(xy), models, np.int32)])
if os.environ.get("IFB_NO_OBJECT"):   # the frame's object is not in the DB: FEAT + MATCH alone (nothing to cluster or pose)
    all_desc[len(db.desc):] = c0.normalize(np.random.default_rng(5).standard_normal(desc.shape).astype(np.float32))
c0.close()
ctxs, streams = [], []
for i in range(depth):
    c = capi.Context(0)
    s = torch.cuda.Stream(device=dev)
    c.set_stream(s.cuda_stream)
    c.db_upload(all_desc, model_of, all_xyz, models + 1)
    c.reserve(1024 * B)
    ctxs.append(c); streams.append(s)
imgs = [torch.from_numpy(gold[f"gray{int(f)}"]).to(dev) for f in gold["frames"]]
h, w = gold["gray0"].shape
prm = capi.default_frame_params()
cam = capi.make_cam(K, CAM0)
torch.cuda.synchronize()
def go(k):
    if B > 1:   # mh_frame_enqueue_image_batch: B images per MATCH launch sequence
        for g in range(k // B):
            ptrs = [imgs[(g * B + j) % len(imgs)].data_ptr() for j in range(B)]
            ctxs[g % depth].frame_enqueue_image_batch(ptrs, w, h, True, 1024, K, CAM0, prm, [g * B + j + 1 for j in range(B)], _cam_struct=cam)
        return
    for i in range(k):
        ctxs[i % depth].frame_enqueue_image(imgs[i % len(imgs)].data_ptr(), w, h, True, 1024, K, CAM0, prm, seed=i + 1, _cam_struct=cam)
        if i % depth == depth - 1 and i < 2 * depth:   # early fetches teach the contexts the keypoint count
            for c in ctxs: c.frame_fetch()
go(4 * depth * B)
for c in ctxs: c.frame_fetch_slot(0) if B > 1 else c.frame_fetch()
t0 = time.perf_counter(); go(n)
for s in streams: s.synchronize()
dt = time.perf_counter() - t0
objs, counts = ctxs[(n // B - 1) % depth].frame_fetch_slot(B - 1) if B > 1 else ctxs[(n - 1) % depth].frame_fetch()
print(f"image->objects: {n/dt:.1f} frames/s ({1e3*dt/n:.3f} ms/frame), depth {depth}" + (f" x {B} images per MATCH launch sequence" if B > 1 else "") + f", DB {models}+1 models / {len(all_desc)} rows, "
      f"keypoints {ctxs[(n-1)%depth].frame_keypoints() if B == 1 else '-'}, counts {counts.tolist()}, objects {len(objs)} best model {objs[np.argmax(objs['score'])]['model'] if len(objs) else None}")
c = ctxs[0]
c.enable_timing(True)
c.frame_enqueue_image(imgs[0].data_ptr(), w, h, True, 1024, K, CAM0, prm, seed=1)
c.frame_fetch()
print("stage ms:", c.timing())
for c in ctxs: c.close()
