"""image_frame_bench.py with T host threads, each driving depth/T contexts (ctypes releases the GIL inside the C ABI):
is image -> objects bound by the host's launch rate?  usage: image_frame_bench_mt.py [models=20] [depth=16] [frames=3000] [threads=2]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from moped_amd import capi, synth
models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
T = int(sys.argv[4]) if len(sys.argv) > 4 else 2
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
c0.close()
ctxs, streams = [], []
for i in range(depth):
    c = capi.Context(0)
    s = torch.cuda.Stream(device=dev)
    c.set_stream(s.cuda_stream)
    if i == 0: c.db_upload(all_desc, model_of, all_xyz, models + 1)
    else: c.db_share(ctxs[0])
    c.reserve(1024)
    ctxs.append(c); streams.append(s)
imgs = [torch.from_numpy(gold[f"gray{int(f)}"]).to(dev) for f in gold["frames"]]
h, w = gold["gray0"].shape
prm = capi.default_frame_params()
cam = capi.make_cam(K, CAM0)
torch.cuda.synchronize()
def go(mine, k):
    for i in range(k):
        mine[i % len(mine)].frame_enqueue_image(imgs[i % len(imgs)].data_ptr(), w, h, True, 1024, K, CAM0, prm, seed=i + 1, _cam_struct=cam)
        if i % len(mine) == len(mine) - 1 and i < 2 * len(mine):
            for c in mine: c.frame_fetch()
go(ctxs, 4 * depth)
for c in ctxs: c.frame_fetch()
parts = [ctxs[t::T] for t in range(T)]
ths = [threading.Thread(target=go, args=(parts[t], n // T)) for t in range(T)]
t0 = time.perf_counter()
for th in ths: th.start()
for th in ths: th.join()
t_host = time.perf_counter() - t0
for s in streams: s.synchronize()
dt = time.perf_counter() - t0
print(f"image->objects: {n/dt:.1f} frames/s with {T} host thread(s), depth {depth}; host enqueue alone {1e3*t_host/n*T:.3f} ms per frame per thread ({n/t_host:.0f} frames/s enqueue rate)")
for c in ctxs: c.close()
