"""Throughput of CLUSTER..FILTER2 alone (match results precomputed) vs frames in flight.
usage: rest_only_probe.py [models] [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moped_amd import synth, capi
from moped_amd.pipeline import FramePipeline, ShardedDB
models = int(sys.argv[1]) if len(sys.argv) > 1 else 3
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
db = synth.make_db(models, 5000)
frames = [synth.make_frame(db, n_vis=2, seed=s) for s in range(8)]
dev = torch.device("cuda:0")
prm = capi.default_frame_params()
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=depth, max_queries=3000, params=prm)
q = [torch.from_numpy(f.desc).to(dev) for f in frames]; uv = [torch.from_numpy(f.uv).to(dev) for f in frames]
top2 = [torch.empty(3 * 3000, dtype=torch.int32, device=dev) for _ in frames]
for i in range(8):
    pipe.ctxs[0].frame_enqueue_match_local(q[i].data_ptr(), 3000, top2[i].data_ptr())
pipe.synchronize()
cam = capi.make_cam(pipe.K, pipe.cam)
def go(n, what):
    for i in range(n):
        s = i % depth
        if what == "rest":
            pipe.ctxs[s].frame_enqueue_rest(uv[i % 8].data_ptr(), 3000, top2[i % 8].data_ptr(), 1, pipe.K, pipe.cam, prm, i + 1, _cam_struct=cam)
        else:
            pipe.enqueue(s, q[i % 8], uv[i % 8], seed=i + 1)
for what in ("rest", "full"):
    go(32, what); pipe.synchronize()
    n = 320
    t0 = time.perf_counter(); go(n, what); th = time.perf_counter() - t0; pipe.synchronize(); dt = time.perf_counter() - t0
    objs, counts = pipe.fetch(0)
    print(f"models {models} depth {depth:2d} {what:5s}: {n/dt:8.1f} frames/s ({1e3*dt/n:.3f} ms/frame, host {1e3*th/n:.3f}) objects {len(objs)} counts {counts}")
pipe.close()
