"""Full frames on a small shard, for rocprofv3 --kernel-trace: usage trace_probe.py [models] [depth] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moped_amd import synth, capi
from moped_amd.pipeline import FramePipeline, ShardedDB
models = int(sys.argv[1]) if len(sys.argv) > 1 else 3
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 160
db = synth.make_db(models, 5000)
frames = [synth.make_frame(db, n_vis=2, seed=s) for s in range(8)]
dev = torch.device("cuda:0")
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=depth, max_queries=3000)
q = [torch.from_numpy(f.desc).to(dev) for f in frames]; uv = [torch.from_numpy(f.uv).to(dev) for f in frames]
for rep in range(2):
    pipe.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        pipe.enqueue(i % depth, q[i % 8], uv[i % 8], seed=i + 1)
    pipe.synchronize()
    dt = time.perf_counter() - t0
print(f"{n/dt:.1f} frames/s ({1e3*dt/n:.3f} ms/frame)")
pipe.close()
