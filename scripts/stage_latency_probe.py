"""Per-stage GPU latency of the frames of slot 0 while `depth` frames are in flight.
usage: stage_latency_probe.py [models] [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moped_amd import synth, capi
from moped_amd.pipeline import FramePipeline, ShardedDB
models = int(sys.argv[1]) if len(sys.argv) > 1 else 3
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
db = synth.make_db(models, 5000)
frames = [synth.make_frame(db, n_vis=2, seed=s) for s in range(8)]
dev = torch.device("cuda:0")
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=depth, max_queries=3000)
q = [torch.from_numpy(f.desc).to(dev) for f in frames]; uv = [torch.from_numpy(f.uv).to(dev) for f in frames]
pipe.ctxs[0].enable_timing(True)
acc = []
for i in range(16): pipe.enqueue(i % depth, q[i % 8], uv[i % 8], seed=i + 1)
pipe.synchronize()
t0 = time.perf_counter(); n = 0
for rep in range(20):
    for i in range(depth):
        pipe.enqueue(i, q[(rep + i) % 8], uv[(rep + i) % 8], seed=rep * depth + i + 1); n += 1
    acc.append(pipe.ctxs[0].timing())     # waits for slot 0's frame only
pipe.synchronize(); dt = time.perf_counter() - t0
keys = list(acc[0].keys())
print(f"models {models} depth {depth}: {n/dt:.0f} frames/s; slot-0 stage latency (ms, median over {len(acc)}):")
print("  " + "  ".join(f"{k[:-3]}={np.median([a[k] for a in acc]):.3f}" for k in keys))
pipe.close()
