"""How long does the host take to enqueue one frame (no GPU wait)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moped_amd import synth
from moped_amd.pipeline import FramePipeline, ShardedDB
db = synth.make_db(20, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0)
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=4, max_queries=3000)
dev = torch.device("cuda:0")
q = [torch.from_numpy(fr.desc).to(dev) for _ in range(4)]
uv = torch.from_numpy(fr.uv).to(dev)
for i in range(8):
    pipe.enqueue(i % 4, q[i % 4], uv, seed=i + 1)
pipe.synchronize()
n = 64
t0 = time.perf_counter()
for i in range(n):
    pipe.enqueue(i % 4, q[i % 4], uv, seed=i + 1)
t1 = time.perf_counter()
pipe.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e6*(t1-t0)/n:.1f} us/frame ; total {1e6*(t2-t0)/n:.1f} us/frame")
c = pipe.ctxs[0]
c.enable_timing(True)
pipe.enqueue(0, q[0], uv, seed=1)
pipe.fetch(0)
print(c.timing())
