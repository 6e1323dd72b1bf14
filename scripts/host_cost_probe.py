"""Host time per frame spent inside the enqueue calls vs the wall time per frame of the pipeline.
usage: host_cost_probe.py [batch] [depth] [frames]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from moped_amd import synth
from moped_amd.pipeline import FramePipeline, ShardedDB
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
Q = 3000
db = synth.make_db(20, 5000)
dev = torch.device("cuda:0")
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=depth, max_queries=Q * B)
frs = [synth.make_frame(db, n_vis=2, seed=s, Q=Q) for s in range(B)]
qd0 = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
work = [torch.empty_like(qd0) for _ in range(depth)]
def go(groups):
    host = 0.0
    for g in range(groups):
        slot = g % depth
        with torch.cuda.stream(pipe.streams[slot]):
            work[slot].copy_(qd0, non_blocking=True)
        t = time.perf_counter()
        if B > 1:
            pipe.enqueue_batch(slot, work[slot], uv, B, list(range(g * B + 1, g * B + B + 1)))
        else:
            pipe.enqueue(slot, work[slot], uv, seed=g + 1)
        host += time.perf_counter() - t
    return host
go(4 * depth)
pipe.synchronize()
t0 = time.perf_counter()
host = go(n // B)
t_issue = time.perf_counter() - t0
pipe.synchronize()
wall = time.perf_counter() - t0
print(f"B={B} depth={depth}: wall {1e6 * wall / n:.1f} us/frame ({n / wall:.0f} frames/s); host inside enqueue {1e6 * host / n:.1f} us/frame; "
      f"issue loop {1e6 * t_issue / n:.1f} us/frame", flush=True)
pipe.close()
