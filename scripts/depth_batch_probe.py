"""Config 5 (depth attributes per query): do the frames of a batch of 4 report what they report one at a time?  (debugging aid)
usage: depth_batch_probe.py [models=50] [depth_kind=1]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch
import bench
from moped_amd import synth

models = int(sys.argv[1]) if len(sys.argv) > 1 else 50
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
env = {"rank": 0, "world": 1, "dev": dev, "local_rank": 0}
db = synth.make_db(models, 5000)
res = {}
for B in (4, 1):
    a = bench.parse(["--models", str(models), "--depth-kind", str(kind), "--frames-per-step", "32", "--batch", str(B)])
    a.depth = 16
    job = bench.Job(a, env, db, models, False, False, B, 32)
    pipe = job.pipe
    out = []
    for i in range(32):
        slot = i % 16
        seed = 500 + i
        if B == 4:
            if i % 4:
                continue
            pg = i // 4
            with torch.cuda.stream(pipe.streams[slot]):
                job.work_b[slot].copy_(job.pristine_b[pg], non_blocking=True)
            pipe.ctxs[slot].frame_set_depth(job.depths_b[pg].data_ptr(), kind, 0.5)
            pipe.enqueue_batch(slot, job.work_b[slot], job.uv_b[pg], 4, [seed + f for f in range(4)])
            out += [r[0] for r in pipe.fetch_batch(slot, 4)]
        else:
            with torch.cuda.stream(pipe.streams[slot]):
                job.work[slot].copy_(job.pristine[i], non_blocking=True)
            pipe.ctxs[slot].frame_set_depth(job.depths[i].data_ptr(), kind, 0.5)
            pipe.enqueue(slot, job.work[slot], job.uvs[i], seed=seed)
            out.append(pipe.fetch(slot)[0])
    res[B] = out
    job.close()
bad = 0
for i, (x, y) in enumerate(zip(res[4], res[1])):
    same = len(x) == len(y) and np.array_equal(x["model"], y["model"]) and np.array_equal(x["pose"].view(np.uint32), y["pose"].view(np.uint32))
    if not same or len(x) != 2:
        bad += not same
        print(f"frame {i}: batch of 4 -> {len(x)} objects {x['model'].tolist()}, alone -> {len(y)} objects {y['model'].tolist()}{'' if same else '   DIFFERENT'}")
print(f"{bad} of 32 frames differ between the batch and the single frame")
