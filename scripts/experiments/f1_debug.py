import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from moped_amd import capi, synth
from moped_amd.pipeline import FramePipeline, ShardedDB
db = synth.make_db(20, 5000)
dev = torch.device("cuda:0")
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000)
for sd in range(3):
    fr = synth.make_frame(db, n_vis=2, seed=sd)
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=3)
    objs, counts = pipe.fetch(0)
    print("one frame", counts, objs["model"], objs["score"])
for c in pipe.ctxs: c.pose_set_split(0)
frs = [synth.make_frame(db, n_vis=2, seed=s) for s in range(4)]
qd = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev); uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
pipe2 = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=12000)
for split in (1, 0):
    pipe2.ctxs[0].pose_set_split(split)
    q = qd.clone(); torch.cuda.synchronize()
    pipe2.enqueue_batch(0, q, uv, 4, [3, 4, 5, 6])
    for f, (objs, counts) in enumerate(pipe2.fetch_batch(0, 4)):
        print("batch split", split, "frame", f, counts, objs["model"], objs["score"])
