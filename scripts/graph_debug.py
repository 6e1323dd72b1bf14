import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moped_amd import synth, capi
from moped_amd.pipeline import FramePipeline, ShardedDB
db = synth.make_db(20, 5000)
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000)
dev = torch.device("cuda:0")
fr = synth.make_frame(db, n_vis=2, seed=0)
raw = torch.from_numpy(fr.desc).to(dev); uv = torch.from_numpy(fr.uv).to(dev); work = torch.empty_like(raw)
for it in range(5):
    work.copy_(raw); torch.cuda.synchronize()
    pipe.enqueue(0, work, uv, seed=3)
    try:
        objs, counts = pipe.fetch(0)
        print(it, counts, objs["model"], objs["score"])
    except Exception as e:
        print(it, "ERR", e)
