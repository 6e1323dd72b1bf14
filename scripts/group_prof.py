import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from moped_amd import synth, capi
from moped_amd.pipeline import FramePipeline, ShardedDB
db = synth.make_db(20, 5000)
dev = torch.device("cuda:0")
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000)
L = capi.load()
out = (C.c_ulonglong * 8)()
fr = synth.make_frame(db, n_vis=2, seed=0)
q = torch.from_numpy(fr.desc).to(dev); uv = torch.from_numpy(fr.uv).to(dev)
for rep in range(3):
    pipe.enqueue(0, q.clone(), uv, seed=rep + 1); objs, counts = pipe.fetch(0); L.mh_debug_group_prof(out, 1)
print("counts", counts, "group phases (cycles): init+merge", out[0], "compaction", out[1], "depthfilter2", out[2], "scan", out[3], "placement", out[4], "representatives", out[5])
pipe.close()
