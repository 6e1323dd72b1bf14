"""Phase cycle counts of the pose kernel over one frame (needs a build with EXTRA=-DPOSE_PROF)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moped_amd import synth, capi
from moped_amd.pipeline import FramePipeline, ShardedDB
db = synth.make_db(20, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0)
dev = torch.device("cuda:0")
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000)
q = torch.from_numpy(fr.desc).to(dev); uv = torch.from_numpy(fr.uv).to(dev)
L = capi.load()
out = (C.c_ulonglong * 8)()
for rep in range(3):
    pipe.enqueue(0, q.clone(), uv, seed=rep + 1); pipe.fetch(0); L.mh_debug_pose_prof(out, 1)
names = ["load+distinct", "hypotheses", "argmax", "inlier list", "LM plain", "LM squared", "-", "tasks"]
n = max(out[7], 1)
print(f"tasks {out[7]} (POSE + POSE2); cycles per task: " + "  ".join(f"{nm}={v // n}" for nm, v in zip(names[:6], out)))
pipe.close()
