"""Phase cycle counts of CLUSTER (meanshift_models_kernel) over one frame of the bench's scene (needs a build with
EXTRA=-DMS_PROF; MH_LIB_PATH): thread 0 of every model's workgroup, summed over the frame's models."""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from moped_amd import synth, capi
from moped_amd.pipeline import FramePipeline, ShardedDB
n_vis = int(sys.argv[1]) if len(sys.argv) > 1 else 2
db = synth.make_db(20, 5000)
dev = torch.device("cuda:0")
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000)
L = capi.load()
out = (C.c_ulonglong * 8)()
fr = synth.make_frame(db, n_vis=n_vis, seed=0)
q = torch.from_numpy(fr.desc).to(dev); uv = torch.from_numpy(fr.uv).to(dev)
for rep in range(3):
    pipe.enqueue(0, q.clone(), uv, seed=rep + 1); objs, counts = pipe.fetch(0); L.mh_debug_ms_prof(out, 1)
names = ["loop top", "(1) means", "(2a) bits", "(2b) walk", "(3) fold", "compact", "emit", "(2') decode"]
print("counts", counts, "CLUSTER phases, cycles of thread 0 summed over the frame's models:", "  ".join(f"{nm}={v}" for nm, v in zip(names, out)), " total", sum(out))
pipe.close()
