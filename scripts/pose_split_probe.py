"""Where does the POSE stage spend its time?  Stage latency (depth 1) for parameter variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moped_amd import synth, capi
from moped_amd.pipeline import FramePipeline, ShardedDB
db = synth.make_db(20, 5000)
frames = [synth.make_frame(db, n_vis=2, seed=s) for s in range(8)]
dev = torch.device("cuda:0")
q = [torch.from_numpy(f.desc).to(dev) for f in frames]; uv = [torch.from_numpy(f.uv).to(dev) for f in frames]
for label, H, l2, l4 in (("default", 1024, 10, 10), ("H=64", 64, 10, 10), ("no LM", 1024, 0, 0), ("LM 10/0", 1024, 10, 0), ("H=64 no LM", 64, 0, 0)):
    prm = capi.default_frame_params()
    for p in (prm.pose1, prm.pose2):
        p.n_hypotheses, p.lm_iters_l2, p.lm_iters_l4 = H, l2, l4
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000, params=prm)
    pipe.ctxs[0].enable_timing(True)
    acc = []
    for i in range(24):
        pipe.enqueue(0, q[i % 8].clone(), uv[i % 8], seed=i + 1)
        t = pipe.ctxs[0].timing()
        if i >= 8: acc.append(t)
    print(f"{label:12s} " + "  ".join(f"{k[:-3]}={np.median([a[k] for a in acc]):.3f}" for k in acc[0]))
    pipe.close()
