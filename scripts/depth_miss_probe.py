import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, torch
from moped_amd import capi, synth
from moped_amd.pipeline import FramePipeline, ShardedDB
db = synth.make_db(50, 5000)
dev = torch.device("cuda:0")
params = capi.default_frame_params()
params.pose1.error_threshold = 8.0
params.f1_min_points, params.f1_feature_distance, params.f1_min_score = 6, 4096.0, 2.0
params.f2_min_points, params.f2_feature_distance, params.f2_min_score = 8, 8192.0, 1e-4
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000, params=params)
c = pipe.ctxs[0]
for s in (int(a) for a in (sys.argv[1:] or range(100))):
    fr = synth.make_frame(db, n_vis=2, seed=s, Q=3000)
    wpts, fill = synth.frame_depth(db, fr, seed=s)
    wgt = (1.0 / (1.0 + (fill / np.float32(0.1)) ** 2)).astype(np.float32)
    d = torch.from_numpy(capi.pack_depth(wpts, wgt).view(np.float32).reshape(-1, 4)).to(dev)
    miss = 0
    for sd in range(12):
        c.frame_set_depth(d.data_ptr(), 1, 0.5)
        pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=1000 + sd)
        objs, counts = pipe.fetch(0)
        if set(objs["model"].tolist()) != set(fr.visible.tolist()):
            miss += 1
            last = (sorted(objs["model"].tolist()), sorted(fr.visible.tolist()), counts.tolist())
    if miss: print("frame seed", s, "misses in", miss, "of 12 seeds", last, flush=True)
    # without depth
c.frame_set_depth(0, 0, 0.5)
print("done")
