import sys, os, copy
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, torch
import orclib
from moped_amd import capi, synth
from moped_amd.pipeline import FramePipeline, ShardedDB
np.set_printoptions(precision=4, suppress=True, linewidth=200)
db = synth.make_db(50, 5000)
dev = torch.device("cuda:0")
params = capi.default_frame_params()
params.pose1.error_threshold = 8.0
params.f1_min_points, params.f1_feature_distance, params.f1_min_score = 6, 4096.0, 2.0
params.f2_min_points, params.f2_feature_distance, params.f2_min_score = 8, 8192.0, 1e-4
s = int(sys.argv[1]) if len(sys.argv) > 1 else 6
fr = synth.make_frame(db, n_vis=2, seed=s, Q=3000)
wpts, fill = synth.frame_depth(db, fr, seed=s)
wgt = (1.0 / (1.0 + (fill / np.float32(0.1)) ** 2)).astype(np.float32)
d = torch.from_numpy(capi.pack_depth(wpts, wgt).view(np.float32).reshape(-1, 4)).to(dev)
print("visible", fr.visible, "planted poses", fr.poses)
for stage2 in (0, 1):
    p = copy.copy(params); p.run_stage2 = stage2
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000, params=p)
    c = pipe.ctxs[0]
    c.frame_set_depth(d.data_ptr(), 1, 0.5)
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=1000)
    objs, counts = pipe.fetch(0)
    print("run_stage2", stage2, "counts", counts.tolist())
    for o in objs:
        m = int(o["model"])
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier & (db.model_of[np.maximum(fr.src_point, 0)] == m))[0]
        e = np.sqrt(((orclib.project(o["pose"], db.xyz[fr.src_point[rows]], synth.K_DEFAULT, synth.CAM_IDENTITY) - fr.uv[rows]) ** 2).sum(1)) if len(rows) else np.zeros(1)
        print("  model", m, "pose", o["pose"], "n", int(o["n_points"]), "score", float(o["score"]), "mean reproj on planted", float(e.mean()), "max", float(e.max()))
    pipe.close()
