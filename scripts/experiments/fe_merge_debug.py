"""Merged front-end batch against the frames alone: counts per frame (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from moped_amd import capi, synth, moped3d
dev = torch.device("cuda:0")
db = synth.make_db(6, 1500, seed=2)
n_vis = (2, 0, 3, 1)
B, Q = len(n_vis), 1500
frs = [synth.make_frame(db, n_vis=n, seed=60 + i, Q=Q, pts_per_obj=130) for i, n in enumerate(n_vis)]
maps = []
for i, f in enumerate(frs):
    img, fill = synth.depth_image(db, f, seed=10 + i, fill_max=0.3)
    maps.append((torch.from_numpy(img).to(dev), torch.from_numpy(fill).to(dev)))
c = capi.Context(0)
c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
c.reserve(B * Q)
table = moped3d.ratio_table(db.xyz, db.model_of, db.n_models, synth.K_DEFAULT)
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode in ("all", "rules"):
    c.frame_set_depth_rules(synth.K_DEFAULT, 64, 0.05, 0.01, table)
if mode in ("all", "linkage"):
    c.frame_set_cluster_linkage(capi.default_linkage_params())
prm = capi.default_frame_params()
for i, f in enumerate(frs):
    c.frame_set_depth_image(maps[i][0].data_ptr(), maps[i][1].data_ptr(), 640, 480, capi.DEPTH_BACKPROJECTION, 0.5, 0.1)
    qd, uv = torch.from_numpy(f.desc).to(dev), torch.from_numpy(f.uv).to(dev)
    c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, 5 + i)
    objs, counts = c.frame_fetch()
    print("alone", i, counts, objs["model"])
uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
c.frame_set_depth_image_batch([m[0].data_ptr() for m in maps], [m[1].data_ptr() for m in maps], 640, 480,
                              capi.DEPTH_BACKPROJECTION, 0.5, 0.1)
for rep in range(2):
    qd = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
    c.frame_enqueue_batch(qd.data_ptr(), uv.data_ptr(), Q, B, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, [5 + i for i in range(B)])
    for f in range(B):
        try:
            objs, counts = c.frame_fetch_slot(f)
            print("batch rep", rep, "frame", f, counts, objs["model"])
        except Exception as e:
            print("batch rep", rep, "frame", f, "ERR", e)
