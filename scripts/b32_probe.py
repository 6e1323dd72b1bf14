"""Which frames of a 32-frame batch lose objects? (debugging aid)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from moped_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Q = 3000
db = synth.make_db(20, 5000)
dev = torch.device("cuda:0")
frs = [synth.make_frame(db, n_vis=2, seed=s, Q=Q) for s in range(B)]
c = capi.Context(0)
c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
c.reserve_batch(Q, B)
prm = capi.default_frame_params()
qd0 = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
for rep in range(4):
    qd = qd0.clone()
    c.frame_enqueue_batch(qd.data_ptr(), uv.data_ptr(), Q, B, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, list(range(1, B + 1)))
    res = [c.frame_fetch_slot(f) for f in range(B)]
    print("rep", rep, "objects", [len(r[0]) for r in res])
    print("       counts[2] (after POSE)", [int(r[1][2]) for r in res], "counts[3] (after FILTER)", [int(r[1][3]) for r in res], "clusters", [int(r[1][1]) for r in res])
c.close()
