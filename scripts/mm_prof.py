"""Phase cycle counts of the MFMA match kernel per workgroup (needs a build with EXTRA=-DMM_PROF)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moped_amd import capi, synth
n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Q = 3000
db = synth.make_db(n_models, 5000); fr = synth.make_frame(db, n_vis=2, seed=0, Q=Q)
c = capi.Context(0); dbn = c.normalize(db.desc); c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
dev = torch.device("cuda:0")
q = torch.from_numpy(fr.desc).to(dev); qn = torch.empty(Q, dtype=torch.float32, device=dev)
idx = torch.empty(Q, dtype=torch.int32, device=dev); d1 = torch.empty(Q, dtype=torch.float32, device=dev); d2 = torch.empty(Q, dtype=torch.float32, device=dev)
c.normalize_dev(q.data_ptr(), qn.data_ptr(), Q)
L = capi.load(); out = (C.c_ulonglong * 8)()
for rep in range(3):
    c.match_local_dev(q.data_ptr(), qn.data_ptr(), Q, idx.data_ptr(), d1.data_ptr(), d2.data_ptr()); c.synchronize(); L.mh_debug_mm_prof(out, 1)
n = max(out[7], 1)
print(f"{n} workgroups; cycles per workgroup: A operands {out[0]//n}, first tile {out[1]//n}, tile loop {out[2]//n}, reduce+write {out[3]//n}")
