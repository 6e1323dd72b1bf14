"""What a slot of the pipeline holds in HBM: free memory before / after every step of setting one up (debugging aid).
usage: hbm_footprint_probe.py [models=20] [B=8]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from moped_amd import capi, synth
models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev); torch.cuda.synchronize()
def free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(dev)[0] / 2 ** 20
db = synth.make_db(models, 5000)
f0 = free()
c = capi.Context(0)
f1 = free(); print(f"context: {f0 - f1:.0f} MB")
c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
f2 = free(); print(f"database ({db.n} rows): {f1 - f2:.0f} MB")
c.reserve_batch(3000, B)
f3 = free(); print(f"reserve_batch(3000, {B}): {f2 - f3:.0f} MB")
c2 = capi.Context(0)
c2.db_share(c)
c2.reserve_batch(3000, B)
f4 = free(); print(f"a second slot sharing the database: {f3 - f4:.0f} MB")
c3 = capi.Context(0)
c3.db_share(c)
c3.reserve(3000)
f5 = free(); print(f"a slot for single frames (reserve(3000)): {f4 - f5:.0f} MB")
