"""Pass B (and the other kernels of the two-stage MATCH) at the judged launch shape, isolated, for the library in
MH_LIB_PATH: 16 distinct frames x 3000 queries against the 20-model DB (or argv: models frames), the library's own HIP
events around every kernel (mh_match_timing), median of `reps` launch sequences.  For A/B runs of kernel builds on ONE box:
  for r in 1 2 3; do for l in a b; do MH_LIB_PATH=$PWD/moped_amd/libmoped_hip_$l.so python scripts/passb_ab.py; done; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moped_amd import capi, synth
models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
Q = 3000 * B
db = synth.make_db(models, 5000)
frs = [synth.make_frame(db, n_vis=2, seed=s, Q=3000) for s in range(B)]
c = capi.Context(0)
c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
c.reserve(Q)
dev = torch.device("cuda:0")
q = torch.from_numpy(np.concatenate([f.desc for f in frs])).to(dev)
qn = torch.empty(Q, dtype=torch.float32, device=dev)
o = [torch.empty(Q, dtype=t, device=dev) for t in (torch.int32, torch.float32, torch.float32)]
s = torch.cuda.Stream(); c.set_stream(s.cuda_stream)
c.normalize_dev(q.data_ptr(), qn.data_ptr(), Q)
for _ in range(3): c.match_local_dev(q.data_ptr(), qn.data_ptr(), Q, *[x.data_ptr() for x in o])
s.synchronize()
c.enable_timing(True)
rows = []
for _ in range(reps):
    c.match_local_dev(q.data_ptr(), qn.data_ptr(), Q, *[x.data_ptr() for x in o])
    s.synchronize()
    rows.append(list(c.match_timing().values()))
c.enable_timing(False)
med = np.median(np.array(rows), axis=0)
st = c.match_stats(Q)
flops = 2.0 * 128 * Q * db.n
print(f"{os.path.basename(os.environ.get('MH_LIB_PATH', 'libmoped_hip.so')):32s} models {models} Q {Q}: pass B {med[3]:.4f} ms = {flops / med[3] / 1e9 / 2500:.4f} of 2.5 PFLOP/s; "
      f"query image {med[0]:.4f} pass A {med[1]:.4f} thresholds {med[2]:.4f} pass C {med[4]:.4f}; checksum {int(o[0].sum().item())} {float(o[1].sum().item()):.6f}", flush=True)
c.close()
