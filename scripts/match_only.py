"""Run only the match stage (pack + match_kernel + combine) N times: a short target
for rocprofv3 counter passes.  usage: match_only.py [n_models] [Q] [reps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moped_amd import capi, synth
n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
db = synth.make_db(n_models, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0, Q=Q)
c = capi.Context(0)
dbn = c.normalize(db.desc)
c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
dev = torch.device("cuda:0")
q = torch.from_numpy(fr.desc).to(dev)
qn = torch.empty(Q, dtype=torch.float32, device=dev)
idx = torch.empty(Q, dtype=torch.int32, device=dev)
d1 = torch.empty(Q, dtype=torch.float32, device=dev)
d2 = torch.empty(Q, dtype=torch.float32, device=dev)
s = torch.cuda.Stream()
c.set_stream(s.cuda_stream)
c.normalize_dev(q.data_ptr(), qn.data_ptr(), Q)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(2):
    e0.record(s)
    for _ in range(reps):
        c.match_local_dev(q.data_ptr(), qn.data_ptr(), Q, idx.data_ptr(), d1.data_ptr(), d2.data_ptr())
    e1.record(s)
    s.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"match stage: {ms:.4f} ms  {2*128*Q*db.n/ms/1e9:.1f} TFLOP/s  N={db.n} Q={Q}")
