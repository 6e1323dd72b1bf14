"""Two-stage MATCH vs the exact kernels: per-launch time of the whole match stage and, with rocprofv3
--kernel-trace --stats around this script, the split over prepare / pass A / pass B / pass C.
usage: screen_probe.py [n_models] [Q] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moped_amd import capi, synth
n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
db = synth.make_db(n_models, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0, Q=Q)
c = capi.Context(0)
c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
dev = torch.device("cuda:0")
q = torch.from_numpy(fr.desc).to(dev)
qn = torch.empty(Q, dtype=torch.float32, device=dev)
out = {m: [torch.empty(Q, dtype=t, device=dev) for t in (torch.int32, torch.float32, torch.float32)] for m in (0, 1)}
s = torch.cuda.Stream()
c.set_stream(s.cuda_stream)
c.normalize_dev(q.data_ptr(), qn.data_ptr(), Q)
for mode in (0, 1):
    c.match_set_mode(mode)
    o = out[mode]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(2):
        c.match_stats(reset=True)
        e0.record(s)
        for _ in range(reps):
            c.match_local_dev(q.data_ptr(), qn.data_ptr(), Q, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
        e1.record(s)
        s.synchronize()
    ms = e0.elapsed_time(e1) / reps
    st = c.match_stats(Q)
    print(f"mode {mode} ({'two-stage' if mode else 'exact f32'}): {ms:.4f} ms/launch  {2*128*Q*db.n/ms/1e9:.1f} TFLOP/s (algorithmic)  "
          f"N={db.n} Q={Q}  stats={st}", flush=True)
same = all(torch.equal(a, b) for a, b in zip(out[0], out[1]))
print("bit-identical:", same)
