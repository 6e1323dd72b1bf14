"""Does a stream of small dependent launches on OTHER streams slow the match kernel down?  The match stage runs back to
back on one stream (8 frames per launch against a 3-model shard, the per-rank load of 8 GPUs) while S other streams each
issue tiny kernels (normalize of 64 rows) one after the other.  usage: launch_pressure_probe.py [streams] [tiny launches per match]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moped_amd import capi, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 15
per = int(sys.argv[2]) if len(sys.argv) > 2 else 48
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
db = synth.make_db(3, 5000)
Q = 24000
fr = synth.make_frame(db, n_vis=2, seed=0, Q=3000)
dev = torch.device("cuda:0")
c = capi.Context(0); dbn = c.normalize(db.desc); c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
q = torch.from_numpy(np.tile(fr.desc, (8, 1))).to(dev); qn = torch.empty(Q, dtype=torch.float32, device=dev)
idx = torch.empty(Q, dtype=torch.int32, device=dev); d1 = torch.empty(Q, dtype=torch.float32, device=dev); d2 = torch.empty_like(d1)
ms = torch.cuda.Stream(); c.set_stream(ms.cuda_stream); c.normalize_dev(q.data_ptr(), qn.data_ptr(), Q)
small = []
for i in range(S):
    cc = capi.Context(0); s = torch.cuda.Stream(); cc.set_stream(s.cuda_stream)
    a = torch.rand(64, 128, device=dev); b = torch.empty(64, device=dev)
    small.append((cc, s, a, b))
torch.cuda.synchronize()
def run(n_match, pressure):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ms)
    for k in range(n_match):
        c.match_local_dev(q.data_ptr(), qn.data_ptr(), Q, idx.data_ptr(), d1.data_ptr(), d2.data_ptr())
        if pressure:
            for j in range(per):
                cc, s, a, b = small[(k * per + j) % S]
                cc.normalize_dev(a.data_ptr(), b.data_ptr(), 64)
    e1.record(ms)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_match
run(5, True)
for pressure in (False, True, False, True):
    print(f"tiny launches on {S} other streams: {'yes' if pressure else 'no '}  match stage {run(40, pressure):.4f} ms per launch of 8 frames")
