"""Timeline of two wavefronts (0 and 4: the two of SIMD 0) of one pass-B workgroup: cycles between events.
Needs the SC_PROF build (see scripts/screen_prof.py)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moped_amd import capi, synth
n_models, Q = 20, 3000
db = synth.make_db(n_models, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0, Q=Q)
c = capi.Context(0)
c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
dev = torch.device("cuda:0")
q = torch.from_numpy(fr.desc).to(dev)
qn = torch.empty(Q, dtype=torch.float32, device=dev)
o = [torch.empty(Q, dtype=t, device=dev) for t in (torch.int32, torch.float32, torch.float32)]
c.normalize_dev(q.data_ptr(), qn.data_ptr(), Q)
c.match_set_mode(1)
L = capi.load()
for _ in range(3):
    c.match_local_dev(q.data_ptr(), qn.data_ptr(), Q, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
c.synchronize()
buf = (C.c_ulonglong * 2048)()
L.mh_debug_screen_trace(buf)
tr = np.frombuffer(buf, dtype=np.uint64).reshape(2, 1024)
names = {1: "tile begins", 2: "staging issued", 3: "LDS reads issued+landed", 4: "chain issued", 5: "prev block finished",
         6: "tile work issued", 7: "own DMA landed", 8: "barrier passed"}
for w in range(2):
    ev = (tr[w] >> np.uint64(56)).astype(int)
    t = (tr[w] & np.uint64((1 << 56) - 1)).astype(np.int64)
    n = int((ev != 0).sum())
    print(f"--- wavefront {4 * w}: {n} events, {t[n - 1]} cycles")
    # third and fourth tile in detail, then the per-event-type averages
    starts = [i for i in range(n) if ev[i] == 1]
    for i in range(starts[2], starts[4] if len(starts) > 4 else n):
        print(f"   +{t[i] - t[i - 1]:6d}  {names[ev[i]]}")
    tot = {}
    for i in range(1, n):
        tot.setdefault(ev[i], []).append(t[i] - t[i - 1])
    for e, v in sorted(tot.items()):
        print(f"  event {e} ({names[e]:>24}): n={len(v):4d} mean {np.mean(v):8.1f} cycles  sum {np.sum(v):9d}")
