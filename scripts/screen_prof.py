"""Where pass B of the two-stage MATCH spends a workgroup's time, and what each part costs (ablations).
Needs a profiling build:  make -C moped_amd/csrc EXTRA=-DSC_PROF OUT=../libmoped_hip_prof.so BUILD=build_prof
then  MH_LIB_PATH=moped_amd/libmoped_hip_prof.so python scripts/screen_prof.py [n_models] [Q]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moped_amd import capi, synth
n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
db = synth.make_db(n_models, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0, Q=Q)
c = capi.Context(0)
c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
dev = torch.device("cuda:0")
q = torch.from_numpy(fr.desc).to(dev)
qn = torch.empty(Q, dtype=torch.float32, device=dev)
o = [torch.empty(Q, dtype=t, device=dev) for t in (torch.int32, torch.float32, torch.float32)]
s = torch.cuda.Stream()
c.set_stream(s.cuda_stream)
c.normalize_dev(q.data_ptr(), qn.data_ptr(), Q)
c.match_set_mode(1)
L = capi.load()
out = (C.c_ulonglong * 8)()
names = {0: "full", 1: "no finish()", 2: "first tile only staged", 4: "no MFMAs", 3: "no finish, no staging", 6: "no MFMA, no staging", 7: "LDS reads only",
         9: "no finish, no tile barrier", 11: "no finish/staging/barrier", 13: "no finish/MFMA/barrier"}
for abl in (0, 1, 9, 11, 4, 13, 7):
    L.mh_debug_screen_prof(out, 1, abl)
    for _ in range(3):
        c.match_local_dev(q.data_ptr(), qn.data_ptr(), Q, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    c.synchronize()
    L.mh_debug_screen_prof(out, 1, abl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record(s)
    for _ in range(reps):
        c.match_local_dev(q.data_ptr(), qn.data_ptr(), Q, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    e1.record(s)
    s.synchronize()
    L.mh_debug_screen_prof(out, 1, abl)
    n = max(out[4], 1)
    cyc, real, wait, tiles, pro = out[0] / n, out[1] / n, out[2] / n, out[3] / n, out[5] / n
    print(f"ablate {abl} ({names[abl]:>24}): match stage {e0.elapsed_time(e1) / reps * 1e3:7.1f} us | pass-B workgroup: {cyc:9.0f} cycles "
          f"= {real / 100:6.1f} us -> {cyc / max(real, 1) * 100:5.0f} MHz; prologue {pro:6.0f}; {tiles:4.1f} tiles, {(cyc - pro) / max(tiles, 1):6.0f} cycles per tile, "
          f"{wait / max(tiles, 1):6.0f} of them in the end-of-tile wait + barrier", flush=True)
