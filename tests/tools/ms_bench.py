import sys, time, numpy as np
import os
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'oracle'))
import orclib
from moped_amd import capi, synth
ctx = capi.Context(0)
rng = np.random.default_rng(0)
def t(pts, label):
    ctx.meanshift(pts)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); got, _ = ctx.meanshift(pts); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]   # median: a call that grows the context's buffers or meets a host hiccup is not the kernel
    t0 = time.perf_counter(); want, it = orclib.meanshift(pts); dc = time.perf_counter() - t0
    print(f"{label}: n={len(pts)} gpu {dt*1e3:.3f} ms (max of 7 calls {max(ts)*1e3:.3f})  cpu {dc*1e3:.3f} ms iters={it} clusters={len(want)}")
for n in (16, 64, 150, 300):
    t(rng.normal([320, 240], 6, size=(n, 2)).astype(np.float32), "tight")
    t(rng.uniform([0, 0], [640, 480], size=(n, 2)).astype(np.float32), "uniform")
db = synth.make_db(20, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0)
dbn = orclib.normalize(db.desc); qn = orclib.normalize(fr.desc)
idx, d1, d2 = orclib.match_2nn(dbn, qn)
out_q, off = orclib.match_accept(idx, d1, d2, 0.8, db.model_of, 20)
for m in range(20):
    q = out_q[off[m]:off[m+1]]
    if len(q) > 20: t(fr.uv[q], f"model{m}")
