"""CLUSTER_LINKAGE: the step on the GPU (all models of a frame in one call) vs the oracle on one core."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import orclib
from moped_amd import capi, synth
db = synth.make_db(20, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0)
img, fill = synth.depth_image(db, fr, seed=0, fill_max=0.3)
dev = torch.device("cuda:0")
c = capi.Context(0)
d_img, d_fill = torch.from_numpy(img).to(dev), torch.from_numpy(fill).to(dev)
c.frame_set_depth_image(d_img.data_ptr(), d_fill.data_ptr(), 640, 480, capi.DEPTH_BACKPROJECTION, 0.5, 0.1)
dbn, qn = orclib.normalize(db.desc), orclib.normalize(fr.desc)
idx, d1, d2 = orclib.match_2nn(dbn, qn)
out_q, off = orclib.match_accept(idx, d1, d2, 0.8, db.model_of, db.n_models)
problems = []
for m in range(db.n_models):
    q = out_q[off[m]:off[m + 1]]
    world, _ = orclib.depthmap_prop(img, fill, fr.uv[q], 0.1)
    problems.append((fr.uv[q], db.xyz[idx[q]], world))
c.cluster_linkage(problems)
for _ in range(3): got = c.cluster_linkage(problems)   # (one call per process takes ~40 ms: a one-off of the HIP runtime)
ts = []
for _ in range(11):
    t0 = time.perf_counter(); got = c.cluster_linkage(problems); ts.append(time.perf_counter() - t0)
tg = sorted(ts)[len(ts) // 2]   # median
t0 = time.perf_counter()
want = [orclib.cluster_linkage(uv, mx, wx, img, fill) for uv, mx, wx in problems]
tc = time.perf_counter() - t0
same = all(len(g[0]) == len(w) and all(np.array_equal(a, b) for a, b in zip(g[0], w)) for g, w in zip(got, want))
print(f"20 models, sizes {[len(p[0]) for p in problems]}: GPU step {tg*1e3:.3f} ms (host buffers in and out), oracle 1 core {tc*1e3:.1f} ms; "
      f"clusters {sum(len(w) for w in want)}, identical {same}")
c.close()
