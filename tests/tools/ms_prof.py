"""Phase cycle counts of the mean-shift kernel (needs a build with EXTRA=-DMS_PROF)."""
import sys, os, ctypes as C, numpy as np
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'oracle'))
from moped_amd import capi
ctx = capi.Context(0)
L = capi.load()
rng = np.random.default_rng(0)
names = ["loop top", "(1) means", "(2a) bits", "(2b) walk", "(3) fold", "compact", "emit", "(2') decode"]
def prof(pts, label):
    out = (C.c_ulonglong * 8)()
    ctx.meanshift(pts); L.mh_debug_ms_prof(out, 1)
    ctx.meanshift(pts); L.mh_debug_ms_prof(out, 1)
    tot = sum(out)
    print(f"{label} n={len(pts)}: total {tot} cycles  " + "  ".join(f"{nm}={v}" for nm, v in zip(names, out) if v))
    st = (C.c_ulonglong * 8)()
    if hasattr(L, "mh_debug_ms_stat"):
        L.mh_debug_ms_stat(st, 1)
        print(f"   (two calls) walk rows {st[0]} with outward {st[1]} outward members {st[2]} hit rows {st[3]} relax rounds {st[4]}; cycles in plain rows {st[5]}, outward rows {st[6]}, hit rows {st[7]}")
for n in (150, 300, 600):
    prof(rng.normal([320, 240], 6, size=(n, 2)).astype(np.float32), "tight")
    prof(rng.normal([320, 240], 40, size=(n, 2)).astype(np.float32), "blob40")
    prof(rng.uniform([0, 0], [640, 480], size=(n, 2)).astype(np.float32), "uniform")
gold = np.load(os.path.join(_R, "tests", "golden", "sift_ref_frames.npz"))
xy = gold["xy0"]
for k in (150, 379, 500, len(xy)):
    sel = np.sort(np.random.default_rng(1).choice(len(xy), k, replace=False))
    prof(np.ascontiguousarray(xy[sel]), "real keypoints")
    import time, orclib
    t0 = time.perf_counter(); want, it = orclib.meanshift(np.ascontiguousarray(xy[sel])); dc = time.perf_counter() - t0
    print(f"   oracle cpu {dc*1e3:.3f} ms iters={it} clusters={len(want)}")
