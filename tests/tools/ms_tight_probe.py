"""Mean shift on dense / uniform point sets of growing size: wall time per call and (MS_PROF builds) phase cycles."""
import sys, os, ctypes as C, numpy as np, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], 'oracle'))
from moped_amd import capi
if os.environ.get('WITH_ORACLE'):
    import orclib
if os.environ.get('WITH_SYNTH'):
    from moped_amd import synth
ctx = capi.Context(0)
L = capi.load()
rng = np.random.default_rng(0)
names = ["loop top", "(1) means", "(2a) bits", "(2b) walk", "(3) fold", "compact", "emit", "-"]
for n in (16, 64, 150, 300):
    pts = rng.normal([320, 240], 6, size=(n, 2)).astype(np.float32)
    uni = rng.uniform([0, 0], [640, 480], size=(n, 2)).astype(np.float32)
    for lab, p in (("tight", pts), ("uniform", uni)):
        out = (C.c_ulonglong * 8)()
        prof = hasattr(L, "mh_debug_ms_prof")
        ctx.meanshift(p)
        if prof: L.mh_debug_ms_prof(out, 1)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); ctx.meanshift(p); ts.append(time.perf_counter() - t0)
        st = (C.c_ulonglong * 8)()
        if prof:
            L.mh_debug_ms_prof(out, 1); L.mh_debug_ms_stat(st, 1)
        print(lab, n, " ".join(f"{x*1e3:.3f}" for x in ts), "ms", "total", sum(out), "  ".join(f"{nm}={v}" for nm, v in zip(names, out) if v), "| stats", list(st))
