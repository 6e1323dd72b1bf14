"""One synchronous frame at a time on the trace build (libmoped_hip_trace.so): the workgroups of CLUSTER, POSE and POSE2
of each frame -- when each started and ended relative to the first of its launch (10 ns ticks) -- to see what a launch's
duration is made of: the longest task, the workgroup that closes the frame (fused FILTER), dispatch.
usage: single_frame_trace.py [models=20] [n_vis=2] [frames=20]"""
import ctypes as C, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MH_LIB_PATH", os.path.join(ROOT, "moped_amd", "libmoped_hip_trace.so"))
import numpy as np, torch
from moped_amd import capi, synth
from moped_amd.pipeline import FramePipeline, ShardedDB
models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_vis = int(sys.argv[2]) if len(sys.argv) > 2 else 2
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
Q = 3000
L = capi.load()
L.mh_trace_enable.argtypes = [C.c_int]
L.mh_trace_fetch.argtypes = [C.c_void_p, C.c_longlong]
L.mh_trace_fetch.restype = C.c_longlong
db = synth.make_db(models, 5000)
dev = torch.device("cuda:0")
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=Q)
frs = [synth.make_frame(db, n_vis=n_vis, seed=s, Q=Q) for s in range(frames)]
dev_fr = [(torch.from_numpy(f.desc).to(dev), torch.from_numpy(f.uv).to(dev)) for f in frs]
for i in range(5):
    pipe.enqueue(0, dev_fr[i % frames][0].clone(), dev_fr[i % frames][1], seed=i + 1); torch.cuda.synchronize(); pipe.fetch(0)
NAMES = {7: "group", 8: "CLUSTER", 9: "POSE"}
rows = {k: [] for k in ("CLUSTER", "POSE", "POSE2")}
for i in range(frames):
    qd = dev_fr[i][0].clone(); torch.cuda.synchronize()
    L.mh_trace_enable(1)
    pipe.enqueue(0, qd, dev_fr[i][1], seed=100 + i)
    pipe.fetch(0)
    buf = np.zeros((1 << 16, 4), np.uint64)
    n = L.mh_trace_fetch(buf.ctypes.data, buf.shape[0])
    L.mh_trace_enable(0)
    r = buf[:n]
    kid = (r[:, 0] >> np.uint64(32)).astype(np.int64)
    ta, tb = r[:, 2].astype(np.int64), r[:, 3].astype(np.int64)
    for name, k in (("CLUSTER", 8), ("POSE", 9)):
        m = kid == k
        if not m.any(): continue
        a, b = ta[m], tb[m]
        order = np.argsort(a); a, b = a[order], b[order]
        if name == "POSE":   # two launches (POSE, POSE2), or four when the stage runs as hypotheses + refine: split at gaps
            cuts = [0] + [j for j in range(1, len(a)) if a[j] > b[:j].max()] + [len(a)]
            groups = [(a[cuts[j]:cuts[j + 1]], b[cuts[j]:cuts[j + 1]]) for j in range(len(cuts) - 1)]
            half = len(groups) // 2
            for gi, (ga, gb) in enumerate(groups):
                rows["POSE" if gi < max(half, 1) else "POSE2"].append((ga - ga.min(), gb - ga.min()))
        else:
            rows[name].append((a - a.min(), b - a.min()))
for name, lst in rows.items():
    if not lst: continue
    span = np.array([b.max() for a, b in lst]) / 100.0
    nwg = np.array([len(a) for a, b in lst])
    longest = np.array([(b - a).max() for a, b in lst]) / 100.0
    med = np.array([np.median(b - a) for a, b in lst]) / 100.0
    last_start = np.array([a.max() for a, b in lst]) / 100.0
    closer = np.array([(b - a)[np.argmax(b)] for a, b in lst]) / 100.0   # residence of the workgroup that ended last
    second_end = np.array([np.sort(b)[-2] if len(b) > 1 else b.max() for a, b in lst]) / 100.0
    print(f"{name:8s} launches {len(lst):3d}: workgroups {nwg.mean():5.1f}; first start -> last end {np.median(span):6.1f} us; "
          f"median workgroup {np.median(med):6.1f}, longest {np.median(longest):6.1f}, the one that ends last {np.median(closer):6.1f}; "
          f"last workgroup starts at {np.median(last_start):5.1f}; second-to-last end at {np.median(second_end):6.1f}")
