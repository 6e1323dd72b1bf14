"""One synchronous frame at a time on the trace build (make EXTRA=-DMH_TRACE BUILD=build_trace OUT=../libmoped_hip_trace.so):
the whole chain's kernels from the inside -- per traced kernel the first workgroup's start and the last one's end on the
device's 100 MHz clock, relative to the frame's first record -- so that what lies BETWEEN the kernels (dispatch, the
end-of-kernel release, launches that are not traced) shows.  usage: single_frame_chain.py [models=20] [n_vis=2] [frames=20]"""
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
NAMES = {1: "normalize", 2: "prepare", 3: "pass A", 4: "tau", 5: "pass B", 6: "pass C", 7: "group", 8: "CLUSTER", 9: "POSE", 10: "other"}
per = []
for i in range(frames):
    qd = dev_fr[i][0].clone(); torch.cuda.synchronize()
    L.mh_trace_enable(1)
    pipe.enqueue(0, qd, dev_fr[i][1], seed=100 + i)
    pipe.fetch(0)
    buf = np.zeros((1 << 18, 4), np.uint64)
    n = L.mh_trace_fetch(buf.ctypes.data, buf.shape[0])
    L.mh_trace_enable(0)
    r = buf[:n]
    kid = (r[:, 0] >> np.uint64(32)).astype(np.int64)
    ta, tb = r[:, 2].astype(np.int64), r[:, 3].astype(np.int64)
    t0 = ta.min()
    segs = []
    for k in sorted(set(kid.tolist())):
        if k > 10: continue
        m = kid == k
        a, b = ta[m], tb[m]
        order = np.argsort(a); a, b = a[order], b[order]
        cuts = [0] + [j for j in range(1, len(a)) if a[j] > b[:j].max() + 100] + [len(a)]   # launches of the same kernel: split at gaps > 1 us
        for j in range(len(cuts) - 1):
            segs.append((a[cuts[j]] - t0, b[cuts[j]:cuts[j + 1]].max() - t0, NAMES.get(k, str(k)), cuts[j + 1] - cuts[j]))
    segs.sort()
    per.append(segs)
n_seg = min(len(s) for s in per)
print(f"{'kernel (from the inside)':26s} {'workgroups':>10s} {'first start':>12s} {'last end':>10s} {'inside':>8s} {'gap before':>11s}   (us, medians over {frames} frames)")
prev_end = None
for j in range(n_seg):
    st = np.median([s[j][0] for s in per]) / 100.0
    en = np.median([s[j][1] for s in per]) / 100.0
    nm = per[0][j][2]
    wg = np.median([s[j][3] for s in per])
    gap = "" if prev_end is None else f"{st - prev_end:11.1f}"
    print(f"{nm:26s} {wg:10.0f} {st:12.1f} {en:10.1f} {en - st:8.1f} {gap:>11s}")
    prev_end = en
pipe.close()
