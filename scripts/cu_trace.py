"""Who holds the compute units?  Runs the bench's pipeline shape (config 1: 16 slots x 8 frames per MATCH launch
sequence) on the trace build (make -C moped_amd/csrc EXTRA=-DMH_TRACE BUILD=build_trace OUT=../libmoped_hip_trace.so),
collects {kernel, compute unit, t0, t1} of every workgroup and rebuilds every compute unit's timeline:
  share of CU time with a MATCH-pass workgroup resident / with only small-kernel workgroups / with nothing,
  per kernel: workgroups, mean residence, CU-seconds, how many share a CU with another small workgroup.
usage: cu_trace.py [models=20] [B=8] [depth=16] [batches=96] [mode=full|match|rest]"""
import ctypes as C
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("MH_LIB_PATH", os.path.join(ROOT, "moped_amd", "libmoped_hip_trace.so"))
import numpy as np
import torch
from moped_amd import capi, synth
from moped_amd.pipeline import FramePipeline, ShardedDB

models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 16
batches = int(sys.argv[4]) if len(sys.argv) > 4 else 96
mode = sys.argv[5] if len(sys.argv) > 5 else "full"
Q = 3000
NAMES = {1: "normalize", 2: "prepare", 3: "pass A", 4: "tau", 5: "pass B", 6: "pass C", 7: "group", 8: "CLUSTER", 9: "POSE", 10: "other"}
BIG = (3, 5)

L = capi.load()
L.mh_trace_enable.argtypes = [C.c_int]
L.mh_trace_fetch.argtypes = [C.c_void_p, C.c_longlong]
L.mh_trace_fetch.restype = C.c_longlong
db = synth.make_db(models, 5000)
dev = torch.device("cuda:0")
frs = [synth.make_frame(db, n_vis=2, seed=s, Q=Q) for s in range(B)]
qd0 = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
prm = capi.default_frame_params()
if mode == "match":      # MATCH + a rest chain with next to nothing in it
    prm.pose1.lm_iters_l2 = prm.pose1.lm_iters_l4 = prm.pose2.lm_iters_l2 = prm.pose2.lm_iters_l4 = 0
    prm.pose1.n_hypotheses = prm.pose2.n_hypotheses = 64
    prm.pose1.max_objects_per_cluster = prm.pose2.max_objects_per_cluster = 1
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=depth, max_queries=Q * B, params=prm)
work = [torch.empty_like(qd0) for _ in range(depth)]


def go(n):
    for g in range(n):
        slot = g % depth
        with torch.cuda.stream(pipe.streams[slot]):
            work[slot].copy_(qd0, non_blocking=True)
        pipe.enqueue_batch(slot, work[slot], uv, B, list(range(g * B + 1, g * B + B + 1)))


go(4 * depth)
pipe.synchronize()
t0 = time.perf_counter()
go(8 * depth)
pipe.synchronize()
fps_untraced = 8 * depth * B / (time.perf_counter() - t0)
L.mh_trace_enable(1)
t0 = time.perf_counter()
go(batches)
pipe.synchronize()
dt = time.perf_counter() - t0
buf = np.zeros((1 << 21, 4), np.uint64)
n = L.mh_trace_fetch(buf.ctypes.data, buf.shape[0])
L.mh_trace_enable(0)
print(f"{models} models, {B} frames per batch, {depth} slots, {batches} batches ({mode}): {batches * B / dt:.0f} frames/s traced, "
      f"{fps_untraced:.0f} untraced; {n} workgroup records")
r = buf[:min(n, buf.shape[0])]
r = r[(r[:, 0] >> np.uint64(32)) <= np.uint64(10)]   # (ids above 10 are phases inside a pass A / B workgroup: scripts/passb_timeline.py)
if os.environ.get("CU_TRACE_SAVE"):
    np.save(os.environ["CU_TRACE_SAVE"], r)
kid = (r[:, 0] >> np.uint64(32)).astype(np.int64)
hw = (r[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = (r[:, 1] & np.uint64(0xF)).astype(np.int64)
ta, tb = r[:, 2].astype(np.int64), r[:, 3].astype(np.int64)
cu = (xcc << 8) | ((hw >> 8) & 0xFF)          # XCC | SE | SH | CU
queue = ((hw >> 24) & 7) | (((hw >> 6) & 3) << 3) | (((hw >> 30) & 3) << 5)   # ME | PIPE | QUEUE
print(f"compute units seen: {len(np.unique(cu))}; hardware queues seen (ME, pipe, queue): {len(np.unique(queue))}")
# steady window: the middle of the traced span
lo, hi = np.quantile(ta, 0.2), np.quantile(tb, 0.8)
span = (hi - lo) * 1e-8
print(f"window {span * 1e3:.2f} ms")
tot_big = tot_small_only = tot_any = 0.0
co_small = {}
for c in np.unique(cu):
    m = cu == c
    ev = []
    for k, a, b in zip(kid[m], ta[m], tb[m]):
        a, b = max(a, lo), min(b, hi)
        if b > a:
            big = 1 if k in BIG else 0
            ev.append((a, 1, big, k))
            ev.append((b, -1, big, k))
    ev.sort()
    nb = ns = 0
    last = lo
    for t, d, big, k in ev:
        if nb > 0:
            tot_big += t - last
        elif ns > 0:
            tot_small_only += t - last
        if nb + ns > 0:
            tot_any += t - last
        last = t
        if big:
            nb += d
        else:
            ns += d
n_cu = len(np.unique(cu))
denom = n_cu * (hi - lo)
print(f"CU time: {100 * tot_big / denom:.1f}% with a pass A/B workgroup resident, {100 * tot_small_only / denom:.1f}% with only "
      f"small-kernel workgroups, {100 * (1 - tot_any / denom):.1f}% empty")
inwin = (tb > lo) & (ta < hi)
print(f"{'kernel':10s} {'workgroups':>10s} {'mean us':>9s} {'p90 us':>9s} {'CU-ms':>9s} {'share of CU time':>17s}")
for k in sorted(NAMES):
    m = inwin & (kid == k)
    if not m.any():
        continue
    d = (np.minimum(tb[m], hi) - np.maximum(ta[m], lo)) * 1e-2
    print(f"{NAMES[k]:10s} {m.sum():10d} {d.mean():9.1f} {np.quantile(d, 0.9):9.1f} {d.sum() * 1e-3:9.2f} {100 * d.sum() * 1e-6 / (n_cu * span):16.2f}%")
# how alone are the small workgroups: for every POSE / CLUSTER workgroup, the number of other small workgroups on its CU at its start
for k in (8, 9, 7):
    m = np.nonzero(inwin & (kid == k))[0]
    if not len(m):
        continue
    others = []
    for i in m[:4000]:
        same = (cu == cu[i]) & (ta <= ta[i]) & (tb > ta[i]) & ~np.isin(kid, BIG)
        others.append(int(same.sum()) - 1)
    others = np.array(others)
    print(f"{NAMES[k]}: small workgroups already on its CU when it starts: mean {others.mean():.2f}, alone {100 * (others == 0).mean():.0f}%")
pipe.close()
