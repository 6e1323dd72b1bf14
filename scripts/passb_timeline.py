"""Where does an ISOLATED pass B launch spend its time?  One slot (nothing else on the chip), the trace build: per
launch of pass B (kernel id 5) the span from the first workgroup's start to the last one's end, the workgroups'
residence, per compute unit how many workgroups it ran and how long it sat empty between / before / after them.
usage: passb_timeline.py [models=20] [B=8]"""
import ctypes as C
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MH_LIB_PATH", os.path.join(ROOT, "moped_amd", "libmoped_hip_trace.so"))
import numpy as np
import torch
from moped_amd import capi, synth

models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Q = 3000
L = capi.load()
L.mh_trace_enable.argtypes = [C.c_int]
L.mh_trace_fetch.argtypes = [C.c_void_p, C.c_longlong]
L.mh_trace_fetch.restype = C.c_longlong
db = synth.make_db(models, 5000)
dev = torch.device("cuda:0")
frs = [synth.make_frame(db, n_vis=2, seed=s, Q=Q) for s in range(B)]
qn = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
Qr = Q * B
c = capi.Context(0)
c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
c.reserve(Qr)
qnorm = torch.empty(Qr, dtype=torch.float32, device=dev)
idx = torch.empty(Qr, dtype=torch.int32, device=dev)
d1 = torch.empty(Qr, dtype=torch.float32, device=dev)
d2 = torch.empty(Qr, dtype=torch.float32, device=dev)
c.normalize_dev(qn.data_ptr(), qnorm.data_ptr(), Qr)
for _ in range(5):
    c.match_local_dev(qn.data_ptr(), qnorm.data_ptr(), Qr, idx.data_ptr(), d1.data_ptr(), d2.data_ptr())
c.synchronize()
L.mh_trace_enable(1)
reps = 10
for _ in range(reps):
    c.match_local_dev(qn.data_ptr(), qnorm.data_ptr(), Qr, idx.data_ptr(), d1.data_ptr(), d2.data_ptr())
c.synchronize()
buf = np.zeros((1 << 20, 4), np.uint64)
n = L.mh_trace_fetch(buf.ctypes.data, buf.shape[0])
L.mh_trace_enable(0)
r = buf[:n]
kid = (r[:, 0] >> np.uint64(32)).astype(np.int64)
hw = (r[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = (r[:, 1] & np.uint64(0xF)).astype(np.int64)
ta, tb = r[:, 2].astype(np.int64), r[:, 3].astype(np.int64)     # s_memrealtime: 100 MHz
cu = (xcc << 8) | ((hw >> 8) & 0xFF)
for k, name in ((3, "pass A"), (5, "pass B")):
    m = kid == k
    a, b, u = ta[m], tb[m], cu[m]
    order = np.argsort(a)
    a, b, u = a[order], b[order], u[order]
    per = len(a) // reps
    print(f"{name}: {len(a)} workgroup records = {reps} launches x {per}; {models} models, {Qr} queries")
    rows = []
    for i in range(reps):
        s = slice(i * per, (i + 1) * per)
        aa, bb, uu = a[s], b[s], u[s]
        t0, t1 = aa.min(), bb.max()
        span = (t1 - t0) * 1e-2
        dur = (bb - aa) * 1e-2
        cus = np.unique(uu)
        busy = 0.0
        lead, tail, gap = [], [], []
        for cc in cus:
            mm = uu == cc
            x, y = aa[mm], bb[mm]
            busy += (y - x).sum() * 1e-2
            lead.append((x.min() - t0) * 1e-2)
            tail.append((t1 - y.max()) * 1e-2)
            if mm.sum() > 1:
                o = np.argsort(x)
                gap.append(float(((x[o][1:] - y[o][:-1]).clip(min=0)).sum()) * 1e-2)
        counts = np.bincount(np.unique(uu, return_inverse=True)[1])
        rows.append((span, dur.mean(), dur.min(), dur.max(), len(cus), counts.min(), counts.max(), np.mean(lead), np.max(lead),
                     np.mean(tail), np.max(tail), np.mean(gap) if gap else 0.0, 100 * busy / (len(cus) * span)))
    rows = np.array(rows)
    med = np.median(rows, axis=0)
    print(f"  span first start -> last end: {med[0]:.1f} us (median of {reps}); workgroup residence mean {med[1]:.1f} us "
          f"(min {med[2]:.1f}, max {med[3]:.1f})")
    print(f"  compute units used {med[4]:.0f}, workgroups per unit {med[5]:.0f}..{med[6]:.0f}; a unit's first workgroup starts "
          f"{med[7]:.1f} us after the launch's first (worst {med[8]:.1f}); its last ends {med[9]:.1f} us before the launch's last "
          f"(worst {med[10]:.1f}); empty between its workgroups {med[11]:.1f} us; units busy {med[12]:.1f}% of the span")
# what do slow workgroups have in common?  (blockIdx.x rides in the trace record: XCD = L & 7; the kernel's unit map gives
# split and query block)
m5 = kid == 5
if m5.any():
    bx = ((r[:, 1] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)[m5]
    d5 = (tb[m5] - ta[m5]) * 1e-2
    a5 = ta[m5]
    U = 504 if True else 0
    U = int(bx.max()) + 1
    nqb = -(-Qr // 1024)
    x, j = bx & 7, bx >> 3
    unit = x * (U >> 3) + np.minimum(x, U & 7) + j
    split, qb = unit // nqb, unit % nqb
    # round: is it the first or the second workgroup of its unit in its launch?
    order = np.argsort(a5)
    launch = np.empty(len(a5), np.int64); launch[order] = np.arange(len(a5)) // U
    first = np.ones(len(a5), bool)
    cu5 = cu[m5]
    seen = {}
    for i in order:
        k = (int(launch[i]), int(cu5[i]))
        first[i] = k not in seen
        seen[k] = 1
    def table(name, key):
        print(f"  mean residence by {name}: " + " ".join(f"{int(k)}:{d5[key == k].mean():.0f}" for k in np.unique(key)))
    print(f"pass B workgroups ({U} per launch, {nqb} query blocks):")
    table("XCD", x)
    table("round on its unit (1 = first)", np.where(first, 1, 2))
    table("query block", qb)
    table("split", split)
# the tile loop inside pass B's workgroups (kernel id 11, trace build): what is left is prologue (the queries' operands
# into registers, thresholds into LDS, the first tile's DMA issued) and epilogue (pending block, parked records)
m5, m11 = kid == 5, kid == 11
if m11.any():
    key5 = {(int(u), int(y)): (int(x), int(y)) for u, x, y in zip(cu[m5], ta[m5], tb[m5])}
    pro, epi, loop = [], [], []
    for u, x, y in zip(cu[m11], ta[m11], tb[m11]):
        # the enclosing workgroup: same unit, starts before, ends after
        cand = [(a0, b0) for (uu, b0), (a0, _) in key5.items() if uu == int(u) and a0 <= x and b0 >= y]
        if cand:
            a0, b0 = min(cand, key=lambda ab: ab[1] - ab[0])
            pro.append((x - a0) * 1e-2); epi.append((b0 - y) * 1e-2); loop.append((y - x) * 1e-2)
    pro, epi, loop = map(np.array, (pro, epi, loop))
    print(f"pass B workgroup: prologue {pro.mean():.1f} us (p90 {np.quantile(pro, .9):.1f}), tile loop {loop.mean():.1f} us "
          f"(min {loop.min():.1f}, max {loop.max():.1f}), epilogue {epi.mean():.1f} us (p90 {np.quantile(epi, .9):.1f})")
c.close()
