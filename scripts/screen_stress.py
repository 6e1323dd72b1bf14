"""Randomised comparison of the two-stage MATCH with the exact f32 kernels: (idx1, d1, d2) bit for bit, over DB and query
shapes the parity tests do not enumerate -- sizes around the launch policy's thresholds (query blocks of 256 / 512 /
1024, one and two workgroups per CU), normalised and unnormalised rows (the row blocks' -dd/2 then spread widely),
near-duplicate clusters, integer-valued descriptors with many exact ties, shards with index_base.
usage: screen_stress.py [cases] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moped_amd import capi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
c = capi.Context(0)
bad = 0
t0 = time.time()
for k in range(cases):
    N = int(rng.choice([4096, 5000, 12345, 40000, 100000, 250000]))
    Q = int(rng.choice([1, 37, 256, 641, 1500, 2048, 3000, 4097, 9000]))
    kind = k % 6
    if kind == 0:      # SIFT-like: non-negative, L2-normalised
        d = np.abs(rng.normal(0, 1, (N, 128))).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    elif kind == 1:    # unnormalised: norms over two orders of magnitude
        d = (np.abs(rng.normal(0, 1, (N, 128))) * rng.uniform(0.05, 5.0, (N, 1))).astype(np.float32)
    elif kind == 2:    # near-duplicate clusters: 64 centres + 1e-4 noise
        ctr = np.abs(rng.normal(0, 1, (64, 128)))
        d = (ctr[rng.integers(0, 64, N)] + rng.normal(0, 1e-4, (N, 128))).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    elif kind == 3:    # small integers: many exact ties
        d = rng.integers(0, 3, (N, 128)).astype(np.float32)
    elif kind == 4:    # signed, normalised
        d = rng.normal(0, 1, (N, 128)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    else:              # mostly normalised with a few huge and a few zero rows
        d = np.abs(rng.normal(0, 1, (N, 128))).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[rng.integers(0, N, 5)] *= 30.0
        d[rng.integers(0, N, 5)] = 0.0
    # queries: perturbed DB rows (true matches), fresh random ones, exact copies
    src = rng.integers(0, N, Q)
    q = d[src] + rng.normal(0, rng.choice([0.0, 1e-3, 0.05, 0.3]), (Q, 128)).astype(np.float32)
    fresh = rng.random(Q) < 0.4
    q[fresh] = np.abs(rng.normal(0, 1, (int(fresh.sum()), 128))).astype(np.float32) * (1.0 if kind != 1 else 2.0)
    q = np.ascontiguousarray(q, np.float32)
    base = int(rng.choice([0, 777, 1 << 20]))
    c.db_upload(np.ascontiguousarray(d), np.zeros(N, np.int32), np.zeros((N, 3), np.float32), 1, index_base=base)
    tq_n = torch.from_numpy(q).to(dev)
    # (any norm term serves: both paths use the same one; the frame path passes the canonical fmaf-chain value)
    qn = torch.from_numpy((q.astype(np.float32) ** 2).sum(1, dtype=np.float32)).to(dev)
    res = {}
    for mode in (0, 1):
        out = [torch.empty(Q, dtype=t, device=dev) for t in (torch.int32, torch.float32, torch.float32)]
        c.match_set_mode(mode)
        c.match_local_dev(tq_n.data_ptr(), qn.data_ptr(), Q, *[o.data_ptr() for o in out])
        c.synchronize()
        res[mode] = [o.cpu().numpy() for o in out]
    c.match_set_mode(-1)
    a, b = res[0], res[1]
    same = (np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
            and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)))
    st = c.match_stats(Q)
    if not same:
        bad += 1
        w = np.nonzero((a[0] != b[0]) | (a[1].view(np.uint32) != b[1].view(np.uint32)) | (a[2].view(np.uint32) != b[2].view(np.uint32)))[0]
        print(f"MISMATCH case {k}: kind {kind} N {N} Q {Q} base {base}: {len(w)} queries, first {w[:5]}: "
              f"exact {[(a[0][i], a[1][i], a[2][i]) for i in w[:3]]} two-stage {[(b[0][i], b[1][i], b[2][i]) for i in w[:3]]}", flush=True)
    elif k % 10 == 0:
        print(f"case {k}: kind {kind} N {N} Q {Q}: ok (two-stage {st['two_stage']}, {time.time() - t0:.0f} s)", flush=True)
    c.match_stats(reset=True)
print(f"{cases} cases, {bad} mismatches")
c.close()
sys.exit(1 if bad else 0)
