"""Numbers before building anything: what would an int8 first stage of the two-stage MATCH leave for the f16 stage?
Queries and DB rows quantised to int8 (one scale for all: 127 / the largest component), exact integer dot products (what
v_mfma_i32_16x16x64_i8 computes), a PROVED margin from the quantisation errors' norms (Cauchy-Schwarz, as the f16 screen's
margin is proved from rounding bounds), thresholds from a 1-in-8 tile sample like pass A's.  Printed: survivors per query
(rows the f16 stage would have to look at) against the f16 screen's own survivors.  CPU only (numpy).
usage: int8_screen_estimate.py [models=20] [queries=600]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import orclib
from moped_amd import synth
models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 600
db = synth.make_db(models, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0, Q=3000)
D = orclib.normalize(db.desc).astype(np.float64)
rng = np.random.default_rng(0)
pick = np.sort(rng.choice(3000, nq, replace=False))
Qv = orclib.normalize(fr.desc)[pick].astype(np.float64)
# the screen value the device maximises: q.d - dd/2 (d^2 = qq + dd - 2 q.d); rows are unit vectors: dd/2 is constant
S = Qv @ D.T
N = D.shape[0]
tiles = np.arange(N) // 128
sample = (tiles % 8) == 4
def survivors(score, margin):
    """rows above (second best of the sampled rows) - margin, per query"""
    samp = np.sort(score[:, sample], axis=1)[:, -2]
    return (score >= (samp - margin)[:, None]).sum(1)
# f16: values rounded to f16, products exact in f32: |err| <= 2^-11 (|q|.|d| + ...) -- the device's margin is ~1e-3
q16, d16 = Qv.astype(np.float16).astype(np.float64), D.astype(np.float16).astype(np.float64)
e16 = np.linalg.norm(Qv - q16, axis=1)[:, None] * np.linalg.norm(d16, axis=1)[None, :] + np.linalg.norm(Qv, axis=1)[:, None] * np.linalg.norm(D - d16, axis=1)[None, :]
s16 = survivors(q16 @ d16.T, 2 * e16.max(1))
# int8: one scale (components are in [0, ~0.45] after normalisation)
for bits, name in ((8, "int8 (127 levels)"),):
    scale = 127.0 / max(Qv.max(), D.max())
    qi, di = np.round(Qv * scale), np.round(D * scale)
    si = (qi @ di.T) / scale ** 2
    eq, ed = np.linalg.norm(Qv - qi / scale, axis=1), np.linalg.norm(D - di / scale, axis=1)
    # |q.d - qi.di/s^2| <= |q - qi/s| |di/s| + |q| |d - di/s|  (exact per vector: the norms are computed at quantisation time)
    bound = eq[:, None] * np.linalg.norm(di / scale, axis=1)[None, :] + np.linalg.norm(Qv, axis=1)[:, None] * ed[None, :]
    worst = np.abs(si - S).max()
    s8 = survivors(si, 2 * bound.max(1))
    s8_rowwise = (si >= (np.sort(si[:, sample], axis=1)[:, -2] - bound.max(1))[:, None] - bound).sum(1)   # per-row error norms in the test
    print(f"{name}: scale {scale:.1f}; largest |int8 score - exact| {worst:.5f}, proved bound (max over rows) median {np.median(bound.max(1)):.5f}")
    print(f"  survivors per query with the query's worst-case margin: median {np.median(s8):.0f}, mean {s8.mean():.1f}, p90 {np.percentile(s8, 90):.0f}, max {s8.max()}")
    print(f"  survivors per query with per-row margins:               median {np.median(s8_rowwise):.0f}, mean {s8_rowwise.mean():.1f}, p90 {np.percentile(s8_rowwise, 90):.0f}, max {s8_rowwise.max()}")
print(f"f16 screen, same sample: survivors per query median {np.median(s16):.0f}, mean {s16.mean():.1f}, p90 {np.percentile(s16, 90):.0f}, max {s16.max()}")
exact = (S >= np.sort(S[:, sample], axis=1)[:, -2][:, None]).sum(1)
print(f"exact values, same sample (no margin):  median {np.median(exact):.0f}, mean {exact.mean():.1f}  (the order-statistics floor of a 1-in-8 sample)")
print(f"{N} rows, {nq} queries; second-nearest screen value: median {np.median(np.sort(S, axis=1)[:, -2]):.4f}, its gap to the 20th: median {np.median(np.sort(S, axis=1)[:, -2] - np.sort(S, axis=1)[:, -20]):.4f}")
