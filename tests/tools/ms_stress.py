"""One-off stress: the HIP mean shift against the oracle on many random point sets (sizes 1..900, tight blobs,
chains, grids with duplicate points, uniform clutter; 2-D and 3-D).  usage: ms_stress.py [cases] [seed]"""
import sys, os, numpy as np
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'oracle'))
import orclib
from moped_amd import capi
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = capi.Context(0)
bad = 0
for k in range(cases):
    sizes = [int(x) for x in os.environ["MS_SIZES"].split(",")] if "MS_SIZES" in os.environ else \
        [1, 2, 7, 20, 63, 64, 65, 128, 129, 200, 383, 385, 511, 513, 700, 900]
    n = int(rng.choice(sizes))
    dim = 3 if k % 7 == 0 else 2
    kind = k % 5
    if kind == 0:
        pts = rng.normal(300, rng.choice([3, 15, 60]), size=(n, dim))
    elif kind == 1:   # a chain of points a bit less than Merge apart: far from transitive
        t = np.arange(n)[:, None] * rng.uniform(5, 25)
        pts = np.concatenate([t, rng.normal(0, 3, size=(n, dim - 1))], 1)
        pts = pts[rng.permutation(n)]
    elif kind == 2:   # grid with duplicates
        pts = rng.integers(0, 12, size=(n, dim)) * rng.choice([4.0, 19.0, 21.0])
    elif kind == 3:
        pts = rng.uniform(0, [640, 480, 100][:dim], size=(n, dim))
    else:             # several blobs + clutter
        c = rng.uniform(0, 600, size=(4, dim))
        pts = np.concatenate([c[rng.integers(0, 4, n - n // 4)] + rng.normal(0, 8, size=(n - n // 4, dim)),
                              rng.uniform(0, 640, size=(n // 4, dim))])
        pts = pts[rng.permutation(len(pts))]
    pts = np.ascontiguousarray(pts, np.float32)
    radius, merge = float(rng.choice([100.0, 200.0, 30.0])), float(rng.choice([20.0, 5.0, 40.0]))
    min_pts, iters = int(rng.choice([1, 4, 7])), int(rng.choice([100, 3]))
    want, _ = orclib.meanshift(pts, radius, merge, min_pts, iters)
    got, _ = ctx.meanshift(pts, radius, merge, min_pts, iters)
    ok = len(want) == len(got) and all(np.array_equal(a, b) for a, b in zip(want, got))
    if not ok:
        bad += 1
        print(f"MISMATCH case {k}: n={n} dim={dim} kind={kind} radius={radius} merge={merge} min_pts={min_pts} iters={iters}: "
              f"{len(want)} vs {len(got)} clusters")
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
