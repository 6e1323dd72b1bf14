"""Randomised check of moped3d's front end inside the device-resident frame: DEPTHFILTER, the depth-adaptive ratio and
DEPTHFILTER2 (the accepted match lists must be the oracle's, index-exact) and, on every other scene, CLUSTER = linkage
on those lists (the cluster count must be the oracle's) -- over random databases, frames, depth maps (NaN bands, far
regions, filled pixels), densities and adaptive tables.  usage: depth_rules_stress.py [scenes=40] [seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import orclib
from moped_amd import capi, synth
from moped_amd.pipeline import FramePipeline, ShardedDB
import test_gpu_depth_rules as T
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
K = synth.K_DEFAULT
dev = torch.device("cuda:0")
bad = 0
t0 = time.perf_counter()
for sc in range(scenes):
    n_models, ppm = int(rng.choice([2, 6, 12])), int(rng.choice([600, 1500]))
    n_vis = int(rng.integers(0, min(n_models, 4) + 1))
    db = synth.make_db(n_models, ppm, seed=int(rng.integers(1 << 30)))
    fr = synth.make_frame(db, n_vis=n_vis, seed=int(rng.integers(1 << 30)), Q=int(rng.choice([500, 1600, 3000])),
                          pts_per_obj=int(rng.choice([40, 140])), outlier_frac=float(rng.choice([0.0, 0.3])))
    Q = len(fr.desc)
    img, fill = synth.depth_image(db, fr, seed=int(rng.integers(1 << 30)), fill_max=float(rng.choice([0.02, 0.3])))
    if rng.random() < 0.6:                                     # a region beyond MaximumDepth
        y, x = int(rng.integers(0, 400)), int(rng.integers(0, 500))
        img[y:y + int(rng.integers(20, 200)), x:x + int(rng.integers(20, 300)), 2] = 5.0
    if rng.random() < 0.6:                                     # a band of NaN depths
        y = int(rng.integers(0, 440))
        img[y:y + int(rng.integers(5, 60)), :, 2] = np.nan
    fd = float(rng.choice([-1, 0.005, 0.02, 0.05, 0.2]))
    md = float(rng.choice([-1, 0.001, 0.004, 0.01, 0.05]))
    adaptive = bool(rng.random() < 0.6)
    linkage = sc % 2 == 1
    table = None
    if adaptive:
        table = np.stack([orclib.adaptive_control_points(db.xyz[db.model_of == m].min(0), db.xyz[db.model_of == m].max(0),
                                                         K, int((db.model_of == m).sum())) for m in range(n_models)])
    idx, d1, d2 = orclib.match_2nn(orclib.normalize(db.desc), orclib.normalize(fr.desc))
    s = dict(db=db, fr=fr, img=img, fill=fill, idx=idx, d1=d1, d2=d2)
    want_q, want_m = T._oracle_lists(s, fd, md, table)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=Q)
    c = pipe.ctxs[0]
    d_img, d_fill = torch.from_numpy(img).to(dev), torch.from_numpy(fill).to(dev)
    c.frame_set_depth_image(d_img.data_ptr(), d_fill.data_ptr(), 640, 480, capi.DEPTH_BACKPROJECTION, 0.5, 0.1)
    c.frame_set_depth_rules(K, 64, fd, md, table)
    if linkage: c.frame_set_cluster_linkage(capi.default_linkage_params())
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=3)
    objs, counts = pipe.fetch(0)
    got_q, got_m = c.frame_fetch_matches()
    ok = np.array_equal(got_q, want_q) and np.array_equal(got_m, want_m) and counts[0] == len(want_q)
    what = "match lists"
    if ok and linkage:
        want_c = 0
        for m in range(n_models):
            q = want_q[want_m == m]
            uv = fr.uv[q]
            world, _ = orclib.depthmap_prop(img, fill, uv, 0.1)
            want_c += len(orclib.cluster_linkage(uv, db.xyz[idx[q]], world, img, fill))
        ok, what = counts[1] == want_c, f"clusters {counts[1]} vs {want_c}"
    if not ok:
        bad += 1
        print(f"MISMATCH scene {sc} ({what}): {n_models} models x {ppm}, Q={Q}, n_vis={n_vis}, densities {fd}/{md}, "
              f"adaptive {adaptive}, linkage {linkage}: {len(got_q)} matches vs {len(want_q)}", flush=True)
    c.frame_set_cluster_linkage(None)
    c.frame_set_depth_rules(off=True)
    c.frame_set_depth_image(0, 0, 0, 0, 0)
    pipe.close()
    if sc % 10 == 9: print(f"{sc + 1} scenes, {bad} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{scenes} moped3d front-end scenes (depth rules; linkage on every other one), {bad} mismatches")
sys.exit(1 if bad else 0)
