"""Randomised check of the merged front-end batch (round 4): B frames with a depth map each through ONE launch per stage
(depth patches, DEPTHFILTER, the depth-adaptive ratio + DEPTHFILTER2 + DEPTHMAP_PROP inside group_kernel, linkage or mean
shift, POSE<depth kind>, FILTER) against the same frames one at a time on the same context -- accepted match lists,
counts, objects bit for bit, twice over the same arenas.  The single-frame path is the one depth_rules_stress.py holds
against the oracle.  usage: depth_batch_stress.py [scenes=40] [seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import orclib
from moped_amd import capi, synth
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
K = synth.K_DEFAULT
dev = torch.device("cuda:0")
bad = n_obj = 0
t0 = time.perf_counter()
for sc in range(scenes):
    n_models, ppm = int(rng.choice([2, 6, 12])), int(rng.choice([600, 1500]))
    B = int(rng.integers(2, 9))
    Q = int(rng.choice([800, 1600, 3000]))   # (>= 4 x 140 planted points: every frame has exactly Q keypoints)
    db = synth.make_db(n_models, ppm, seed=int(rng.integers(1 << 30)))
    frs, maps = [], []
    for f in range(B):
        fr = synth.make_frame(db, n_vis=int(rng.integers(0, min(n_models, 4) + 1)), seed=int(rng.integers(1 << 30)), Q=Q,
                              pts_per_obj=int(rng.choice([40, 140])), outlier_frac=float(rng.choice([0.0, 0.3])))
        img, fill = synth.depth_image(db, fr, seed=int(rng.integers(1 << 30)), fill_max=float(rng.choice([0.02, 0.3])))
        if rng.random() < 0.5:
            y, x = int(rng.integers(0, 400)), int(rng.integers(0, 500))
            img[y:y + int(rng.integers(20, 200)), x:x + int(rng.integers(20, 300)), 2] = 5.0
        if rng.random() < 0.5:
            y = int(rng.integers(0, 440))
            img[y:y + int(rng.integers(5, 60)), :, 2] = np.nan
        assert len(fr.desc) == Q
        frs.append(fr)
        maps.append((torch.from_numpy(img).to(dev), torch.from_numpy(fill).to(dev)))
    fd = float(rng.choice([-1, 0.005, 0.02, 0.05]))
    md = float(rng.choice([-1, 0.001, 0.004, 0.01]))
    rules = bool(rng.random() < 0.8)
    linkage = bool(rng.random() < 0.6)
    kind = int(rng.choice([capi.DEPTH_BACKPROJECTION, 2]))
    table = None
    if rules and rng.random() < 0.6:
        table = np.stack([orclib.adaptive_control_points(db.xyz[db.model_of == m].min(0), db.xyz[db.model_of == m].max(0),
                                                         K, int((db.model_of == m).sum())) for m in range(n_models)])
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    c.reserve(B * Q)
    if rules: c.frame_set_depth_rules(K, 64, fd, md, table)
    if linkage: c.frame_set_cluster_linkage(capi.default_linkage_params())
    prm = capi.default_frame_params()
    alone = []
    for f, fr in enumerate(frs):
        c.frame_set_depth_image(maps[f][0].data_ptr(), maps[f][1].data_ptr(), 640, 480, kind, 0.5, 0.1)
        qd, uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
        c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, K, synth.CAM_IDENTITY, prm, 100 + f)
        objs, counts = c.frame_fetch()
        alone.append((objs, counts, c.frame_fetch_matches()))
        n_obj += len(objs)
    uv = torch.cat([torch.from_numpy(fr.uv) for fr in frs]).to(dev)
    c.frame_set_depth_image_batch([m[0].data_ptr() for m in maps], [m[1].data_ptr() for m in maps], 640, 480, kind, 0.5, 0.1)
    why = None
    for rep in range(2):
        qd = torch.cat([torch.from_numpy(fr.desc) for fr in frs]).to(dev)
        c.frame_enqueue_batch(qd.data_ptr(), uv.data_ptr(), Q, B, K, synth.CAM_IDENTITY, prm, [100 + f for f in range(B)])
        for f in range(B):
            objs, counts = c.frame_fetch_slot(f)
            a, ac, (aq, am) = alone[f]
            gq, gm = c.frame_fetch_matches_slot(f)
            if not (np.array_equal(gq, aq) and np.array_equal(gm, am)):
                why = why or (f"rep {rep} frame {f}: match lists ({len(gq)} vs {len(aq)} alone, only in the batch "
                              f"{sorted(set(gq.tolist()) - set(aq.tolist()))[:6]}, only alone {sorted(set(aq.tolist()) - set(gq.tolist()))[:6]})")
            elif not np.array_equal(counts, ac): why = why or f"rep {rep} frame {f}: counts {counts} vs {ac}"
            elif not (len(objs) == len(a) and np.array_equal(objs["model"], a["model"]) and
                      np.array_equal(objs["pose"].view(np.uint32), a["pose"].view(np.uint32)) and
                      np.array_equal(objs["score"].view(np.uint32), a["score"].view(np.uint32))):
                why = why or f"rep {rep} frame {f}: objects"
    if why:
        bad += 1
        print(f"MISMATCH scene {sc} ({why}): {n_models} models x {ppm}, B={B}, Q={Q}, rules {rules} {fd}/{md} table "
              f"{table is not None}, linkage {linkage}, kind {kind}", flush=True)
    c.close()
    if sc % 10 == 9: print(f"{sc + 1} scenes, {bad} mismatches, {n_obj} objects, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{scenes} merged front-end batches of 2..8 frames against the frames alone ({n_obj} objects), {bad} mismatches")
sys.exit(1 if bad else 0)
