"""Randomised parity of frames with several images (mh_frame_set_images) against the oracle's restatement of the
reference's per-image CLUSTER, per-correspondence camera in POSE and (coord2D, image) ownership in FILTER:
random camera rigs (2-4 cameras), databases, visible objects, points per object.  Exact: accepted matches, clusters;
model sets equal, poses within 1 px of the oracle's (differences are reseeded on the oracle's side first).
usage: frame_stress_images.py [scenes=40] [seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import orclib
from moped_amd import capi, synth
from moped_amd.pipeline import FramePipeline, ShardedDB
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad = spread = scores = 0
t0 = time.perf_counter()
for sc in range(scenes):
    n_models, ppm = int(rng.choice([3, 6, 10])), int(rng.choice([800, 1500]))
    db = synth.make_db(n_models, ppm, seed=int(rng.integers(1 << 30)))
    n_cam = int(rng.integers(2, 5))
    cams = [synth.camera_pose(0.0)] + [synth.camera_pose(float(rng.uniform(-0.15, 0.15)), (float(rng.uniform(-0.12, 0.12)), float(rng.uniform(-0.03, 0.03)), 0.0))
                                       for _ in range(n_cam - 1)]
    n_vis = int(rng.integers(0, min(n_models, 4) + 1))
    pts = int(rng.choice([30, 60, 130]))
    fr = synth.make_frame_images(db, cams, n_vis=n_vis, seed=int(rng.integers(1 << 30)), q_per_image=int(rng.choice([400, 900, 1500])),
                                 pts_per_obj=pts, outlier_frac=float(rng.choice([0.0, 0.2, 0.4])))
    seed = int(rng.integers(1, 1 << 20))
    Q = len(fr.uv)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=Q)
    c = pipe.ctxs[0]
    q_img = torch.from_numpy(fr.image).to(dev)
    c.frame_set_images(q_img.data_ptr(), fr.Ks, fr.cams)
    q_desc, q_uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
    pipe.enqueue(0, q_desc, q_uv, seed=seed)
    objs, counts = pipe.fetch(0)
    idx, d1, d2 = orclib.match_2nn(orclib.normalize(db.desc), orclib.normalize(fr.desc))
    def oracle(s_):
        return orclib.frame_rest_images(fr.uv, fr.image, idx, d1, d2, db.model_of, db.xyz, db.n_models, fr.Ks, fr.cams, seed=s_)
    om, op, osc, oc = oracle(seed)
    why = []
    if counts[0] != oc[0]: why.append(f"matches {counts[0]} vs {oc[0]}")
    if counts[1] != oc[1]: why.append(f"clusters {counts[1]} vs {oc[1]}")
    def differ(om, op, osc):
        if sorted(objs["model"].tolist()) != sorted(om.tolist()):
            return [f"models {sorted(objs['model'].tolist())} vs {sorted(om.tolist())}"]
        w = []
        for m, p, s_ in zip(om, op, osc):
            g = objs[objs["model"] == m][0]
            rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
            rows = rows[db.model_of[fr.src_point[rows]] == m]
            if len(rows) < 8: continue
            xyz, uv, img = db.xyz[fr.src_point[rows]], fr.uv[rows], fr.image[rows]
            e = lambda pose: float(np.sqrt(((orclib.project_images(pose, xyz, img, fr.Ks, fr.cams) - uv) ** 2).sum(1)).mean())
            if e(g["pose"]) > e(p) + 1.0: w.append(f"model {m}: pose {e(g['pose']):.2f} px vs oracle {e(p):.2f}")
            if abs(g["score"] - s_) > 0.05 * s_ + 1e-3: w.append(f"model {m}: score {g['score']:.3f} vs {s_:.3f}")
        return w
    if pts >= 60:
        w = differ(om, op, osc)
        if w:
            for alt in (1, 2, 3, 4, 5):
                if not differ(*oracle(seed + 7919 * alt)[:3]):
                    print(f"note scene {sc}: " + "; ".join(w) + f" -- the oracle gives the device's outcome with seed + {7919 * alt}", flush=True)
                    w = []
                    spread += 1
                    break
        soft = [x for x in w if ": score " in x]
        if soft and len(soft) == len(w):
            print(f"score scene {sc}: cameras {n_cam} models {n_models}x{ppm} Q={Q} n_vis={n_vis} pts={pts}: " + "; ".join(soft), flush=True)
            scores += 1
            w = []
        why += w
    if why:
        bad += 1
        print(f"MISMATCH scene {sc}: cameras {n_cam} models {n_models}x{ppm} Q={Q} n_vis={n_vis} pts={pts}: " + "; ".join(why), flush=True)
    c.frame_set_images(0)
    pipe.close()
    if sc % 10 == 9: print(f"{sc + 1} scenes, {bad} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{scenes} scenes with several images, {bad} mismatches ({spread} object-level differences inside the oracle's seed-to-seed spread; "
      f"{scores} scenes with an object whose FILTER2 score differs by more than 5% at a pose inside the 1 px bar)")
sys.exit(1 if bad else 0)
