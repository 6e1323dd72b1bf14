"""What the device and the oracle (one thread: deterministic) make of given scenes of frame_stress.py, point by point:
for every object the two sides report, the squared reprojection errors of the model's accepted matches under the
device's, the oracle's and the planted pose, and which points make the difference in FILTER2's score.
usage: stress_fixture_probe.py scene [scene ...]      (needs a GPU)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import orclib, stress_scene
from moped_amd import capi, synth
from moped_amd.pipeline import FramePipeline, ShardedDB
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY
np.set_printoptions(precision=2, suppress=True, linewidth=220)
dev = torch.device("cuda:0")
for sc in [int(a) for a in sys.argv[1:]]:
    p = stress_scene.scene_params(sc)
    db, fr = stress_scene.build(p)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=4000)
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=p["seed"])
    objs, counts = pipe.fetch(0)
    pipe.close()
    idx, d1, d2 = orclib.match_2nn(orclib.normalize(db.desc), orclib.normalize(fr.desc))
    om, op, osc, oc, oinl = orclib.frame_rest_inliers(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0, n_threads=1, seed=p["seed"])
    out_q, off = orclib.match_accept(idx, d1, d2, 0.8, db.model_of, db.n_models)
    print(f"== scene {sc}: {p}")
    print("   device", [(int(o['model']), round(float(o['score']), 3), int(o['n_points'])) for o in objs], counts.tolist())
    print("   oracle", [(int(m), round(float(s), 3)) for m, s in zip(om, osc)], list(map(int, oc)))
    for m, po, so in zip(om, op, osc):
        g = objs[objs["model"] == m]
        if not len(g):
            print(f"   model {m}: only the oracle")
            continue
        g = g[0]
        if abs(g["score"] - so) <= 0.05 * so:
            continue
        q = out_q[off[m]:off[m + 1]]
        uv, xyz = fr.uv[q], db.xyz[idx[q]]
        e2 = lambda pose: ((orclib.project(pose, xyz, K, CAM0) - uv) ** 2).sum(1)
        pl = fr.poses[list(fr.visible).index(m)]
        ed, eo, ep = e2(g["pose"]), e2(po), e2(pl)
        near = (ed < 4096) | (eo < 4096)
        print(f"   model {m}: score device {g['score']:.3f} oracle {so:.3f}; points within FeatureDistance of either pose: {int(near.sum())}")
        print("     device e2 ", ed[near])
        print("     oracle e2 ", eo[near])
        print("     planted e2", ep[near])
        print("     score terms device - oracle", (1 / (ed[near] + 1) - 1 / (eo[near] + 1)))
        print("     sum over true inliers (planted e2 < 1): device", float((1 / (ed[ep < 1] + 1)).sum()), "oracle", float((1 / (eo[ep < 1] + 1)).sum()),
              "; mean px error over them: device", float(np.sqrt(ed[ep < 1]).mean()), "oracle", float(np.sqrt(eo[ep < 1]).mean()))
