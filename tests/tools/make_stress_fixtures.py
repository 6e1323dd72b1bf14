"""tests/golden/stress_scenes.json: the scenes the 2 x 1000-scene frame stress of round 4 / 5 singled out, as data -- each
scene's generator parameters (tests/tools/stress_scene.py replays frame_stress.py's draws) and what the ORACLE makes of
it on one thread (deterministic: libc rand() seeded per scene).  CPU only.
usage: python tests/tools/make_stress_fixtures.py > tests/golden/stress_scenes.json"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import orclib, stress_scene
from moped_amd import synth
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY
SCENES = [
    # (stress seed, scene, class, what the stress run of that round said)
    (0, 241, "marginal", "r04: an object only the oracle reports -- 18 clean matches of model 1 that mean shift splits into clusters of 8, 9, 8 "
                         "with 5, 6, 5 of them each; the reference passes ':204' on the middle one through a five-point fit that a match "
                         "12 px off drags towards itself (7 'inliers'), FILTER then hands the object all 18.  r05: the near-miss rescue "
                         "(fit over the near miss's inliers + the points within 25 thresholds, then n_pts_align-point samples) finds it."),
    (0, 139, "score", "r04: FILTER2 score 24.56 vs the oracle's 28.43 at a pose inside the bar -- one match with 7.9 px^2 under the planted pose "
                      "sits among the max-consensus winner's inliers and the fourth-power refine bends towards it (clean inliers 0.57 px "
                      "instead of 0.33).  r05: odd replicas refine over the inliers within half the threshold, FILTER keeps the better."),
    (0, 209, "score", "as 139 (29.07 vs 34.37; the match: 6.9 px^2)"),
    (0, 460, "score", "as 139 (53.04 vs 56.21; 4.6 px^2)"),
    (0, 561, "score", "as 139 (27.37 vs 30.71; 5.8 px^2)"),
    (0, 90, "score", "as 139 (11.52 vs 14.24; 9.0 px^2; r04 listed it as inside the device's seed-to-seed spread)"),
    (0, 335, "score", "as 139 (37.66 vs 46.16; 14.9 px^2; r04: inside the device's spread)"),
    (1, 120, "sensitive", "r05: a planted object only the device reports -- model 2's 7 good matches sit in a 44-point cluster; the reference's "
                          "600 five-point draws per replica land on five of the seven with probability 6%, the 4 x 1024 three-point "
                          "samples here almost surely.  The object meets ':204' under its planted pose."),
]
out = []
for stress_seed, scene, cls, why in SCENES:
    p = stress_scene.scene_params(scene, stress_seed)
    db, fr = stress_scene.build(p)
    idx, d1, d2 = orclib.match_2nn(orclib.normalize(db.desc), orclib.normalize(fr.desc))
    om, op, osc, oc, oinl = orclib.frame_rest_inliers(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0, n_threads=1, seed=p["seed"])
    out.append(dict(params=p, class_=cls, why=why, visible=[int(v) for v in fr.visible],
                    oracle=dict(counts=[int(c) for c in oc], models=[int(m) for m in om], scores=[round(float(s), 3) for s in osc],
                                inliers=[len(i) for i in oinl])))
print(json.dumps(out, indent=1))
