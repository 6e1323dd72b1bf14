"""A cluster of a depth frame through mh_pose_ransac_depth with many seeds, against the oracle's depth RANSAC on the same points.
usage: depth_cluster_debug.py [frame seed] [model rank among the visible]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
import orclib
from moped_amd import capi, synth
np.set_printoptions(precision=4, suppress=True, linewidth=220)
db = synth.make_db(50, 5000)
s = int(sys.argv[1]) if len(sys.argv) > 1 else 6
j = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fr = synth.make_frame(db, n_vis=2, seed=s, Q=3000)
wpts, fill = synth.frame_depth(db, fr, seed=s)
wgt = orclib.cauchy_weight(fill, 0.1)
K, CAM = synth.K_DEFAULT, synth.CAM_IDENTITY
c = capi.Context(0)
dbn = c.normalize(db.desc)
c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
acc_idx, raw, d1, d2 = c.match(c.normalize(fr.desc), ratio=0.8)
m = int(fr.visible[j])
a = np.nonzero((acc_idx >= 0) & (db.model_of[np.maximum(acc_idx, 0)] == m))[0]
uv, xyz = fr.uv[a], db.xyz[acc_idx[a]]
cl = orclib.meanshift(uv, 200.0, 20.0, 7, 100)[0]
big = np.asarray(max(cl, key=len))
print("model", m, "matches", len(a), "clusters", [len(x) for x in cl])
pl = fr.poses[j]
rows = a[big]
planted = (fr.src_point[rows] >= 0) & ~fr.is_outlier[rows]
e_pl = np.sqrt(((orclib.project(pl, xyz[big], K, CAM) - uv[big]) ** 2).sum(1))
print("cluster", len(big), "planted inliers", int(planted.sum()), "points within 2.83 px of the planted pose", int((e_pl < 8 ** 0.5).sum()),
      "of them not planted inliers", int(((e_pl < 8 ** 0.5) & ~planted).sum()))
# depth consistency of the cluster's points under the planted pose: camera-frame point vs the depth attribute
R = synth.quat_to_R(pl[:4])
pc = xyz[big].astype(np.float64) @ R.T + pl[4:]
dd = np.linalg.norm(pc - wpts[rows], axis=1)
print("3-D distance attribute <-> model point under the planted pose, cluster points inside 2.83 px: max %.4f m, those above 2 cm: %s" %
      (dd[e_pl < 8 ** 0.5].max(), np.round(dd[(e_pl < 8 ** 0.5) & (dd > 0.02)], 3).tolist()))
prm = capi.default_frame_params().pose1
prm.error_threshold = 8.0
corr = capi.pack_corr(uv[big], xyz[big])
dep = capi.pack_depth(wpts[rows], wgt[rows])
bad = 0
for sd in range(1, 33):
    out = c.pose_ransac_depth(corr, dep, [0, len(big)], K, CAM, prm, 1, 0.5, seed=sd)
    for o in out:
        e = np.sqrt(((orclib.project(o["pose"], xyz[big][planted], K, CAM) - uv[big][planted]) ** 2).sum(1)).mean()
        if e > 2.0:
            bad += 1
            if bad <= 4: print("  seed", sd, "inliers", int(o["n_inliers"]), "err", float(o["err"]), "pose", o["pose"], "mean reproj on planted %.2f" % e)
print("device: %d of %d (seed, replica) results more than 2 px off on the planted inliers" % (bad, 32 * prm.max_objects_per_cluster))
for kind in (0,):
    out = c.pose_ransac(corr, [0, len(big)], K, CAM, prm, seed=1)
    e = [np.sqrt(((orclib.project(o["pose"], xyz[big][planted], K, CAM) - uv[big][planted]) ** 2).sum(1)).mean() for o in out]
    print("device without depth residuals, seed 1:", np.round(e, 3).tolist())
c.close()
