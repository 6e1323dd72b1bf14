"""Randomised frame-level parity: device-resident frames (mh_frame_enqueue, and the same frames through
mh_frame_enqueue_batch) against the oracle pipeline over random databases, query counts, numbers of visible objects,
points per object and outlier shares.  Exact: accepted matches, mean-shift clusters, detected model set; within the
north-star tolerance: poses (<= 1 px of the oracle's on the planted inliers); FILTER2 scores are compared at 5% and reported.
usage: frame_stress.py [scenes=60] [seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import orclib
from moped_amd import capi, synth
from moped_amd.pipeline import FramePipeline, ShardedDB
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY
dev = torch.device("cuda:0")
bad = 0
scores = 0
spread = 0
marginal = 0
sensitive = 0
done = 0
t0 = time.perf_counter()
while done < scenes:
    n_models, ppm = int(rng.choice([3, 6, 12, 20])), int(rng.choice([600, 1500, 3000]))
    db_seed = int(rng.integers(1 << 30))
    only = int(os.environ.get("FRAME_STRESS_ONLY", "-1"))   # look at one scene of the sequence (same random draws)
    if only >= 0 and not (done <= only < done + 6):
        for k in range(6):
            Q = int(rng.choice([300, 900, 2000, 3000, 4000])); n_vis = int(rng.integers(0, min(n_models, 8) + 1)); rng.choice([12, 40, 150, 300])
            rng.integers(1 << 30); rng.choice([0.0, 0.2, 0.5]); rng.integers(1, 1 << 20)
        done += 6
        continue
    db = synth.make_db(n_models, ppm, seed=db_seed)
    dbn = orclib.normalize(db.desc)
    Qmax = 4000
    prm = capi.default_frame_params()
    if "FRAME_STRESS_LM" in os.environ:   # experiment: LM iteration caps of both POSE stages
        prm.pose1.lm_iters_l2 = prm.pose1.lm_iters_l4 = prm.pose2.lm_iters_l2 = prm.pose2.lm_iters_l4 = int(os.environ["FRAME_STRESS_LM"])
    if "FRAME_STRESS_L2" in os.environ:   # experiment: cap of the plain-residual phase alone (0 = the squared-residual phase starts at the P3P pose)
        prm.pose1.lm_iters_l2 = prm.pose2.lm_iters_l2 = int(os.environ["FRAME_STRESS_L2"])
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=2, max_queries=Qmax, params=prm)
    pipe.ctxs[1].reserve(3 * Qmax)
    group = []
    for k in range(6):
        Q = int(rng.choice([300, 900, 2000, 3000, 4000]))
        n_vis = int(rng.integers(0, min(n_models, 8) + 1))
        pts = int(rng.choice([12, 40, 150, 300]))
        pts = min(pts, ppm // 2, Q // max(n_vis, 1))
        fr = synth.make_frame(db, n_vis=n_vis, seed=int(rng.integers(1 << 30)), Q=Q, pts_per_obj=pts,
                              outlier_frac=float(rng.choice([0.0, 0.2, 0.5])))
        seed = int(rng.integers(1, 1 << 20))
        q_desc, q_uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
        pipe.enqueue(0, q_desc, q_uv, seed=seed)
        objs, counts = pipe.fetch(0)
        qn = orclib.normalize(fr.desc)
        idx, d1, d2 = orclib.match_2nn(dbn, qn)
        om, op, osc, oc, oinl = orclib.frame_rest_inliers(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0, n_threads=8, seed=seed)
        why = []
        if counts[0] != oc[0]: why.append(f"matches {counts[0]} vs {oc[0]}")
        if counts[1] != oc[1]: why.append(f"clusters {counts[1]} vs {oc[1]}")
        # objects: RANSAC is randomised on both sides (the oracle keeps the reference's 5-point LM starts, the device
        # samples P3P hypotheses) -- compared where the planted objects are unambiguous (>= 40 points), and a
        # difference only counts when the oracle does not show the same outcome under another seed of its own
        def objects_differ(om, op, osc, oinl):
            w = []
            if sorted(objs["model"].tolist()) != sorted(om.tolist()):
                return [f"models {sorted(objs['model'].tolist())} vs {sorted(om.tolist())}"]
            for m, p, sc, inl in zip(om, op, osc, oinl):
                g = objs[objs["model"] == m][0]
                if len(inl) >= 7:   # the bar over the oracle's own inlier set
                    ei = lambda pose: float(np.sqrt(((orclib.project(pose, db.xyz[idx[inl]], K, CAM0) - fr.uv[inl]) ** 2).sum(1)).mean())
                    if ei(g["pose"]) > ei(p) + 1.0: w.append(f"model {m}: pose {ei(g['pose']):.2f} px vs oracle {ei(p):.2f} over the oracle's inliers")
                rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
                rows = rows[db.model_of[fr.src_point[rows]] == m]
                if len(rows) < 8: continue
                xyz, uv = db.xyz[fr.src_point[rows]], fr.uv[rows]
                e = lambda pose: float(np.sqrt(((orclib.project(pose, xyz, K, CAM0) - uv) ** 2).sum(1)).mean())
                if e(g["pose"]) > e(p) + 1.0: w.append(f"model {m}: pose {e(g['pose']):.2f} px vs oracle {e(p):.2f}")
                if abs(g["score"] - sc) > 0.05 * sc + 1e-3: w.append(f"model {m}: score {g['score']:.3f} vs {sc:.3f}")
            return w
        if pts >= 40:
            w = objects_differ(om, op, osc, oinl)
            if w:
                for alt in (1, 2, 3, 4, 5):   # the oracle's own spread
                    om2, op2, osc2, _, oinl2 = orclib.frame_rest_inliers(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0, n_threads=8, seed=seed + 7919 * alt)
                    if not objects_differ(om2, op2, osc2, oinl2):
                        print(f"note scene {done}: " + "; ".join(w) + f" -- the oracle gives the device's outcome with seed + {7919 * alt}", flush=True)
                        w = []
                        spread += 1
                        break
            if w:
                first = objs
                for alt in (1, 2, 3, 4, 5):   # ... and the device's
                    pipe.enqueue(0, q_desc, q_uv, seed=seed + 104729 * alt)
                    objs, _ = pipe.fetch(0)
                    if not objects_differ(om, op, osc, oinl):
                        print(f"note scene {done}: " + "; ".join(w) + f" -- the device gives the oracle's outcome with seed + {104729 * alt}", flush=True)
                        w = []
                        spread += 1
                        break
                objs = first
            # An object only the oracle reports, none of whose clusters holds MORE than MinNPtsObject correspondences within
            # POSE's threshold of the PLANTED pose: under the true pose it does not meet the reference's own acceptance rule
            # (:204); the reference accepts it through a least-squares fit of five points that a near-outlier dragged
            # towards itself (seven "inliers" where six exist).  The P3P stage does not produce such a pose.  Reported as
            # `marginal`, not failed.
            if w and len(w) == 1 and w[0].startswith("models "):
                dev_m, orc_m = sorted(first["model"].tolist()), sorted(om.tolist())
                extra = [m for m in set(orc_m) if orc_m.count(m) > dev_m.count(m)]
                if extra and not [m for m in set(dev_m) if dev_m.count(m) > orc_m.count(m)]:
                    mq_, mm_ = pipe.ctxs[0].frame_fetch_matches()
                    marginal_all = True
                    for m in extra:
                        if m not in fr.visible.tolist():
                            marginal_all = False
                            break
                        acc_ = mq_[mm_ == m]
                        pl_ = fr.poses[list(fr.visible).index(m)]
                        e2_ = ((orclib.project(pl_, db.xyz[idx[acc_]], K, CAM0) - fr.uv[acc_]) ** 2).sum(1)
                        cls_, _ = pipe.ctxs[0].meanshift(np.ascontiguousarray(fr.uv[acc_]))
                        best_ = max([int((e2_[np.asarray(c_)] < prm.pose1.error_threshold).sum()) for c_ in cls_] + [0])
                        if best_ > prm.pose1.min_n_pts_object:
                            marginal_all = False
                    if marginal_all:
                        print(f"marginal scene {done}: " + w[0] + f" -- no cluster of model(s) {extra} has more than {prm.pose1.min_n_pts_object} correspondences within "
                              f"{prm.pose1.error_threshold} px^2 of the planted pose", flush=True)
                        marginal += 1
                        w = []
            # The mirror image: an object only the DEVICE reports that is a planted, visible object with MORE than
            # MinNPtsObject correspondences of one cluster within POSE's threshold of its planted pose, found at a pose
            # inside the bar -- the reference's own acceptance rule (:204) holds for it, but its 600 draws of five points
            # rarely all land on the few good ones of a cluster of clutter (seed 1, scene 120: 7 good among 44 -- a chance
            # of (7/44)^5 per draw, 6% per replica; the 4 x 1024 three-point samples here meet it almost surely).
            # Reported as `sensitive`, not failed.
            if w and len(w) == 1 and w[0].startswith("models "):
                dev_m, orc_m = sorted(first["model"].tolist()), sorted(om.tolist())
                extra = [m for m in set(dev_m) if dev_m.count(m) > orc_m.count(m)]
                if extra and not [m for m in set(orc_m) if orc_m.count(m) > dev_m.count(m)]:
                    mq_, mm_ = pipe.ctxs[0].frame_fetch_matches()
                    ok_all, notes = True, []
                    for m in extra:
                        if m not in fr.visible.tolist() or dev_m.count(m) != 1:
                            ok_all = False
                            break
                        acc_ = mq_[mm_ == m]
                        pl_ = fr.poses[list(fr.visible).index(m)]
                        e2_ = ((orclib.project(pl_, db.xyz[idx[acc_]], K, CAM0) - fr.uv[acc_]) ** 2).sum(1)
                        cls_, _ = pipe.ctxs[0].meanshift(np.ascontiguousarray(fr.uv[acc_]))
                        good_ = [(int((e2_[np.asarray(c_)] < prm.pose1.error_threshold).sum()), len(c_)) for c_ in cls_]
                        g_ = first[first["model"] == m][0]
                        rows_ = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
                        rows_ = rows_[db.model_of[fr.src_point[rows_]] == m]
                        e_dev = float(np.sqrt(((orclib.project(g_["pose"], db.xyz[fr.src_point[rows_]], K, CAM0) - fr.uv[rows_]) ** 2).sum(1)).mean()) if len(rows_) else 9e9
                        if max([g for g, _ in good_] + [0]) <= prm.pose1.min_n_pts_object or e_dev > 1.0:
                            ok_all = False
                        notes.append(f"model {m}: clusters (good, size) {good_}, device pose {e_dev:.2f} px off the planted points")
                    if ok_all:
                        print(f"sensitive scene {done}: " + w[0] + " -- " + "; ".join(notes), flush=True)
                        sensitive += 1
                        w = []
            # what is left after reseeding both sides: a different model set or a pose outside the bar is a mismatch;
            # a FILTER2 score that differs by more than 5% while the pose is inside the bar is reported and counted,
            # not failed -- FILTER's arithmetic is bit-exact for equal poses (tests/test_gpu_steps.py), the score follows
            # the pose, and the two RANSACs pick their inlier sets differently on purpose
            soft = [x for x in w if ": score " in x]
            if soft and len(soft) == len(w):
                print(f"score scene {done}: models {n_models}x{ppm} Q={Q} n_vis={n_vis} pts={pts}: " + "; ".join(soft), flush=True)
                scores += 1
                w = []
            why += w
        group.append((fr, seed, objs, counts))
        if only == done:
            print("device objects:", [(int(o["model"]), round(float(o["score"]), 3), int(o["n_points"])) for o in objs], "counts", counts.tolist())
            print("oracle objects:", [(int(m), round(float(sc), 3)) for m, sc in zip(om, osc)], "counts", list(oc))
            dbg = int(os.environ.get("FRAME_STRESS_MODEL", "-1"))
            if dbg >= 0:   # every accepted match of that model under the device's, the oracle's and the planted pose
                mq_, mm_ = pipe.ctxs[0].frame_fetch_matches()
                acc = mq_[mm_ == dbg]                                # the frame's accepted matches of the model, in list order
                xyz, uv = db.xyz[idx[acc]], fr.uv[acc]
                e2 = lambda pose: ((orclib.project(pose, xyz, K, CAM0) - uv) ** 2).sum(1)
                pl = fr.poses[list(fr.visible).index(dbg)]
                g = objs[objs["model"] == dbg][0]["pose"] if (objs["model"] == dbg).any() else pl   # (a side without the object: the planted pose in its place)
                o = op[list(om).index(dbg)] if dbg in list(om) else pl
                np.set_printoptions(precision=2, suppress=True, linewidth=200)
                print("planted inlier?", (fr.src_point[acc] >= 0) & ~fr.is_outlier[acc])
                print("device  e2:", e2(g), "pose", g)
                print("oracle  e2:", e2(o), "pose", o)
                print("planted e2:", e2(pl), "pose", pl)
                # CLUSTER on its own over the model's accepted matches, then POSE on every cluster
                cls, _ = pipe.ctxs[0].meanshift(np.ascontiguousarray(uv))
                print("mean-shift clusters of the model's matches:", [len(c_) for c_ in cls])
                for c_ in cls:
                    c_ = np.asarray(c_)
                    print("   cluster", c_.tolist(), "e2 < 10 under the planted pose:", int((e2(pl)[c_] < 10).sum()), "under the oracle's:", int((e2(o)[c_] < 10).sum()),
                          "planted e2", np.round(e2(pl)[c_], 2).tolist())
                    orc1 = orclib.ransac(uv[c_], xyz[c_], K, CAM0, orclib.POSE1) if hasattr(orclib, "ransac") else None
                    if orc1 is not None: print("   oracle RANSAC on it:", orc1[0], "inliers under its pose:", int((e2(orc1[1])[c_] < 10).sum()) if orc1[0] else 0)
                    o1 = pipe.ctxs[0].pose_ransac(capi.pack_corr(uv[c_], xyz[c_]), [0, len(c_)], K, CAM0, prm.pose1, seed=seed)
                    print("   POSE on it:", [(int(x["n_inliers"]), round(float((1.0 / (e2(x["pose"]) + 1.0)).sum()), 2)) for x in o1])
                # the step on its own over exactly these matches as one cluster, POSE2's parameters
                outp = pipe.ctxs[0].pose_ransac(capi.pack_corr(uv, xyz), [0, len(uv)], K, CAM0, prm.pose2, seed=seed)
                for o_ in outp:
                    print("step pose_ransac: inliers", int(o_["n_inliers"]), "err", float(o_["err"]), "score-like", float((1.0 / (e2(o_["pose"]) + 1.0)).sum()), "pose", o_["pose"])
                po = orclib.ransac(uv, xyz, K, CAM0, orclib.POSE2) if hasattr(orclib, "ransac") else None
                if po is not None: print("oracle ransac:", po, "score-like", float((1.0 / (e2(po[1] if isinstance(po, tuple) else po) + 1.0)).sum()))
                print("device score-like", float((1.0 / (e2(g) + 1.0)).sum()), "oracle", float((1.0 / (e2(o) + 1.0)).sum()))
                # the frame again, stopped after POSE / after FILTER: what each side holds for this model there
                import copy
                for stop_after, label in ((0, "after POSE"),):
                    p1 = copy.copy(prm); p1.run_stage2 = stop_after
                    c0 = pipe.ctxs[0]
                    c0.frame_enqueue(q_desc.data_ptr(), q_uv.data_ptr(), Q, K, CAM0, p1, seed)
                    o1, c1 = c0.frame_fetch()
                    print(f"device {label}:", [(int(x["model"]), int(x["n_points"]), round(float(x["score"]), 2)) for x in o1], "counts", c1.tolist())
                    fo = orclib.default_frame_params(run_stage2=bool(stop_after))
                    om1, op1, osc1, oc1 = orclib.frame_rest(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0, params=fo, n_threads=1, seed=seed)
                    print(f"oracle {label}:", [int(m) for m in om1], "counts", list(oc1))
                mq, mm = pipe.ctxs[0].frame_fetch_matches()
                print("device matches of the model (queries):", mq[mm == dbg].tolist())
                print("oracle accepted (approx rule) queries:", acc.tolist())
        if why:
            bad += 1
            print(f"MISMATCH scene {done}: models {n_models}x{ppm} Q={Q} n_vis={n_vis} pts={pts}: " + "; ".join(why), flush=True)
        done += 1
    # the same frames, three at a time through one MATCH launch sequence: bit-identical objects
    same_q = {}
    for g in group: same_q.setdefault(len(g[0].desc), []).append(g)
    for Q, gs in same_q.items():
        for i in range(0, len(gs) - 1, 3):
            part = gs[i:i + 3]
            if len(part) < 2: continue
            qd = torch.cat([torch.from_numpy(g[0].desc) for g in part]).to(dev)
            uv = torch.cat([torch.from_numpy(g[0].uv) for g in part]).to(dev)
            pipe.enqueue_batch(1, qd, uv, len(part), [g[1] for g in part])
            for f, (objs, counts) in enumerate(pipe.fetch_batch(1, len(part))):
                a, ac = part[f][2], part[f][3]
                if not (np.array_equal(counts, ac) and np.array_equal(objs["model"], a["model"]) and
                        np.array_equal(objs["pose"].view(np.uint32), a["pose"].view(np.uint32))):
                    bad += 1
                    print(f"MISMATCH batch of Q={Q}: frame {f} differs from the frame alone", flush=True)
    pipe.close()
    print(f"{done} scenes, {bad} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{done} scenes, {bad} mismatches ({spread} object-level differences inside the seed-to-seed spread of the oracle or of the device; "
      f"{scores} scenes with an object whose FILTER2 score differs by more than 5% at a pose inside the 1 px bar; "
      f"{marginal} scenes with an object only the oracle reports that has no cluster of more than MinNPtsObject points within the threshold of its planted pose; "
      f"{sensitive} scenes with a planted object only the device reports, at a pose inside the bar, that meets the reference's acceptance rule)")
sys.exit(1 if bad else 0)
