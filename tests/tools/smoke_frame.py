"""One small frame through the HIP path on cuda:0, checked against the CPU oracle.
Called by __graft_entry__.smoke().  Lives under tests/ because it imports oracle/ (the
checker): nothing inside the moped_amd package does."""
import numpy as np


def run(n_models: int = 4, pts_per_model: int = 1500, Q: int = 800, verbose: bool = True):
    import torch
    import orclib  # oracle/ (test infrastructure)
    from moped_amd import capi, synth
    from moped_amd.pipeline import FramePipeline, ShardedDB

    db = synth.make_db(n_models, pts_per_model)
    fr = synth.make_frame(db, n_vis=2, seed=3, Q=Q, pts_per_obj=120)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=Q)
    dev = torch.device("cuda:0")
    q_desc = torch.from_numpy(fr.desc).to(dev)
    q_uv = torch.from_numpy(fr.uv).to(dev)
    pipe.enqueue(0, q_desc, q_uv, seed=7)
    objs, counts = pipe.fetch(0)

    # oracle on the same inputs
    dbn = orclib.normalize(db.desc)
    qn = orclib.normalize(fr.desc)
    assert np.array_equal(q_desc.cpu().numpy().view(np.uint32), qn.view(np.uint32)), "A1 normalise mismatch"
    idx, d1, d2 = orclib.match_2nn(dbn, qn)
    om, op, osc, ocounts, oinl = orclib.frame_rest_inliers(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models,
                                                           synth.K_DEFAULT, synth.CAM_IDENTITY, n_threads=1, seed=1)
    assert counts[0] == ocounts[0], f"match count {counts[0]} != oracle {ocounts[0]}"
    assert counts[1] == ocounts[1], f"cluster count {counts[1]} != oracle {ocounts[1]}"
    assert sorted(objs["model"].tolist()) == sorted(om.tolist()), (objs["model"], om)
    for m, p, inl in zip(om, op, oinl):
        g = objs[objs["model"] == m][0]
        # the pose bar over the oracle's inlier set, then over the planted points
        xyz, uv = db.xyz[idx[inl]], fr.uv[inl]
        e_o = np.sqrt(((orclib.project(p, xyz, synth.K_DEFAULT, synth.CAM_IDENTITY) - uv) ** 2).sum(1)).mean()
        e_g = np.sqrt(((orclib.project(g["pose"], xyz, synth.K_DEFAULT, synth.CAM_IDENTITY) - uv) ** 2).sum(1)).mean()
        assert len(inl) >= 7 and e_g <= e_o + 1.0, (m, len(inl), e_g, e_o)
        rows = np.nonzero((fr.src_point >= 0) & (db.model_of[np.maximum(fr.src_point, 0)] == m) & ~fr.is_outlier)[0]
        xyz = db.xyz[fr.src_point[rows]]
        uv = fr.uv[rows]
        e_o = np.sqrt(((orclib.project(p, xyz, synth.K_DEFAULT, synth.CAM_IDENTITY) - uv) ** 2).sum(1)).mean()
        e_g = np.sqrt(((orclib.project(g["pose"], xyz, synth.K_DEFAULT, synth.CAM_IDENTITY) - uv) ** 2).sum(1)).mean()
        assert e_g <= e_o + 1.0, (m, e_g, e_o)
    if verbose:
        print(f"smoke ok: {counts[0]} matches, {counts[1]} clusters, {len(objs)} objects "
              f"(models {objs['model'].tolist()}), oracle agrees")
    pipe.close()
    return objs
