"""BASELINE config 2 at full size (200 models, 1 M descriptors, Q = 3000): the oracle cannot redo the
whole search in seconds, so parity is checked through size-independent properties --
  * a random sample of the queries against the oracle's exact search (bit-exact),
  * eight model shards + merge == the single-shard search (the exchange-1 contract),
  * every planted keypoint finds the row it was rendered from,
  * the whole frame finds the planted objects at their planted poses."""
import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY


@pytest.fixture(scope="module")
def big():
    db = synth.make_db(200, 5000)
    fr = synth.make_frame(db, n_vis=5, seed=77)
    return db, fr


def test_config2_match_properties(big):
    import torch
    db, fr = big
    c = capi.Context(0)
    dbn = c.normalize(db.desc)                     # A1 on the device (bit-exact vs the oracle: test_gpu_steps)
    qn = orclib.normalize(fr.desc)
    Q = qn.shape[0]
    dev = torch.device("cuda:0")
    tq = torch.from_numpy(qn).to(dev)
    qnorm = torch.from_numpy(orclib.row_norms(qn)).to(dev)
    out = [torch.empty(Q, dtype=t, device=dev) for t in (torch.int32, torch.float32, torch.float32)]
    c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
    c.match_local_dev(tq.data_ptr(), qnorm.data_ptr(), Q, *[o.data_ptr() for o in out])
    c.synchronize()
    gi, g1, g2 = [o.cpu().numpy() for o in out]
    # (1) a sample of queries against the oracle's exact search over all 1 M rows
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(Q, 48, replace=False))
    oi, od1, od2 = orclib.match_2nn(dbn, qn[pick])
    assert np.array_equal(gi[pick], oi) and np.array_equal(g1[pick], od1) and np.array_equal(g2[pick], od2)
    # (2) eight shards by model + merge == one shard
    S = 8
    idx_s = torch.empty((S, Q), dtype=torch.int32, device=dev)
    d1_s = torch.empty((S, Q), dtype=torch.float32, device=dev)
    d2_s = torch.empty((S, Q), dtype=torch.float32, device=dev)
    rows = db.n // S
    for s in range(S):
        lo, hi = s * rows, (s + 1) * rows
        c.db_upload(dbn[lo:hi], db.model_of[lo:hi], db.xyz[lo:hi], db.n_models, index_base=lo)
        c.match_local_dev(tq.data_ptr(), qnorm.data_ptr(), Q, idx_s[s].data_ptr(), d1_s[s].data_ptr(), d2_s[s].data_ptr())
        c.synchronize()
    mi = torch.empty(Q, dtype=torch.int32, device=dev)
    m1 = torch.empty(Q, dtype=torch.float32, device=dev)
    m2 = torch.empty(Q, dtype=torch.float32, device=dev)
    c.match_merge_dev(idx_s.data_ptr(), d1_s.data_ptr(), d2_s.data_ptr(), S, Q, mi.data_ptr(), m1.data_ptr(), m2.data_ptr())
    c.synchronize()
    assert np.array_equal(mi.cpu().numpy(), gi) and np.array_equal(m1.cpu().numpy(), g1) and np.array_equal(m2.cpu().numpy(), g2)
    # (3) planted keypoints (model descriptor + N(0, 0.01^2) noise) find the row they were rendered from
    planted = np.nonzero(fr.src_point >= 0)[0]
    assert (gi[planted] == fr.src_point[planted]).mean() > 0.99
    assert np.all(g1 >= 0) and np.all(g2 >= g1)
    c.close()


def test_config2_frame_finds_the_planted_objects(big):
    import torch
    from moped_amd.pipeline import FramePipeline, ShardedDB
    db, fr = big
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000)
    dev = torch.device("cuda:0")
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=5)
    objs, counts = pipe.fetch(0)
    assert set(objs["model"].tolist()) == set(fr.visible.tolist()) and len(objs) == len(fr.visible)
    for o in objs:
        j = list(fr.visible).index(o["model"])
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
        rows = rows[db.model_of[fr.src_point[rows]] == o["model"]]
        e = np.sqrt(((orclib.project(o["pose"], db.xyz[fr.src_point[rows]], K, CAM0) - fr.uv[rows]) ** 2).sum(1)).mean()
        assert e < 1.0                                            # mean reprojection error on the planted points
        assert np.linalg.norm(o["pose"][4:] - fr.poses[j][4:]) < 0.01
    pipe.close()
