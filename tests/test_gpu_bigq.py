"""Frames larger than the bench's: 9000 queries (three compaction super-passes in `group_kernel`) and more than 2048
accepted matches (its scans leave the LDS copies) -- the per-model match lists must still be the oracle's, in order,
and every planted object is found."""
import numpy as np
import pytest

import orclib
from moped_amd import synth
from moped_amd.pipeline import FramePipeline, ShardedDB

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_models,n_vis,Q,pts", [(12, 10, 9000, 300), (12, 3, 5000, 140)])
def test_large_frames_keep_the_oracles_match_lists(n_models, n_vis, Q, pts):
    import torch
    db = synth.make_db(n_models, 3000)
    fr = synth.make_frame(db, n_vis=n_vis, seed=9, Q=Q, pts_per_obj=pts)
    dev = torch.device("cuda:0")
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=Q)
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=3)
    objs, counts = pipe.fetch(0)
    got_q, got_m = pipe.ctxs[0].frame_fetch_matches()
    pipe.close()
    idx, d1, d2 = orclib.match_2nn(orclib.normalize(db.desc), orclib.normalize(fr.desc), n_threads=8)
    ok = (idx >= 0) & (d1 < np.float32(0.8) * d2)
    model = db.model_of[np.maximum(idx, 0)]
    qs = np.nonzero(ok)[0]
    qs = qs[np.lexsort((qs, model[qs]))]                       # matches[model] lists, ascending query
    assert np.array_equal(got_q, qs.astype(np.int32)) and np.array_equal(got_m, model[qs].astype(np.int32))
    assert counts[0] == len(qs)
    assert set(objs["model"].tolist()) == set(fr.visible.tolist())
