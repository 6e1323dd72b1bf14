"""Round-robin model -> rank assignment (interleaving spreads POSE; mh_db_upload_blocks): a shard's rows
are several runs of global rows, every index the context reports stays global, ties break as in the unsharded DB, and
the W shards' frames add up -- bit for bit -- to the single-context frame.  The reference searches ONE tree over all
models (moped2/libmoped/src/match/MATCH_ANN_CPU.hpp:76-107)."""
import numpy as np
import pytest

from moped_amd import capi, synth
from moped_amd.pipeline import ShardedDB

pytestmark = pytest.mark.gpu
N_MODELS, PPM, Q = 7, 1400, 1300


def _canon(x):
    key = [(int(o["model"]),) + tuple(o["pose"].view(np.uint32).tolist()) for o in x]
    return x[sorted(range(len(x)), key=lambda i: key[i])]


@pytest.mark.parametrize("world", [2, 3])
def test_round_robin_shards_add_up_to_the_single_frame(world):
    import torch
    dev = torch.device("cuda:0")
    db = synth.make_db(N_MODELS, PPM)
    fr = synth.make_frame(db, n_vis=3, seed=5, Q=Q, pts_per_obj=120)
    prm = capi.default_frame_params()
    one = capi.Context(0)
    dbn = one.normalize(db.desc)
    one.db_upload(dbn, db.model_of, db.xyz, db.n_models)
    one.reserve(Q)
    qd, uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
    one.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, 7)
    ref_objs, ref_counts = one.frame_fetch()
    ref_q, ref_m = one.frame_fetch_matches()
    acc, raw, d1, d2 = one.match(one.normalize(fr.desc))
    assert len(ref_objs) >= 3
    # the shards
    ctxs, tops = [], []
    for r in range(world):
        sh = ShardedDB(dbn, db.xyz, db.model_of, db.n_models, r, world, assign="round-robin")
        assert len(sh.block_rows) > 1                       # really several runs of global rows
        c = capi.Context(0)
        sh.upload(c, sh.desc)
        c.reserve(Q)
        # the per-step entry point reports GLOBAL rows
        a_r, raw_r, d1_r, d2_r = c.match(one.normalize(fr.desc))
        own = np.isin(raw, sh.rows)
        assert np.array_equal(raw_r[own], raw[own]) and np.all(np.isin(raw_r, sh.rows))
        top = torch.empty(3 * Q, dtype=torch.int32, device=dev)
        qd_r = torch.from_numpy(fr.desc).to(dev)
        c.frame_enqueue_match_local(qd_r.data_ptr(), Q, top.data_ptr())
        c.synchronize()
        ctxs.append(c)
        tops.append(top)
    gathered = torch.cat(tops)
    objs, n_matches, n_clusters = [], 0, 0
    for r, c in enumerate(ctxs):
        c.frame_enqueue_rest(uv.data_ptr(), Q, gathered.data_ptr(), world, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, 7)
        o, cnt = c.frame_fetch()
        mq, mm = c.frame_fetch_matches()
        assert np.all(mm % world == r)                      # only matches of models this rank owns
        sel = ref_m % world == r
        assert np.array_equal(mq, ref_q[sel]) and np.array_equal(mm, ref_m[sel])
        objs.append(o)
        n_matches += int(cnt[0])
        n_clusters += int(cnt[1])
        c.close()
    one.close()
    assert n_matches == int(ref_counts[0]) and n_clusters == int(ref_counts[1])
    got, ref = _canon(np.concatenate(objs)), _canon(ref_objs)
    assert len(got) == len(ref) and np.array_equal(got["model"], ref["model"])
    assert np.array_equal(got["pose"].view(np.uint32), ref["pose"].view(np.uint32))
    assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))


def test_block_tables_are_checked():
    c = capi.Context(0)
    db = synth.make_db(2, 300)
    with pytest.raises(capi.MhError):   # descending
        c.db_upload_blocks(db.desc, db.model_of, db.xyz, 2, [300, 0], [300, 300])
    with pytest.raises(capi.MhError):   # rows do not add up
        c.db_upload_blocks(db.desc, db.model_of, db.xyz, 2, [0, 400], [300, 200])
    with pytest.raises(capi.MhError):   # overlapping
        c.db_upload_blocks(db.desc, db.model_of, db.xyz, 2, [0, 200], [300, 300])
    c.close()
