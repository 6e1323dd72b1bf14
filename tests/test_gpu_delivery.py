"""Delivery of every frame's objects to the host inside a stream of batches (mh_frame_fetch_batch_async /
mh_frame_fetch_wait).  Reference behaviour: MopedPimpl::processImages hands each frame's list<SP_Object> to its caller
(moped2/libmoped/src/moped.cpp:166-194; moped2/moped_test.cpp:205-207 prints them) -- so a host that keeps batches in
flight must see EVERY frame's objects, not only those of the batches that happen to be last.  16 slots x 16 frames x
3 rounds, every delivered record bit for bit the frame alone (mh_frame_enqueue + mh_frame_fetch)."""
import numpy as np
import pytest

from moped_amd import capi, synth
from moped_amd.pipeline import FramePipeline, ShardedDB

pytestmark = pytest.mark.gpu
Q = 3000


def _same(rec, objs, counts):
    n = int(rec["head"]["n_objects"])
    o = rec["objects"][:n]
    return (n == len(objs) and int(rec["head"]["flags"]) == 0 and np.array_equal(rec["head"]["counts"], counts) and
            np.array_equal(o["model"], objs["model"]) and
            np.array_equal(o["pose"].view(np.uint32), objs["pose"].view(np.uint32)) and
            np.array_equal(o["score"].view(np.uint32), objs["score"].view(np.uint32)) and
            np.array_equal(o["n_points"], objs["n_points"]))


@pytest.fixture(scope="module")
def scene():
    import torch
    db = synth.make_db(20, 5000)
    n_vis = (2, 2, 5, 1, 2, 3, 0, 2, 4, 2, 0, 1, 2, 2, 3, 2, 10, 2, 1, 2, 2, 0, 2, 7, 2, 2, 1, 3, 2, 2, 2, 5)
    frames = [synth.make_frame(db, n_vis=n, seed=500 + i, Q=Q) for i, n in enumerate(n_vis)]
    return db, frames, torch


def test_every_batch_of_a_stream_is_delivered_and_equals_the_frames_alone(scene):
    db, frames, torch = scene
    dev = torch.device("cuda:0")
    B, slots, rounds = 16, 16, 3
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=slots, max_queries=B * Q, batch=B)
    pipe.attach_delivery(max_objects=16, B=B)
    pool_groups = len(frames) // B
    qd_pool = [torch.cat([torch.from_numpy(f.desc) for f in frames[g * B:(g + 1) * B]]).to(dev) for g in range(pool_groups)]
    uv_pool = [torch.cat([torch.from_numpy(f.uv) for f in frames[g * B:(g + 1) * B]]).to(dev) for g in range(pool_groups)]
    work = [torch.empty_like(qd_pool[0]) for _ in range(slots)]
    torch.cuda.synchronize()
    delivered = {}   # (round, slot) -> copy of the records
    pending = {}
    for r in range(rounds):
        for s in range(slots):
            if s in pending:   # the slot's previous batch reaches the host before the slot is reused
                delivered[pending.pop(s)] = pipe.take_delivery(s).copy()
            g = (r * slots + s) % pool_groups
            with torch.cuda.stream(pipe.streams[s]):
                work[s].copy_(qd_pool[g], non_blocking=True)
            seeds = [10_000 * r + 100 * s + f + 1 for f in range(B)]
            pipe.enqueue_batch(s, work[s], uv_pool[g], B, seeds)
            pipe.deliver(s, tag=(r << 8) | s)
            pending[s] = (r, s)
    for s, key in pending.items():
        delivered[key] = pipe.take_delivery(s).copy()
    assert len(delivered) == rounds * slots
    # every frame of every batch alone, on a context of its own
    c = capi.Context(0)
    c.db_share(pipe.ctxs[0])
    c.reserve(Q)
    one = torch.empty(Q, 128, dtype=torch.float32, device=dev)
    n_frames = n_objects = 0
    for (r, s), recs in sorted(delivered.items()):
        g = (r * slots + s) % pool_groups
        assert np.all(recs["head"]["tag"] == ((r << 8) | s)) and np.array_equal(recs["head"]["frame"], np.arange(B))
        for f in range(B):
            fr = frames[g * B + f]
            one.copy_(torch.from_numpy(fr.desc))
            c.frame_enqueue(one.data_ptr(), uv_pool[g][f * Q:(f + 1) * Q].data_ptr(), Q, synth.K_DEFAULT, synth.CAM_IDENTITY,
                            pipe.params, 10_000 * r + 100 * s + f + 1)
            objs, counts = c.frame_fetch()
            assert _same(recs[f], objs, counts), (r, s, f)
            n_frames += 1
            n_objects += len(objs)
    assert n_frames == rounds * slots * B and n_objects >= 1.5 * n_frames
    c.close()
    pipe.close()


def test_pageable_block_goes_through_the_staging_buffer_and_a_second_delivery_before_the_wait_is_refused(scene):
    db, frames, torch = scene
    dev = torch.device("cuda:0")
    B = 4
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=B * Q, batch=B)
    qd = torch.cat([torch.from_numpy(f.desc) for f in frames[:B]]).to(dev)
    uv = torch.cat([torch.from_numpy(f.uv) for f in frames[:B]]).to(dev)
    torch.cuda.synchronize()
    pipe.enqueue_batch(0, qd, uv, B, [1, 2, 3, 4])
    c = pipe.ctxs[0]
    cap = 3   # fewer than the frame with five objects has: n_objects still says how many there were
    pageable = np.zeros(B, capi.frame_block_dtype(cap))
    c.frame_fetch_batch_async(B, cap, pageable.ctypes.data, 77)
    with pytest.raises(capi.MhError):
        c.frame_fetch_batch_async(B, cap, pageable.ctypes.data, 78)
    c.frame_fetch_wait()
    for f in range(B):
        objs, counts = c.frame_fetch_slot(f)
        assert pageable[f]["head"]["n_objects"] == len(objs) and pageable[f]["head"]["tag"] == 77
        k = min(len(objs), cap)
        assert np.array_equal(pageable[f]["objects"][:k]["pose"].view(np.uint32), objs[:k]["pose"].view(np.uint32))
    assert c.frame_fetch_query()
    pipe.close()
