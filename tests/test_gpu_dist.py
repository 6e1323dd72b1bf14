"""Rehearsal of the N > 1 path on ONE GPU: two processes, each with its own model
shard and mh_ctx on cuda:0, gloo for the two exchanges (RCCL cannot put two ranks on
one device).  Everything except the collective's transport is the code bench.py runs
with --gpus N: mh_frame_enqueue_match_local -> exchange 1 -> mh_frame_enqueue_rest ->
exchange 2.  The merged result must equal the single-context frame."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pytestmark = pytest.mark.gpu

N_MODELS, PPM, Q = 8, 1500, 1200


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from moped_amd import synth
    from moped_amd.pipeline import FramePipeline, ShardedDB
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    db = synth.make_db(N_MODELS, PPM)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models, rank, world), depth=1, max_queries=Q)
    dev = torch.device("cuda:0")
    res = {}
    for seed in (0, 1):
        fr = synth.make_frame(db, n_vis=3, seed=seed, Q=Q, pts_per_obj=120)
        pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=seed + 5)
        if seed == 1:
            res["prev0"] = pipe.previous_objects(0)    # frame 0's objects, carried by frame 1's exchange
        objs = pipe.gather_objects(0)
        local, counts = pipe.fetch(0)
        res[f"objs{seed}"] = objs
        res[f"local{seed}"] = local
        res[f"counts{seed}"] = counts
    # the same two frames as ONE batch: one MATCH launch over both frames' queries, one exchange
    frs = [synth.make_frame(db, n_vis=3, seed=seed, Q=Q, pts_per_obj=120) for seed in (0, 1)]
    qd = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
    quv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
    pipe.enqueue_batch(0, qd, quv, 2, [5, 6])
    merged = pipe.flush_objects_batch(0, 2)
    res["batch0"], res["batch1"] = merged[0], merged[1]
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **res)
    pipe.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_equals_single_context(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orclib
    from moped_amd import synth
    from moped_amd.pipeline import FramePipeline, ShardedDB
    world = 2
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    db = synth.make_db(N_MODELS, PPM)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=Q)
    dev = torch.device("cuda:0")
    z = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for r in range(world):   # both frames as one batch (one MATCH launch, one exchange): the same merged objects
        for seed in (0, 1):
            assert np.array_equal(z[r][f"batch{seed}"]["model"], z[r][f"objs{seed}"]["model"])
            assert np.array_equal(z[r][f"batch{seed}"]["pose"], z[r][f"objs{seed}"]["pose"])
    for r in range(world):   # exchange 2 riding on the next frame's exchange 1 == the explicit gather
        assert np.array_equal(z[r]["prev0"]["model"], z[r]["objs0"]["model"])
        assert np.array_equal(z[r]["prev0"]["pose"], z[r]["objs0"]["pose"])
    for seed in (0, 1):
        fr = synth.make_frame(db, n_vis=3, seed=seed, Q=Q, pts_per_obj=120)
        pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=seed + 5)
        single, counts = pipe.fetch(0)
        merged = z[0][f"objs{seed}"]
        assert np.array_equal(z[1][f"objs{seed}"]["model"], merged["model"])       # every rank sees the same list
        # accepted matches: each kept by exactly one rank
        assert z[0][f"counts{seed}"][0] + z[1][f"counts{seed}"][0] == counts[0]
        assert z[0][f"counts{seed}"][1] + z[1][f"counts{seed}"][1] == counts[1]
        assert sorted(merged["model"].tolist()) == sorted(single["model"].tolist())
        assert set(single["model"].tolist()) == set(fr.visible.tolist())
        # rank r only reports models it owns
        for r in range(world):
            lo, hi = (r * N_MODELS) // world, ((r + 1) * N_MODELS) // world
            assert all(lo <= m < hi for m in z[r][f"local{seed}"]["model"])
        for o in merged:
            s = single[single["model"] == o["model"]][0]
            rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
            rows = rows[db.model_of[fr.src_point[rows]] == o["model"]]
            xyz, uv = db.xyz[fr.src_point[rows]], fr.uv[rows]
            e_m = np.sqrt(((orclib.project(o["pose"], xyz, synth.K_DEFAULT, synth.CAM_IDENTITY) - uv) ** 2).sum(1)).mean()
            e_s = np.sqrt(((orclib.project(s["pose"], xyz, synth.K_DEFAULT, synth.CAM_IDENTITY) - uv) ** 2).sum(1)).mean()
            assert e_m < 1.0 and abs(e_m - e_s) < 0.5
    pipe.close()


def _rccl_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from moped_amd import synth
    from moped_amd.pipeline import FramePipeline, ShardedDB
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    db = synth.make_db(N_MODELS, PPM)
    dev = torch.device("cuda:0")
    res = {}
    for tag, force in (("direct", False), ("rccl", True)):
        pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models, rank, world), depth=2,
                             max_queries=Q, force_exchange=force)
        frames = [synth.make_frame(db, n_vis=3, seed=s, Q=Q, pts_per_obj=120) for s in range(4)]
        qd = [torch.from_numpy(f.desc).to(dev) for f in frames]
        uv = [torch.from_numpy(f.uv).to(dev) for f in frames]
        # two rounds over both slots: the second reuses the slot's send/receive blocks
        for i in range(4):
            if i >= 2:
                objs, counts = pipe.fetch(i % 2)
                res[f"{tag}_objs{i - 2}"], res[f"{tag}_counts{i - 2}"] = objs, counts
            pipe.enqueue(i % 2, qd[i], uv[i], seed=i + 5)
            if force and i >= 2:   # exchange 2 rode along: the slot's previous frame, from all ranks
                res[f"{tag}_prev{i - 2}"] = pipe.previous_objects(i % 2)
        for i in (2, 3):
            if force and i == 3:
                res[f"{tag}_gathered"] = pipe.gather_objects(i % 2)
            objs, counts = pipe.fetch(i % 2)
            res[f"{tag}_objs{i}"], res[f"{tag}_counts{i}"] = objs, counts
        if force:
            # frames 0..3 again as two batches of two (one MATCH launch + one exchange per batch): same objects
            keep = []   # the inputs must outlive the enqueued work (torch would hand a freed tensor's memory out again)
            def fresh(a, b):
                keep.append(torch.cat([torch.from_numpy(frames[k].desc) for k in range(a, b)]).to(dev))
                keep.append(torch.cat(uv[a:b]))
                return keep[-2], keep[-1]
            for g in range(2):
                pipe.enqueue_batch(g, *fresh(2 * g, 2 * g + 2), 2, [2 * g + 5, 2 * g + 6])
            for g in range(2):
                for f, (objs, counts) in enumerate(pipe.fetch_batch(g, 2)):
                    res[f"batch_objs{2 * g + f}"], res[f"batch_counts{2 * g + f}"] = objs, counts
            # a second batch in slot 0 carries the first one's objects; the last batch is flushed explicitly
            pipe.enqueue_batch(0, *fresh(2, 4), 2, [7, 8])
            prev = pipe.previous_objects_batch(0)
            res["batch_prev0"], res["batch_prev1"] = prev[0], prev[1]
            fl = pipe.flush_objects_batch(0, 2)
            res["batch_flush2"], res["batch_flush3"] = fl[0], fl[1]
        pipe.close()
    np.savez(os.path.join(out_dir, "rccl.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_exchange_path_world1_equals_direct(tmp_path):
    """The N > 1 code path (match_local -> RCCL all-gather on the slot's stream -> rest,
    exchange 2) with the real `nccl` backend and a single rank: same objects, bit for bit,
    as the fused single-context frame."""
    port = 29900 + (os.getpid() % 1000)
    mp.spawn(_rccl_worker, args=(0 + 1, port, str(tmp_path)), nprocs=1, join=True)
    z = np.load(os.path.join(str(tmp_path), "rccl.npz"))
    for i in range(4):
        assert np.array_equal(z[f"direct_counts{i}"], z[f"rccl_counts{i}"])
        a, b = z[f"direct_objs{i}"], z[f"rccl_objs{i}"]
        assert len(a) == len(b) and len(a) >= 3
        assert np.array_equal(a["model"], b["model"])
        assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["score"], b["score"])
    for i in range(4):   # batches of two frames through one MATCH launch and one exchange: the same objects
        a, b = z[f"direct_objs{i}"], z[f"batch_objs{i}"]
        assert np.array_equal(z[f"direct_counts{i}"], z[f"batch_counts{i}"])
        assert np.array_equal(a["model"], b["model"]) and np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["score"], b["score"])
    for i in range(2):
        assert np.array_equal(z[f"batch_prev{i}"]["pose"], z[f"direct_objs{i}"]["pose"])
        assert np.array_equal(z[f"batch_flush{i + 2}"]["pose"], z[f"direct_objs{i + 2}"]["pose"])
    for i in range(2):   # the objects that rode on the next frame's exchange are the frame's own
        p = z[f"rccl_prev{i}"]
        assert np.array_equal(p["model"], z[f"rccl_objs{i}"]["model"]) and np.array_equal(p["pose"], z[f"rccl_objs{i}"]["pose"])
    g = z["rccl_gathered"]
    assert np.array_equal(g["model"], z["rccl_objs3"]["model"]) and np.array_equal(g["pose"], z["rccl_objs3"]["pose"])
