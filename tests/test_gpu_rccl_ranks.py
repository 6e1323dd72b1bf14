"""RCCL with MORE THAN ONE rank (skipped on hosts with fewer than two GPUs: the one-GPU box runs the same entry points
over RCCL at world 1 and over the host transport at world 2..8, tests/test_gpu_comm.py).  Two processes, one GPU each,
torch.distributed "nccl" only for handing rank 0's communicator ids over; the frames' collectives are issued by
libmoped_hip.so on the slots' streams (csrc/comm.hip).  FramePipeline with more slots than communicators -- slot i
uses communicator i % n_comms, several streams share one communicator -- so the ncclAllGather calls of different slots
interleave on each communicator: correct only if both ranks issue them in the same order, which the sequence tags in
the exchanged blocks check on the device.  Objects of every frame (batches, piggy-backed previous objects, the flush
gather) must be the single-context frame's, bit for bit.  The loop this serves: moped2/libmoped/src/moped.cpp:166-194."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL with world > 1 needs two GPUs")]

N_MODELS, PPM, Q, B, DEPTH, N_COMMS = 9, 1500, 1200, 3, 6, 2


def _frames(db, synth, n):
    return [synth.make_frame(db, n_vis=3, seed=s, Q=Q, pts_per_obj=120) for s in range(n)]


def _worker(rank, world, port, out_dir, assign, G=0):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from moped_amd import synth
    from moped_amd.pipeline import FramePipeline, ShardedDB
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dev = torch.device(f"cuda:{rank}")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    db = synth.make_db(N_MODELS, PPM)
    if G:   # a models x frames grid: rank = r G + g, the communicators span the G ranks of frame group r only
        r, g = rank // G, rank % G
        pipe = FramePipeline(rank, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models, g, G, assign=assign), depth=DEPTH,
                             max_queries=Q * B, n_comms=N_COMMS, batch=B, id_leader=r * G)
        info = pipe.comm_info()
        assert info["world"] == G and info["rank"] == g and info["transport"].startswith("RCCL")
        frs = _frames(db, synth, B * DEPTH * (world // G))[r * B * DEPTH:(r + 1) * B * DEPTH]   # the group's own frames
    else:
        pipe = FramePipeline(rank, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models, rank, world, assign=assign), depth=DEPTH,
                             max_queries=Q * B, n_comms=N_COMMS, batch=B)
        info = pipe.comm_info()
        assert info["world"] == world and info["rank"] == rank and info["transport"].startswith("RCCL")
        frs = _frames(db, synth, B * DEPTH)
    res = {}
    keep = []
    for rnd in range(2):                      # two rounds over all slots: the second carries the first one's objects
        for slot in range(DEPTH):
            fs = frs[slot * B:(slot + 1) * B]
            qd = torch.cat([torch.from_numpy(f.desc) for f in fs]).to(dev)
            uv = torch.cat([torch.from_numpy(f.uv) for f in fs]).to(dev)
            keep.append((qd, uv))
            pipe.enqueue_batch(slot, qd, uv, B, [100 * rnd + slot * B + k + 1 for k in range(B)])
    for slot in range(DEPTH):
        prev = pipe.previous_objects_batch(slot)          # round 0's objects, all ranks', rode on round 1's exchange
        last = pipe.flush_objects_batch(slot, B)          # round 1's: explicit gather
        for k in range(B):
            res[f"prev_{slot}_{k}"] = prev[k]
            res[f"last_{slot}_{k}"] = last[k]
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **res)
    pipe.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("assign", ["block", "round-robin"])
def test_two_ranks_over_rccl_equal_the_single_context(tmp_path, assign):
    from moped_amd import capi, synth
    world = 2
    port = 29900 + os.getpid() % 90
    mp.spawn(_worker, args=(world, port, str(tmp_path), assign), nprocs=world, join=True)
    db = synth.make_db(N_MODELS, PPM)
    frs = _frames(db, synth, B * DEPTH)
    dev = torch.device("cuda:0")
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    c.reserve(Q)
    prm = capi.default_frame_params()

    def single(fr, seed):
        qd, uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
        c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, seed)
        return c.frame_fetch()[0]

    def canon(x):
        key = [(int(o["model"]),) + tuple(o["pose"].view(np.uint32).tolist()) for o in x]
        return x[sorted(range(len(x)), key=lambda i: key[i])]

    got = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for slot in range(DEPTH):
        for k in range(B):
            fr = frs[slot * B + k]
            for rnd, name in ((0, "prev"), (1, "last")):
                want = canon(single(fr, 100 * rnd + slot * B + k + 1))
                assert len(want) >= 3
                for r in range(world):
                    g = canon(got[r][f"{name}_{slot}_{k}"])
                    assert len(g) == len(want) and np.array_equal(g["model"], want["model"]), (name, slot, k, r)
                    assert np.array_equal(g["pose"].view(np.uint32), want["pose"].view(np.uint32))
                    assert np.array_equal(g["score"].view(np.uint32), want["score"].view(np.uint32))
    c.close()


@pytest.mark.skipif(torch.cuda.device_count() < 4, reason="a 2 x 2 grid over RCCL needs four GPUs")
def test_grid_two_by_two_over_rccl_equals_the_single_context(tmp_path):
    """bench.py's grid partition on real devices: four processes = 2 model shards x 2 frame groups, each frame group's
    communicators created from its own shard 0's ids (handed over by all_gather_object on the default group), the two
    groups' collectives in flight at the same time on disjoint communicators; every rank's objects are the
    single-context result of its group's frames."""
    from moped_amd import capi, synth
    world, G = 4, 2
    port = 29800 + os.getpid() % 90
    mp.spawn(_worker, args=(world, port, str(tmp_path), "round-robin", G), nprocs=world, join=True)
    db = synth.make_db(N_MODELS, PPM)
    all_frs = _frames(db, synth, B * DEPTH * (world // G))
    dev = torch.device("cuda:0")
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    c.reserve(Q)
    prm = capi.default_frame_params()

    def canon(x):
        key = [(int(o["model"]),) + tuple(o["pose"].view(np.uint32).tolist()) for o in x]
        return x[sorted(range(len(x)), key=lambda i: key[i])]

    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"r{rank}.npz"))
        frs = all_frs[(rank // G) * B * DEPTH:(rank // G + 1) * B * DEPTH]
        for slot in range(DEPTH):
            for k in range(B):
                fr = frs[slot * B + k]
                for rnd, name in ((0, "prev"), (1, "last")):
                    qd, uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
                    c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, synth.K_DEFAULT, synth.CAM_IDENTITY, prm, 100 * rnd + slot * B + k + 1)
                    want = canon(c.frame_fetch()[0])
                    g = canon(got[f"{name}_{slot}_{k}"])
                    assert len(g) == len(want) >= 3 and np.array_equal(g["model"], want["model"]), (rank, name, slot, k)
                    assert np.array_equal(g["pose"].view(np.uint32), want["pose"].view(np.uint32))
                    assert np.array_equal(g["score"].view(np.uint32), want["score"].view(np.uint32))
    c.close()
