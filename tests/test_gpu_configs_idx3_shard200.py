"""BASELINE configs[3]'s per-rank load at its own size: the 200-model database (1 M descriptors)
sharded 25 models per rank over 8 ranks.  A 1-GPU box cannot hold 8 RCCL ranks, so every rank r is
run here in turn exactly as it would run on its own GPU -- its 25-model shard (index_base = first
global row != 0 for r > 0, global model ids), `mh_frame_enqueue_match_local`, the all-gather through
the real `nccl` backend (world 1: it moves this rank's block into slot r of the exchange buffer; the
other seven slots hold what the other ranks' GPUs would have sent, computed the same way), then
`mh_frame_enqueue_rest_strided` over the 8 blocks.  What must hold:

  * every rank's merged global top-2 (group_kernel's shard merge) selects exactly the matches of
    its own models out of the single-context 200-model search,
  * the objects of rank r are bit-identical (model, pose, score) to the objects the single-context
    200-model frame reports for rank r's models, and the union over the 8 ranks is that frame's list.
Only "RCCL with more than one rank" is left to hardware with N > 1 GPUs."""
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

N_MODELS, PPM, Q, WORLD, N_VIS = 200, 5000, 3000, 8, 10


def _pick_seed():
    """A frame seed whose planted models fall on ranks 0, 3 and 7 (and others): make_frame's first
    draw is the visible set."""
    for seed in range(1000):
        vis = np.random.default_rng([0xF4A3E, seed]).choice(N_MODELS, size=N_VIS, replace=False)
        owners = set(int(m) * WORLD // N_MODELS for m in vis)
        if {0, 3, 7} <= owners and len(owners) >= 6:
            return seed
    raise AssertionError("no seed")


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from moped_amd import capi, synth
    from moped_amd.pipeline import FramePipeline, ShardedDB
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    dev = torch.device("cuda:0")
    db = synth.make_db(N_MODELS, PPM)
    fr = synth.make_frame(db, n_vis=N_VIS, seed=_pick_seed(), Q=Q)
    K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY
    prm = capi.default_frame_params()
    res = {"visible": fr.visible}
    q_uv = torch.from_numpy(fr.uv).to(dev)

    # the single-context 200-model frame
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=Q, params=prm)
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), q_uv, seed=31)
    res["single_objs"], res["single_counts"] = pipe.fetch(0)
    res["single_mq"], res["single_mm"] = pipe.ctxs[0].frame_fetch_matches()
    pipe.close()

    # every rank's context: its shard behind the exchange path
    stride = 3 * Q
    ctxs, locals_ = [], []
    stream = torch.cuda.Stream(device=dev)
    for r in range(WORLD):
        sh = ShardedDB(db.desc, db.xyz, db.model_of, db.n_models, r, WORLD)
        assert sh.row_lo == r * 25 * PPM and (r == 0 or sh.row_lo != 0)
        c = capi.Context(0)
        c.set_stream(stream.cuda_stream)
        c.db_upload(c.normalize(sh.desc), sh.model_of, sh.xyz, sh.n_models, index_base=sh.row_lo)
        c.reserve(Q)
        ctxs.append(c)
    gathered_all = torch.zeros(WORLD * stride, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        # what every rank's GPU would send: its local top-2 block
        for r in range(WORLD):
            local = torch.zeros(stride, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()   # torch's fills and copies before the context's own stream touches the buffers
            ctxs[r].frame_enqueue_match_local(torch.from_numpy(fr.desc).to(dev).data_ptr(), Q, local.data_ptr())
            stream.synchronize()
            locals_.append(local)
            gathered_all[r * stride:(r + 1) * stride] = local
        stream.synchronize()
        for r in range(WORLD):
            # rank r's receive buffer: the other ranks' blocks as they would arrive, its own through RCCL
            gathered = gathered_all.clone()
            gathered[r * stride:(r + 1) * stride] = 0
            qd = torch.from_numpy(fr.desc).to(dev)
            local = torch.zeros(stride, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ctxs[r].frame_enqueue_match_local(qd.data_ptr(), Q, local.data_ptr())
            dist.all_gather_into_tensor(gathered[r * stride:(r + 1) * stride], local)
            ctxs[r].frame_enqueue_rest_strided(q_uv.data_ptr(), Q, gathered.data_ptr(), WORLD, stride, K, CAM0, prm, 31)
            objs, counts = ctxs[r].frame_fetch()
            mq, mm = ctxs[r].frame_fetch_matches()
            res[f"objs{r}"], res[f"counts{r}"], res[f"mq{r}"], res[f"mm{r}"] = objs, counts, mq, mm
    for c in ctxs:
        c.close()
    np.savez(os.path.join(out_dir, "shards.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_config3_every_rank_of_eight_equals_its_part_of_the_single_context_frame(tmp_path):
    port = 30100 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    z = np.load(os.path.join(str(tmp_path), "shards.npz"))
    single, counts = z["single_objs"], z["single_counts"]
    assert set(single["model"].tolist()) == set(z["visible"].tolist())
    owner = lambda m: int(m) * WORLD // N_MODELS
    seen = []
    n_matches = n_clusters = 0
    for r in range(WORLD):
        objs = z[f"objs{r}"]
        want = single[[owner(m) == r for m in single["model"]]]
        # bit-identical objects, in the single-context list's order (model order)
        assert np.array_equal(objs["model"], want["model"])
        assert np.array_equal(objs["pose"].view(np.uint32), want["pose"].view(np.uint32))
        assert np.array_equal(objs["score"].view(np.uint32), want["score"].view(np.uint32))
        # its match lists = the single-context lists of its models (global 2-NN + ratio test after the merge)
        keep = np.array([owner(m) == r for m in z["single_mm"]], bool)
        assert np.array_equal(z[f"mq{r}"], z["single_mq"][keep]) and np.array_equal(z[f"mm{r}"], z["single_mm"][keep])
        n_matches += z[f"counts{r}"][0]
        n_clusters += z[f"counts{r}"][1]
        seen += objs["model"].tolist()
    assert n_matches == counts[0] and n_clusters == counts[1]
    assert sorted(seen) == sorted(single["model"].tolist())
    for r in (0, 3, 7):
        assert len(z[f"objs{r}"]) >= 1          # the ranks the seed was picked for do have work
