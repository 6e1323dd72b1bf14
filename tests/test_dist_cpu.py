"""N > 1 path on CPU: two gloo ranks shard the model DB, run the oracle matcher on
their shard, do exchange 1 with the same helper the GPU pipeline uses, merge, and
keep the matches of the models they own.  The union must equal the single-rank result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, out_dir, n_models, assign):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orclib
    from moped_amd import synth
    from moped_amd.pipeline import ShardedDB, exchange_top2, owner_of_model
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base, _, _ = synth.load_sift_fixture()
    db = synth.make_db(n_models, 300, seed=11)
    dbn = orclib.normalize(db.desc)
    qn = orclib.normalize(base[:500])
    sh = ShardedDB(dbn, db.xyz, db.model_of, db.n_models, rank, world, assign=assign)
    idx, d1, d2 = orclib.match_2nn(sh.desc, qn)
    # local row -> global row: + index_base for one block (mh_db_upload), the block tables for round-robin (mh_db_upload_blocks)
    idx = np.where(idx >= 0, sh.rows[np.maximum(idx, 0)] if len(sh.rows) else -1, -1).astype(np.int32)
    if assign == "block" and len(sh.rows):
        assert np.array_equal(sh.rows, np.arange(sh.row_lo, sh.row_hi))
    local = torch.from_numpy(np.stack([idx, d1.view(np.int32), d2.view(np.int32)]))
    g = exchange_top2(local, world).numpy()          # [W][3][Q], the layout mh_frame_enqueue_rest takes
    gi, g1, g2 = orclib.match_merge(np.ascontiguousarray(g[:, 0]), np.ascontiguousarray(g[:, 1]).view(np.float32),
                                    np.ascontiguousarray(g[:, 2]).view(np.float32))
    # every rank holds the same merged top-2; it keeps the matches of the models it owns
    acc = (gi >= 0) & (g1 / g2 < np.float32(0.8))
    mine = np.array([acc[q] and owner_of_model(int(db.model_of[gi[q]]), db.n_models, world, assign) == rank
                     for q in range(len(gi))])
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), gi=gi, g1=g1, g2=g2, mine=mine)
    dist.barrier()
    dist.destroy_process_group()


# (8, 16): BASELINE configs[3]'s world size; (8, 5): more ranks than models, three ranks own nothing
@pytest.mark.parametrize("world,n_models,assign", [(2, 5, "block"), (3, 5, "block"), (8, 16, "block"), (8, 5, "block"),
                                                   (2, 5, "round-robin"), (3, 7, "round-robin"), (8, 20, "round-robin")])
def test_model_sharded_match_equals_single_rank(tmp_path, world, n_models, assign):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orclib
    from moped_amd import synth
    port = 29500 + (os.getpid() % 2000) + world + n_models
    mp.spawn(_worker, args=(world, port, str(tmp_path), n_models, assign), nprocs=world, join=True)
    base, _, _ = synth.load_sift_fixture()
    db = synth.make_db(n_models, 300, seed=11)
    dbn = orclib.normalize(db.desc)
    qn = orclib.normalize(base[:500])
    oi, o1, o2 = orclib.match_2nn(dbn, qn)
    acc = (oi >= 0) & (o1 / o2 < np.float32(0.8))
    owned = np.zeros(len(oi), int)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.array_equal(z["gi"], oi) and np.array_equal(z["g1"], o1) and np.array_equal(z["g2"], o2)
        owned += z["mine"].astype(int)
    assert np.array_equal(owned, acc.astype(int))   # every accepted match kept by exactly one rank


@pytest.mark.parametrize("assign", ["block", "round-robin"])
@pytest.mark.parametrize("world,n_models", [(2, 5), (3, 7), (8, 20), (8, 5)])
def test_shards_partition_the_rows(world, n_models, assign):
    """Every row belongs to exactly one rank; a shard is runs of consecutive global rows in ascending order (what
    mh_db_upload_blocks asks for); the block assignment is one run."""
    from moped_amd import synth
    from moped_amd.pipeline import ShardedDB, models_of_rank, owner_of_model
    db = synth.make_db(n_models, 40, seed=3)
    seen = np.zeros(db.n, int)
    for r in range(world):
        sh = ShardedDB(db.desc, db.xyz, db.model_of, db.n_models, r, world, assign=assign)
        seen[sh.rows] += 1
        assert np.array_equal(sh.desc, db.desc[sh.rows]) and np.array_equal(sh.model_of, db.model_of[sh.rows])
        assert int(sh.block_rows.sum()) == len(sh.rows)
        ends = sh.block_global_row + sh.block_rows
        assert np.all(sh.block_global_row[1:] > ends[:-1]) if len(ends) > 1 else True   # maximal runs, ascending
        assert np.array_equal(np.concatenate([np.arange(g, g + n) for g, n in zip(sh.block_global_row, sh.block_rows)])
                              if len(sh.rows) else np.zeros(0, int), sh.rows)
        if assign == "block":
            assert len(sh.block_rows) <= 1
        for m in models_of_rank(n_models, r, world, assign):
            assert owner_of_model(int(m), n_models, world, assign) == r
    assert np.all(seen == 1)


def _grid_worker(rank, world, port, out_dir, n_models, G):
    """bench.py --parallelism grid on gloo: rank = r G + g, frame group r = the G ranks r G .. r G + G - 1 with a
    subgroup of their own; group r matches ITS queries against its shards and exchanges inside the subgroup only."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orclib
    from moped_amd import synth
    from moped_amd.pipeline import ShardedDB, exchange_top2, owner_of_model
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R = world // G
    groups = [dist.new_group(list(range(r * G, (r + 1) * G)), backend="gloo") for r in range(R)]   # every rank makes every group
    r, g = rank // G, rank % G
    base, _, _ = synth.load_sift_fixture()
    db = synth.make_db(n_models, 300, seed=11)
    dbn = orclib.normalize(db.desc)
    qn = orclib.normalize(base[200 * r:200 * r + 300])          # the group's own frame
    sh = ShardedDB(dbn, db.xyz, db.model_of, db.n_models, g, G, assign="round-robin")
    idx, d1, d2 = orclib.match_2nn(sh.desc, qn)
    idx = np.where(idx >= 0, sh.rows[np.maximum(idx, 0)] if len(sh.rows) else -1, -1).astype(np.int32)
    local = torch.from_numpy(np.stack([idx, d1.view(np.int32), d2.view(np.int32)]))
    gth = exchange_top2(local, G, group=groups[r]).numpy()       # [G][3][Q]: only the group's ranks take part
    gi, g1, g2 = orclib.match_merge(np.ascontiguousarray(gth[:, 0]), np.ascontiguousarray(gth[:, 1]).view(np.float32),
                                    np.ascontiguousarray(gth[:, 2]).view(np.float32))
    acc = (gi >= 0) & (g1 / g2 < np.float32(0.8))
    mine = np.array([acc[q] and owner_of_model(int(db.model_of[gi[q]]), db.n_models, G, "round-robin") == g
                     for q in range(len(gi))])
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), gi=gi, g1=g1, g2=g2, mine=mine)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,G", [(8, 2), (8, 4), (4, 2)])
def test_grid_of_model_shards_and_frame_groups_equals_single_rank(tmp_path, world, G):
    """2 x 4 (what bench.py's auto takes for the 20-model DB at N = 8), 4 x 2 and 2 x 2: in every frame group the
    merged top-2 is the single rank's on the group's queries and every accepted match is kept by exactly one of the
    group's ranks; the groups never see each other's queries."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orclib
    from moped_amd import synth
    n_models = 20
    port = 31500 + (os.getpid() % 2000) + world + G
    mp.spawn(_grid_worker, args=(world, port, str(tmp_path), n_models, G), nprocs=world, join=True)
    base, _, _ = synth.load_sift_fixture()
    db = synth.make_db(n_models, 300, seed=11)
    dbn = orclib.normalize(db.desc)
    for r in range(world // G):
        qn = orclib.normalize(base[200 * r:200 * r + 300])
        oi, o1, o2 = orclib.match_2nn(dbn, qn)
        acc = (oi >= 0) & (o1 / o2 < np.float32(0.8))
        owned = np.zeros(len(oi), int)
        for g in range(G):
            z = np.load(os.path.join(str(tmp_path), f"r{r * G + g}.npz"))
            assert np.array_equal(z["gi"], oi) and np.array_equal(z["g1"], o1) and np.array_equal(z["g2"], o2)
            owned += z["mine"].astype(int)
        assert np.array_equal(owned, acc.astype(int))
