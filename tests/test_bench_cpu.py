"""bench.py's host-side choices (no GPU)."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_auto_partition_keeps_model_sharding_and_takes_the_grid_the_table_projects():
    b = _bench()
    # BASELINE.json north_star: "partition across the 8 GPUs of one node by sharding the model database" -- the DB stays
    # sharded (G >= 2) for every N > 1; a DB too small for N shards gets frame groups beside them
    assert b.choose_partition(20, 1) == (1, 1)
    assert b.choose_partition(20, 2) == (2, 1)
    assert b.choose_partition(20, 4) == (2, 2)
    assert b.choose_partition(20, 8) == (2, 4)
    assert b.choose_partition(200, 8) == (8, 1) and b.choose_partition(200, 4) == (4, 1) and b.choose_partition(200, 2) == (2, 1)
    for n_models, world in ((20, 8), (200, 8), (50, 4), (20, 2), (3, 8)):
        G, R = b.choose_partition(n_models, world)
        assert G >= 2 and G * R == world
    # the projections the choice rests on (profiles/per_rank_load.json): north_star's >= 6x at 8 GPUs for both DBs
    assert b.projected_speedup(20, 2, 4) >= 6.0 and b.projected_speedup(200, 8, 1) >= 6.0
    assert b.projected_speedup(20, 8, 1) < 3.5          # why pure sharding of the small DB is not what auto runs
    # the explicit forms
    assert b.choose_partition(20, 8, "models") == (8, 1) and b.choose_partition(20, 8, "frames") == (1, 8)
    assert b.choose_partition(20, 8, "grid", "4x2") == (4, 2)
    import pytest
    with pytest.raises(SystemExit):
        b.choose_partition(20, 8, "grid", "3x2")


def test_per_rank_table_interpolates():
    b = _bench()
    t = {"10": 20000.0, "20": 10000.0}
    assert abs(b.per_rank_us(10, t) - 50.0) < 1e-9 and abs(b.per_rank_us(20, t) - 100.0) < 1e-9
    assert abs(b.per_rank_us(15, t) - 75.0) < 1e-9
    assert abs(b.per_rank_us(5, t) - 50.0) < 1e-9       # below the table: the smallest shard's cost (it does not shrink further)
    assert abs(b.per_rank_us(40, t) - 200.0) < 1e-9     # beyond: the last slope


def test_scaling_and_partition_labels():
    b = _bench()
    assert b.scaling_label(1, 1) == "n/a"                                       # nothing scales at N = 1
    assert b.scaling_label(8, 1) == "strong" and b.scaling_label(1, 8) == "weak" and b.scaling_label(2, 4) == "weak"
    assert b.partition_label(1, 1) == "single GPU" and b.partition_label(8, 1) == "model-shard x8"
    assert b.partition_label(1, 4).startswith("frame-parallel x4") and b.partition_label(2, 4) == "grid: 2 model shards x 4 frame groups"


def test_gpus_n_without_a_launcher_never_prints_a_line_for_ranks_that_did_not_run():
    """`python bench.py --gpus 2` on a host with fewer than 2 devices (this container has none): non-zero exit, no JSON
    line -- never `"n_gpus": 2` from one process."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MH_BENCH_REHEARSE")}
    import torch
    if torch.cuda.device_count() >= 2:
        return   # (a multi-GPU host really runs it: covered by tests/test_gpu_bench_rehearsal.py)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 2" in r.stderr


def test_default_batch_follows_the_partition():
    b = _bench()
    from moped_amd import capi
    a = b.parse(["--gpus", "8"])
    assert b.default_batch(a, True) == capi.MAX_BATCH      # 2.5 models per rank: as many frames per launch as the library takes
    assert b.default_batch(a, True, 2) == capi.MAX_BATCH   # a 2 x 4 grid: 10 models per rank
    a200 = b.parse(["--gpus", "8", "--models", "200"])
    assert b.default_batch(a200, True, 2) == 16            # 100 models per rank
    a = b.parse(["--gpus", "8", "--models", "200"])
    assert b.default_batch(a, True) == 16                  # configs[3]: 25 models per rank, MATCH fills the chip
    a = b.parse([])
    assert b.default_batch(a, False) == 16
    a = b.parse(["--depth-kind", "1"])
    assert b.default_batch(a, False) == 16                 # depth attributes travel with a merged batch (round 4)
    a = b.parse(["--depth-kind", "1", "--moped3d-frontend"])
    assert b.default_batch(a, False) == 16                 # a depth map per frame travels with a merged batch too (round 4)
