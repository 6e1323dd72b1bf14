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


def test_default_partition_is_the_north_stars():
    b = _bench()
    # BASELINE.json north_star: "partition across the 8 GPUs of one node by sharding the model database" -- whatever the
    # DB size; the frame-parallel figure rides along as `replicated_frames`
    for n_models, world in ((20, 8), (200, 8), (200, 2), (50, 4), (20, 1), (20, 2)):
        assert b.choose_parallelism(n_models, world) == "models"


def test_scaling_label():
    b = _bench()
    assert b.scaling_label("models", 1) == "n/a" and b.scaling_label("frames", 1) == "n/a"   # nothing scales at N = 1
    assert b.scaling_label("models", 8) == "strong" and b.scaling_label("frames", 8) == "weak"


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
    a = b.parse(["--gpus", "8", "--models", "200"])
    assert b.default_batch(a, True) == 16                  # configs[3]: 25 models per rank, MATCH fills the chip
    a = b.parse([])
    assert b.default_batch(a, False) == 16
    a = b.parse(["--depth-kind", "1"])
    assert b.default_batch(a, False) == 16                 # depth attributes travel with a merged batch (round 4)
    a = b.parse(["--depth-kind", "1", "--moped3d-frontend"])
    assert b.default_batch(a, False) == 16                 # a depth map per frame travels with a merged batch too (round 4)
