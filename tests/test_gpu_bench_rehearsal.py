"""bench.py's N > 1 code paths rehearsed on one GPU (MH_BENCH_REHEARSE=1: both ranks on cuda:0, gloo for the timing
contract, the library's host transport for a sharded DB's exchange): plain `python bench.py --gpus 2` -- no launcher in
the test -- starts its two ranks itself, the line the driver will read comes out, names the partition that ran (the
north star's model sharding by default) and the ranks the exchange saw, and finds the planted objects.  The numbers of
such a run mean nothing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, launcher=False, port=0, gpus=2, roofline=False):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MH_BENCH_REHEARSE"] = "1"
    # --watchdog: a rank that waits for a collective the others never issue ends with its stack, not with the suite's timeout
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
            "--h2d-steps", "0", "--watchdog", "300"] + ([] if roofline else ["--no-roofline"]) + extra
    if launcher:   # the way the driver starts N > 1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 prints ONE line
    return json.loads(lines[0])


def test_plain_python_gpus_2_starts_two_ranks_and_shards_the_models():
    d = _run(["--frames-per-step", "64", "--secondary-steps", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "model-shard x2"
    assert d["config"]["ranks_launched_by"] == "bench.py"
    ex = d["config"]["exchange"]
    assert ex["world"] == 2 and ex["rank"] == 0        # the transport of the frames' exchange saw two ranks
    assert d["config"]["objects_per_frame"] == 2.0 and d["value"] > 0 and d["steps"] == 2 and d["warmup"] == 1
    assert "suspect" not in d
    # every timed frame's objects (all ranks') reached the host inside the timed region
    det = d["config"]["objects_detail"]
    assert d["config"]["results_delivered"] == "every frame" and det["frames"] == 2 * d["config"]["frames_per_step"]
    assert det["frames_missing_a_planted_object"] == 0 and det["objects_per_frame_histogram"] == [0, 0, det["frames"]]
    assert d["config"]["env_overrides"].get("MH_BENCH_REHEARSE") == "1"
    # the other partition and the other workload ride in the same line
    assert d["replicated_frames"]["parallelism"].startswith("frame-parallel x2") and d["replicated_frames"]["value"] > 0
    assert d["replicated_frames"]["objects_per_frame"] == 2.0
    s200 = d["sharded_200_models"]
    assert s200["parallelism"] == "model-shard x2" and s200["models_per_rank"] == 100 and s200["objects_per_frame"] == 2.0


def test_under_the_drivers_launcher_block_models():
    """(the default assignment is round-robin: the first test; here the contiguous blocks, under the driver's launcher)"""
    d = _run(["--models", "50", "--assign", "block", "--frames-per-step", "32", "--no-secondary"], launcher=True,
             port=29741 + os.getpid() % 100)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "model-shard x2"
    assert d["config"]["model_assignment"] == "block" and d["config"]["ranks_launched_by"] == "torch.distributed.run"
    assert d["config"]["objects_per_frame"] == 2.0 and d["config"]["frames_per_match_launch"] == 16


def test_frames_partition_is_an_option():
    d = _run(["--parallelism", "frames", "--frames-per-step", "64", "--no-secondary"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"].startswith("frame-parallel x2")
    assert d["config"]["objects_per_frame"] == 2.0
    assert d["config"]["objects_detail"]["frames"] == 2 * d["config"]["frames_per_step"] and "suspect" not in d


def test_four_ranks_with_every_leg_of_the_default_run():
    """The line exactly as the driver asks for it -- roofline, POSE figures, secondary partitions -- with four ranks: every
    leg that only rank 0 runs must be free of collectives (round 3: the POSE measurement enqueued a sharded batch on rank
    0 alone and the job hung in its all-gather; the two-rank tests had run with --no-roofline).  Round 5: `auto` takes
    the models x frames grid the committed per-rank loads project fastest -- 2 model shards x 2 frame groups for the
    20-model DB at N = 4, each frame group exchanging among its own two ranks -- and the pure sharding of the north
    star rides along as `pure_model_shard`."""
    d = _run(["--frames-per-step", "128", "--secondary-steps", "1"], gpus=4, roofline=True)
    assert d["n_gpus"] == 4 and d["scaling"] == "weak" and d["config"]["parallelism"] == "grid: 2 model shards x 2 frame groups"
    part = d["config"]["partition"]
    assert part["model_shards"] == 2 and part["frame_groups"] == 2 and part["models_per_rank"] == 10
    assert part["projected_speedup"] >= 3.0
    assert d["config"]["model_assignment"] == "round-robin" and d["config"]["exchange"]["world"] == 2   # a frame group's ranks
    assert d["config"]["objects_per_frame"] == 2.0 and "suspect" not in d
    # both frame groups' frames were delivered and counted: 2 groups x 2 steps x frames_per_step
    det = d["config"]["objects_detail"]
    assert det["frames_missing_a_planted_object"] == 0
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["frac"] > 0
    pm = d["pure_model_shard"]
    assert pm["parallelism"] == "model-shard x4" and pm["models_per_rank"] == 5 and pm["objects_per_frame"] == 2.0
    assert d["replicated_frames"]["objects_per_frame"] == 2.0 and d["sharded_200_models"]["models_per_rank"] == 50


def test_four_ranks_pure_model_sharding_is_still_an_option():
    d = _run(["--parallelism", "models", "--frames-per-step", "64", "--no-secondary"], gpus=4)
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["config"]["parallelism"] == "model-shard x4"
    assert d["config"]["exchange"]["world"] == 4 and d["config"]["objects_per_frame"] == 2.0 and "suspect" not in d


def test_explicit_grid_under_the_drivers_launcher():
    d = _run(["--parallelism", "grid", "--grid", "2x2", "--models", "50", "--frames-per-step", "32", "--no-secondary"], gpus=4,
             launcher=True, port=29871 + os.getpid() % 100)
    assert d["n_gpus"] == 4 and d["config"]["parallelism"] == "grid: 2 model shards x 2 frame groups"
    assert d["config"]["partition"]["models_per_rank"] == 25 and d["config"]["exchange"]["world"] == 2
    assert d["config"]["objects_per_frame"] == 2.0 and "suspect" not in d
