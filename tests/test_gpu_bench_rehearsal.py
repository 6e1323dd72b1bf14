"""bench.py's N > 1 code paths rehearsed on one GPU (MH_BENCH_REHEARSE=1: both ranks on cuda:0, gloo for the timing
contract, the library's host transport for a sharded DB's exchange): the line the driver will read comes out, names the
partition that ran and finds the planted objects.  The numbers of such a run mean nothing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, port):
    env = dict(os.environ, MH_BENCH_REHEARSE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-roofline", "--h2d-steps", "0"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_small_db_split_the_frames():
    d = _run(["--frames-per-step", "64"], 29621 + os.getpid() % 100)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"].startswith("frame-parallel x2")
    assert d["config"]["objects_per_frame"] == 2.0 and d["value"] > 0 and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["env_overrides"].get("MH_BENCH_REHEARSE") == "1"


def test_two_ranks_large_db_shard_the_models():
    d = _run(["--models", "50", "--parallelism", "models", "--frames-per-step", "32"], 29741 + os.getpid() % 100)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "model-shard x2"
    assert d["config"]["objects_per_frame"] == 2.0 and d["config"]["frames_per_match_launch"] == 8
