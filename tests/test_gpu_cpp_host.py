"""The C++ streaming host (moped_amd/host/moped_hip_bench.cpp): no Python and no torch between the frames and the C ABI.
A C++ host that keeps frames in flight -- 16 contexts sharing one model database, batches through
mh_frame_enqueue_batch, descriptors resident or crossing PCIe from pinned memory -- must find what the Python
pipeline finds on the same scene, and deliver the throughput bench.py's h2d-inclusive figure claims for the C ABI
(within 5% of it; checked in bench.py's own line as `cpp_host`, here at a size the suite affords)."""
import json
import os
import subprocess
import sys

import pytest

from moped_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "moped_amd", "host")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames_file(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_scene
    subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_bench"])
    db = synth.make_db(20, 5000)
    frames = [synth.make_frame(db, n_vis=2, seed=s, Q=3000) for s in range(32)]
    path = str(tmp_path_factory.mktemp("cpp") / "frames.bin")
    dump_scene.dump_frames(path, db, frames)
    return path


def _run(path, *args):
    out = subprocess.check_output([os.path.join(HOST, "moped_hip_bench"), path, "--json", *args], text=True, timeout=300)
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_streaming_host_finds_every_object_and_streams(frames_file):
    d = _run(frames_file, "--steps", "3", "--frames-per-step", "512")
    assert d["slots"] == 16 and d["frames_per_batch"] == 8 and d["queries"] == 3000 and d["rows"] == 100000
    assert d["objects_per_frame"] == 2.0                  # both visible objects of every frame, in every slot's last batch
    # config 1 runs at ~13 000 frames/s from Python; a C++ host must not be far below, resident or from pinned memory
    assert d["fps_resident"] > 8000 and d["fps_pinned_host"] > 2500
    # (alone on the chip the pinned-host rate is within 10% of the resident one -- 12 400 against 13 000 frames/s --, and
    # bench.py measures it that way, before it touches the GPU itself; as a child of this suite's process, whose earlier
    # tests left streams and queues behind, it has been seen between 5 000 and 8 700)
    assert 0.3 < d["single_frame_latency_ms"] < 3.0


def test_streaming_host_other_shapes(frames_file):
    for args in (("--slots", "4", "--batch", "1"), ("--slots", "2", "--batch", "32")):
        d = _run(frames_file, "--steps", "2", "--frames-per-step", "128", *args)
        assert d["objects_per_frame"] == 2.0 and d["fps_resident"] > 0
