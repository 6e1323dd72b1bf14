"""The N > 1 path behind the C ABI (csrc/comm.hip): mh_comm_* + mh_frame_enqueue_sharded*.  A sharded frame's objects
must be the single-context frame's, bit for bit, whatever the number of ranks and the transport:
  - the C++ host that owns its devices (moped_hip_test --world W; the loop of moped2/libmoped/src/moped.cpp:166-194):
    W = 1 over RCCL (mh_comm_create_all, ncclAllGather with one rank), W = 2, 3, 8 as threads that share the one
    device of this box over the host transport;
  - the same entry points from Python (capi.Comm)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from moped_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "moped_amd", "host")
pytestmark = pytest.mark.gpu
N_MODELS, PPM, Q = 8, 1500, 1200


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_scene
    subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_test"])
    db = synth.make_db(N_MODELS, PPM)
    fr = synth.make_frame(db, n_vis=3, seed=3, Q=Q, pts_per_obj=120)
    path = str(tmp_path_factory.mktemp("comm") / "scene.bin")
    dump_scene.dump(path, db, fr)
    return db, fr, path


@pytest.fixture(scope="module")
def single(scene):
    """The frame in one context holding the whole DB, seed 7 (the harness's)."""
    import torch
    db, fr, _ = scene
    dev = torch.device("cuda:0")
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    c.reserve(Q)
    qd, uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
    c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, synth.K_DEFAULT, synth.CAM_IDENTITY, capi.default_frame_params(), 7)
    objs, counts = c.frame_fetch()
    c.close()
    assert len(objs) >= 3
    return objs, counts


def _run_harness(path, world):
    out = subprocess.check_output([os.path.join(HOST, "moped_hip_test"), "--world", str(world), path, "2"], text=True)
    objs, head, transport = [], None, None
    for line in out.splitlines():
        w = line.split()
        if w[0] == "OBJ":
            objs.append((int(w[1].replace("model", "")), [float(x) for x in w[5:9] + w[2:5]], float(w[9])))
        elif w[0] == "MATCHES":
            head = (int(w[1]), int(w[3]))
        elif w[0] == "TRANSPORT":
            transport = w[1]
    arr = np.zeros(len(objs), capi.OBJECT_DTYPE)
    for i, (m, pose, score) in enumerate(objs):
        arr[i]["model"], arr[i]["pose"], arr[i]["score"] = m, np.array(pose, np.float32), np.float32(score)
    return arr, head, transport


def _same(a, b):
    """The same objects bit for bit; ranks deliver theirs in rank order (= model blocks), one context in FILTER2's
    list order, so both sides are put in (model, pose) order first."""
    def canon(x):
        key = [(int(o["model"]),) + tuple(o["pose"].view(np.uint32).tolist()) for o in x]
        return x[sorted(range(len(x)), key=lambda i: key[i])]
    a, b = canon(a), canon(b)
    return (len(a) == len(b) and np.array_equal(a["model"], b["model"]) and
            np.array_equal(a["pose"].view(np.uint32), b["pose"].view(np.uint32)) and
            np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)))


def test_cxx_host_world1_over_rccl(scene, single):
    objs, head, transport = _run_harness(scene[2], 1)
    assert transport == "rccl"
    assert head == (int(single[1][0]), int(single[1][1]))
    assert _same(objs, single[0])


@pytest.mark.parametrize("world", [2, 3, 8])
def test_cxx_host_ranks_as_threads(scene, single, world):
    objs, head, transport = _run_harness(scene[2], world)
    assert transport == "host"            # one device on this box: more ranks than devices
    assert head[0] == int(single[1][0])   # accepted matches: every one lands on exactly one rank
    assert head[1] == int(single[1][1])
    assert _same(objs, single[0])


def test_python_comm_create_all_and_sharded_all(scene, single):
    import torch
    db, fr, _ = scene
    dev = torch.device("cuda:0")
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    c.reserve(2 * Q)
    comms = capi.Comm.create_all([c])
    assert comms[0].info() == (0, 1, True)
    prm = capi.default_frame_params()
    # a batch of two frames (the same one twice, different seeds) through one MATCH launch and one exchange
    qd = torch.from_numpy(np.concatenate([fr.desc, fr.desc])).to(dev)
    uv = torch.from_numpy(np.concatenate([fr.uv, fr.uv])).to(dev)
    capi.frame_enqueue_sharded_all([c], comms, [qd.data_ptr()], [uv.data_ptr()], Q, 2, synth.K_DEFAULT,
                                   synth.CAM_IDENTITY, prm, [7, 8])
    o0, c0 = c.frame_fetch_slot(0)
    assert _same(o0, single[0]) and np.array_equal(c0, single[1])
    # the next batch carries this one's objects (exchange 2 rides on exchange 1)
    qd2 = torch.from_numpy(np.concatenate([fr.desc, fr.desc])).to(dev)
    capi.frame_enqueue_sharded_all([c], comms, [qd2.data_ptr()], [uv.data_ptr()], Q, 2, synth.K_DEFAULT,
                                   synth.CAM_IDENTITY, prm, [9, 7])
    # ... and the delivery form of the same: both frames of the previous batch in one stream-ordered copy into pinned
    # host memory, no stream synchronisation (mh_frame_fetch_previous_async / mh_frame_fetch_wait)
    cap = 8
    block = torch.zeros(capi.frame_block_bytes(2, cap), dtype=torch.uint8).pin_memory()
    c.frame_fetch_previous_async(cap, block.data_ptr(), tag=0xABCD)
    c.frame_fetch_wait()
    recs = block.numpy().view(capi.frame_block_dtype(cap))
    assert np.all(recs["head"]["tag"] == 0xABCD) and np.array_equal(recs["head"]["frame"], [0, 1])
    for f in (0, 1):
        n = int(recs[f]["head"]["n_objects"])
        assert n == len(c.frame_previous_objects(f)) and recs[f]["head"]["flags"] == 0
        assert _same(recs[f]["objects"][:n], c.frame_previous_objects(f))
    assert _same(c.frame_previous_objects(0), single[0])
    assert _same(c.frame_gather_objects(comms[0], 1), single[0])     # slot 1 of the current batch: seed 7 again
    c.synchronize()
    comms[0].close()
    c.close()


def test_comm_argument_errors(scene):
    c = capi.Context(0)
    with pytest.raises(capi.MhError):
        capi.Comm.create(c, b"\0" * 128, 3, 2)         # rank outside the world
    with pytest.raises(capi.MhError):
        c.frame_previous_objects(0)                     # no sharded frame yet
    c.close()


def _two_ranks_as_threads(db, fr, seeds_by_rank, batch=1):
    """Two shard contexts on this one device, one Python thread each, over the host transport; -> per rank either the
    gathered objects or the MhError the fetch raised."""
    import threading
    import torch
    from moped_amd.pipeline import ShardedDB
    dev = torch.device("cuda:0")
    world = 2
    board, cv = {"n": 0, "round": 0, "blocks": [None] * world}, threading.Condition()

    def allgather_for(rank):
        def fn(blob):
            with cv:
                my_round = board["round"]
                board["blocks"][rank] = blob
                board["n"] += 1
                if board["n"] == world:
                    board["out"] = b"".join(board["blocks"])
                    board["n"] = 0
                    board["round"] += 1
                    cv.notify_all()
                else:
                    cv.wait_for(lambda: board["round"] != my_round, timeout=60)
                return board["out"]
        return fn

    out = [None] * world

    def run(rank):
        try:
            c = capi.Context(0)
            sh = ShardedDB(db.desc, db.xyz, db.model_of, db.n_models, rank, world)
            sh.upload(c, c.normalize(sh.desc))
            c.reserve_batch(Q, batch)
            comm = capi.Comm.create_host(c, rank, world, allgather_for(rank))
            qd = torch.from_numpy(np.concatenate([fr.desc] * batch)).to(dev)
            uv = torch.from_numpy(np.concatenate([fr.uv] * batch)).to(dev)
            prm = capi.default_frame_params()
            c.frame_enqueue_sharded_batch(comm, qd.data_ptr(), uv.data_ptr(), Q, batch, synth.K_DEFAULT, synth.CAM_IDENTITY, prm,
                                          seeds_by_rank[rank])
            try:
                c.frame_fetch_slot(0)
                out[rank] = c.frame_gather_objects(comm, 0)
            except capi.MhError as e:
                out[rank] = e
                try:
                    c.frame_gather_objects(comm, 0)    # (the other rank may be waiting in its gather)
                except capi.MhError:
                    pass
            comm.close()
            c.close()
        except Exception as e:   # pragma: no cover - reported by the assertion below
            out[rank] = e
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    return out


def test_python_two_ranks_as_threads(scene, single):
    db, fr, _ = scene
    out = _two_ranks_as_threads(db, fr, [[7], [7]])
    assert all(isinstance(o, np.ndarray) for o in out), out
    assert _same(out[0], single[0]) and _same(out[1], single[0])


def test_ranks_that_disagree_on_the_frame_fail_loudly(scene):
    """Every block of a frame exchange carries (sequence number on its communicator, frame seed): ranks whose
    collectives pair up differently -- here: rank 1 believes it is exchanging another frame -- get an error at fetch,
    not a silently mixed frame (several communicators in flight are only correct if every rank issues its
    collectives in the same order)."""
    db, fr, _ = scene
    out = _two_ranks_as_threads(db, fr, [[7], [8]])
    assert all(isinstance(o, capi.MhError) for o in out), out
    assert "different orders" in str(out[0])


def test_grid_two_shards_by_two_frame_groups_as_threads(scene):
    """bench.py --parallelism grid on one device: 4 ranks = 2 model shards x 2 frame groups (rank = r G + g), each frame
    group with a communicator of its own over its two ranks (here: the host transport, one board per group) and its own
    frame; the groups run concurrently and never exchange.  Every rank's gathered objects are the single-context result
    of ITS group's frame, bit for bit, with the round-robin model assignment bench.py uses."""
    import threading
    import torch
    from moped_amd.pipeline import ShardedDB
    db, fr_a, _ = scene
    fr_b = synth.make_frame(db, n_vis=2, seed=11, Q=Q, pts_per_obj=120)
    frames = [fr_a, fr_b]
    dev = torch.device("cuda:0")
    G, R = 2, 2
    want = []
    c = capi.Context(0)
    c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
    c.reserve(Q)
    for r, fr in enumerate(frames):
        qd, uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
        c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, synth.K_DEFAULT, synth.CAM_IDENTITY, capi.default_frame_params(), 20 + r)
        want.append(c.frame_fetch()[0])
    c.close()
    assert len(want[0]) >= 3 and len(want[1]) >= 2
    assert sorted(want[0]["model"].tolist()) != sorted(want[1]["model"].tolist())   # the groups see different objects

    boards = [{"n": 0, "round": 0, "blocks": [None] * G, "cv": threading.Condition()} for _ in range(R)]

    def allgather_for(r, g):
        board = boards[r]
        cv = board["cv"]

        def fn(blob):
            with cv:
                my_round = board["round"]
                board["blocks"][g] = blob
                board["n"] += 1
                if board["n"] == G:
                    board["out"] = b"".join(board["blocks"])
                    board["n"] = 0
                    board["round"] += 1
                    cv.notify_all()
                else:
                    cv.wait_for(lambda: board["round"] != my_round, timeout=60)
                return board["out"]
        return fn

    out = [None] * (G * R)

    def run(rank):
        r, g = rank // G, rank % G
        try:
            c = capi.Context(0)
            sh = ShardedDB(db.desc, db.xyz, db.model_of, db.n_models, g, G, assign="round-robin")
            sh.upload(c, c.normalize(sh.desc))
            c.reserve_batch(Q, 2)
            comm = capi.Comm.create_host(c, g, G, allgather_for(r, g))
            assert comm.info()[:2] == (g, G)
            fr = frames[r]
            qd = torch.from_numpy(np.concatenate([fr.desc] * 2)).to(dev)
            uv = torch.from_numpy(np.concatenate([fr.uv] * 2)).to(dev)
            c.frame_enqueue_sharded_batch(comm, qd.data_ptr(), uv.data_ptr(), Q, 2, synth.K_DEFAULT, synth.CAM_IDENTITY,
                                          capi.default_frame_params(), [20 + r, 20 + r])
            c.frame_fetch_slot(0)
            out[rank] = [c.frame_gather_objects(comm, f) for f in (0, 1)]
            comm.close()
            c.close()
        except Exception as e:   # pragma: no cover - reported by the assertion below
            out[rank] = e
    ts = [threading.Thread(target=run, args=(k,)) for k in range(G * R)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    for rank, o in enumerate(out):
        assert isinstance(o, list), (rank, o)
        for f in (0, 1):
            assert _same(o[f], want[rank // G]), (rank, f)
