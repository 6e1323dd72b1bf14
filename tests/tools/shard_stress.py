"""Randomised check of the N > 1 path: the C++ host (moped_hip_test --world W: W ranks as threads on this one device over
the host transport; mh_comm_* + mh_frame_enqueue_sharded) against the single-context frame on random databases, frames
and rank counts -- the objects must be bit-identical whatever the number of shards (models per rank down to one, ranks
without any visible model, ranks without models at all).  usage: shard_stress.py [scenes=30] [seed=0]"""
import os, sys, subprocess, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import dump_scene
from moped_amd import capi, synth
import test_gpu_comm as T
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
subprocess.check_call(["make", "-s", "-C", T.HOST, "moped_hip_test"])
dev = torch.device("cuda:0")
bad = 0
t0 = time.perf_counter()
with tempfile.TemporaryDirectory() as tmp:
    for sc in range(scenes):
        n_models, ppm = int(rng.choice([2, 5, 8, 13, 20])), int(rng.choice([600, 1500]))
        Q = int(rng.choice([600, 1200, 3000]))
        n_vis = int(rng.integers(0, min(n_models, 6) + 1))
        db = synth.make_db(n_models, ppm, seed=int(rng.integers(1 << 30)))
        fr = synth.make_frame(db, n_vis=n_vis, seed=int(rng.integers(1 << 30)), Q=Q, pts_per_obj=int(rng.choice([40, 120])),
                              outlier_frac=float(rng.choice([0.0, 0.3])))
        Q = len(fr.desc)   # (planted points beyond the requested count make the frame longer)
        world = int(rng.choice([2, 3, 5, 8]))
        path = os.path.join(tmp, "scene.bin")
        dump_scene.dump(path, db, fr)
        c = capi.Context(0)
        c.db_upload(c.normalize(db.desc), db.model_of, db.xyz, db.n_models)
        c.reserve(Q)
        qd, uv = torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev)
        c.frame_enqueue(qd.data_ptr(), uv.data_ptr(), Q, synth.K_DEFAULT, synth.CAM_IDENTITY, capi.default_frame_params(), 7)
        want, counts = c.frame_fetch()
        c.close()
        got, head, transport = T._run_harness(path, world)
        ok = T._same(got, want) and head == (int(counts[0]), int(counts[1]))
        if not ok:
            bad += 1
            print(f"MISMATCH scene {sc}: {n_models} models x {ppm}, Q={Q}, n_vis={n_vis}, world {world} ({transport}): "
                  f"{len(got)} objects {head} vs {len(want)} {counts[:2].tolist()}", flush=True)
        if sc % 10 == 9: print(f"{sc + 1} scenes, {bad} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{scenes} sharded scenes (world 2-8 as threads over the host transport), {bad} mismatches")
sys.exit(1 if bad else 0)
