import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from moped_amd import synth, capi
from moped_amd.pipeline import FramePipeline, ShardedDB
B, depth, n, Q = 4, 16, 20480, 3000
db = synth.make_db(20, 5000)
dev = torch.device("cuda:0")
frs = [synth.make_frame(db, n_vis=2, seed=s, Q=Q) for s in range(B)]
qd0 = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev)
uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
def run(label, mod):
    prm = capi.default_frame_params()
    mod(prm)
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=depth, max_queries=Q * B, params=prm)
    work = [torch.empty_like(qd0) for _ in range(depth)]
    def go(groups):
        for g in range(groups):
            slot = g % depth
            with torch.cuda.stream(pipe.streams[slot]):
                work[slot].copy_(qd0, non_blocking=True)
            pipe.enqueue_batch(slot, work[slot], uv, B, list(range(g * B + 1, g * B + B + 1)))
    go(768); pipe.synchronize()
    t0 = time.perf_counter(); go(n // B); pipe.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {n / dt:.0f} frames/s", flush=True)
    pipe.close()
def lm0(p): p.pose1.lm_iters_l2 = p.pose1.lm_iters_l4 = p.pose2.lm_iters_l2 = p.pose2.lm_iters_l4 = 0
def hyp64(p): p.pose1.n_hypotheses = p.pose2.n_hypotheses = 64
def nostage2(p): p.run_stage2 = 0
def rep1(p): p.pose1.max_objects_per_cluster = p.pose2.max_objects_per_cluster = 1
run("full", lambda p: None)
run("no LM", lm0)
run("64 hypotheses", hyp64)
run("1 replica", rep1)
run("no FILTER/POSE2/FILTER2", nostage2)
run("no LM + 64 hypotheses + 1 replica (same launches, little work)", lambda p: (lm0(p), hyp64(p), rep1(p)))
run("all of these", lambda p: (lm0(p), hyp64(p), rep1(p), nostage2(p)))
run("full again", lambda p: None)
