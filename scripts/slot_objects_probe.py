"""Per-slot object counts of a frame-parallel job that follows a closed sharded one in the same process (debugging aid)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch
from moped_amd import synth
first = sys.argv[1] if len(sys.argv) > 1 else "sharded"
args = b.parse(["--frames-per-step", "64"])
args.depth = 16
torch.cuda.set_device(0)
env = {"rank": 0, "local_rank": 0, "world": 1, "dev": torch.device("cuda:0"), "red_dev": torch.device("cuda:0")}
db = synth.make_db(args.models, 5000)
if first == "sharded":
    a1 = b.parse(["--frames-per-step", "64", "--force-exchange"]); a1.depth = 16
    j1 = b.Job(a1, env, db, args.models, False, True, 32, 64)
    j1.timed(2, 1)
    print("first job (sharded, forced exchange, 32 frames per batch): objects per frame", j1.detections_per_frame(), flush=True)
    j1.close()
elif first == "plain":
    j1 = b.Job(args, env, db, args.models, True, False, 8, 64)
    j1.timed(2, 1)
    print("first job (plain): objects per frame", j1.detections_per_frame(), flush=True)
    j1.close()
B = 8
job = b.Job(args, env, db, args.models, True, False, B, args.frames_per_step)
dt, _ = job.timed(1, 1)
for slot in sorted(job.last_slots):
    objs = [len(r[0]) for r in job.pipe.fetch_batch(slot, B)]
    cnts = [job.pipe.ctxs[slot].frame_fetch_slot(f)[1].tolist() for f in range(B)]
    print("slot", slot, "pool group", job.last_slots[slot], "objects", objs, "counts", cnts[0], cnts[-1])
job.close()
