"""Does the C++ streaming host run slower as the child of a process that holds a GPU context?  (debugging aid)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from moped_amd import synth
import dump_scene
db = synth.make_db(20, 5000)
frames = [synth.make_frame(db, n_vis=2, seed=s, Q=3000) for s in range(32)]
dump_scene.dump_frames("/tmp/frames.bin", db, frames)
def run(tag):
    out = subprocess.check_output([os.path.join(ROOT, "moped_amd/host/moped_hip_bench"), "/tmp/frames.bin", "--json", "--steps", "5", "--frames-per-step", "1024", "--batch", "16"], text=True)
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print(tag, "affinity", len(os.sched_getaffinity(0)), "resident", d["fps_resident"], "pinned", d["fps_pinned_host"], flush=True)
run("before torch:")
import torch
run("torch imported:")
torch.cuda.init(); x = torch.zeros(1 << 20, device="cuda:0"); torch.cuda.synchronize()
run("GPU context held by the parent:")
y = torch.zeros(1 << 28).pin_memory()
run("+ 1 GB pinned in the parent:")
del x, y; torch.cuda.empty_cache()
run("released:")
