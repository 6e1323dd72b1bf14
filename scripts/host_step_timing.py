import os, sys, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from moped_amd import synth
import dump_scene
HOST = os.path.join(ROOT, "moped_amd", "host")
subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_test"])
db = synth.make_db(20, 5000)
fr = synth.make_frame(db, n_vis=2, seed=5, Q=3000)
dump_scene.dump("/tmp/scene.bin", db, fr)
print(subprocess.check_output([os.path.join(HOST, "moped_hip_test"), "/tmp/scene.bin", "10"], text=True))
