#!/bin/bash
# dumps a scene of 32 frames and runs the C++ streaming host and the STEP-plugin harness on it (what bench.py's host_side_figures does)
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
from moped_amd import synth
import dump_scene
db = synth.make_db(20, 5000)
frames = [synth.make_frame(db, n_vis=2, seed=s, Q=3000) for s in range(32)]
dump_scene.dump_frames("/tmp/frames.bin", db, frames)
dump_scene.dump("/tmp/scene.bin", db, frames[0])
print("dumped")
PY
moped_amd/host/moped_hip_bench /tmp/frames.bin --json --steps 5; echo "rc=$?"
moped_amd/host/moped_hip_bench /tmp/frames.bin --steps 5 --slots 16 --batch 16; echo "rc=$?"
