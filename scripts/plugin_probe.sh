#!/bin/bash
# plugin_resident / plugin_path of bench.py alone: dump a scene, run moped_hip_test both ways
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import dump_scene
from moped_amd import synth
db = synth.make_db(20, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0)
dump_scene.dump("/tmp/scene.bin", db, fr)
PY
for i in 1 2; do
moped_amd/host/moped_hip_test --resident /tmp/scene.bin 30 | grep -E "^TIME|^OBJ" | tr '\n' ' '; echo
done
moped_amd/host/moped_hip_test /tmp/scene.bin 30 | grep -E "^TIME" | tr '\n' ' '; echo
