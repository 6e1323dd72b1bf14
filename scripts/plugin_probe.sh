#!/bin/bash
# plugin_resident / plugin_path of bench.py alone: dump a scene, run moped_hip_test both ways
# (round 5: + the six steps with the device-side hand-over off, MH_STEP_HANDOVER=0 = every step uploads its inputs)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
python - <<'PY'
import sys, os
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import dump_scene
from moped_amd import synth
db = synth.make_db(20, 5000)
for nv in (2, 10):
    fr = synth.make_frame(db, n_vis=nv, seed=0)
    dump_scene.dump("/tmp/scene%d.bin" % nv, db, fr)
PY
for nv in 2 10; do
echo "== $nv visible objects"
for i in 1 2; do
echo -n "resident: "; moped_amd/host/moped_hip_test --resident /tmp/scene$nv.bin 60 | grep -E "^TIME" | tr '\n' ' '; echo
done
for i in 1 2; do
echo -n "six steps, hand-over: "; moped_amd/host/moped_hip_test /tmp/scene$nv.bin 60 | grep -E "^TIME|^HANDOVER" | tr '\n' ' '; echo
done
echo -n "six steps, upload paths: "; MH_STEP_HANDOVER=0 moped_amd/host/moped_hip_test /tmp/scene$nv.bin 60 | grep -E "^TIME|^HANDOVER" | tr '\n' ' '; echo
done
