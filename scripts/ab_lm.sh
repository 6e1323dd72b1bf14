#!/usr/bin/env bash
# A/B of the LM / sampling variants (libmoped_hip_ab_<name>.so, built with the LM_* / DRAW_MULHI macros of csrc/pose.hip)
# on the scenes of tests/tools/frame_stress.py that a change moved.  usage (GPU box): bash scripts/ab_lm.sh "names" "scenes"
root=$(cd "$(dirname "$0")/.." && pwd)
for n in $1; do
  for sc in $2; do
    echo "== $n scene group of $sc"
    MH_LIB_PATH=$root/moped_amd/libmoped_hip_ab_$n.so FRAME_STRESS_ONLY=$sc timeout -k 10 200 python $root/tests/tools/frame_stress.py 600 2>&1 | grep -v "amdgpu.ids" | grep -v "scenes, 0 mismatches, " | tail -4
  done
done
