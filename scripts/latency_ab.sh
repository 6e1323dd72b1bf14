#!/bin/bash
# one frame alone, enqueue -> objects on the host, interleaved A/B of library builds on ONE box:
#   latency_ab.sh libA.so libB.so ...   (paths relative to the repo; MH_LIB_PATH selects the build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$@"; do
    for nv in 2 10; do
      echo -n "$lib: "; MH_LIB_PATH=$PWD/$lib python3 scripts/single_frame_timeline.py run 20 $nv 2>&1 | grep -v amdgpu.ids
    done
  done
done
