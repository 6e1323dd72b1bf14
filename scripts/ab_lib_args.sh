#!/bin/bash
# like ab_lib.sh with bench arguments: scripts/ab_lib_args.sh reps "bench args" lib1.so lib2.so ...
reps=$1; args=$2; shift 2
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    MH_LIB_PATH=$PWD/moped_amd/$lib python bench.py --no-secondary --no-cpu-baseline --h2d-steps 0 --steps 10 --warmup 2 $args 2>/dev/null | grep "^{" \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$lib', 'round $r', d['value'], 'frames/s', 'objects', d['config']['objects_per_frame'], 'stage', (r.get('match_stage') or {}).get('kernels_ms'))"
  done
done
