#!/bin/bash
# A/B of library builds on one box: bench.py's config-1 line (value only) for each library named, interleaved, `reps` rounds.
# usage: scripts/ab_lib.sh reps lib1.so lib2.so ...   (paths relative to moped_amd/)
reps=$1; shift
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    MH_LIB_PATH=$PWD/moped_amd/$lib python bench.py --no-secondary --no-cpu-baseline --no-roofline --h2d-steps 0 --steps 10 --warmup 2 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'round $r', d['value'], 'frames/s', 'objects', d['config']['objects_per_frame'])"
  done
done
