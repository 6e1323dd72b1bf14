#!/bin/bash
# bench.py's value for several library builds x several bench argument strings, interleaved, on one box.
# usage: scripts/ab_libs_args.sh reps "lib1.so lib2.so" "args1" "args2" ...   (libs relative to moped_amd/; env passes through)
reps=$1; libs=$2; shift 2
for a in "$@"; do
  echo "== bench.py $a"
  for r in $(seq 1 $reps); do
    for lib in $libs; do
      MH_LIB_PATH=$PWD/moped_amd/$lib python bench.py --no-secondary --no-cpu-baseline --no-roofline --h2d-steps 0 --steps 10 --warmup 2 $a 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   $lib', 'round $r', d['value'], 'frames/s; objects', d['config']['objects_per_frame'])"
    done
  done
done
