#!/bin/bash
# A/B of experiment switches on one box: bench.py's value for each "NAME=VALUE[,NAME=VALUE..]" setting (or "-" = none), interleaved.
# The experiment build (make -C moped_amd/csrc EXTRA=-DMH_EXPERIMENTS BUILD=build_exp OUT=../libmoped_hip_exp.so) reads the switches;
# the product build ignores them.  usage: scripts/ab_env.sh reps "bench args" setting1 setting2 ...
reps=$1; args=$2; shift 2
export MH_LIB_PATH=$PWD/moped_amd/libmoped_hip_exp.so
for r in $(seq 1 $reps); do
  for st in "$@"; do
    envs=""; [ "$st" != "-" ] && envs=$(echo "$st" | tr ',' ' ')
    env $envs python bench.py --no-secondary --no-cpu-baseline --h2d-steps 0 --steps 10 --warmup 2 $args 2>/dev/null \
      | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline',{}); print('$st', 'round $r', d['value'], 'frames/s', 'objects', d['config']['objects_per_frame'], 'B', d['config']['frames_per_match_launch'], 'passB ms', r.get('ms_per_launch'), 'frac', r.get('frac'), 'stage', r.get('match_stage',{}).get('kernels_ms'))"
  done
done
