#!/usr/bin/env bash
# frames/s vs HW queues / frames in flight, 20-model (N=1 workload) and 3-model (per-rank load at N=8)
cd "$(dirname "$0")/.."
for q in 4 8 16; do for d in 4 8 16; do
  [ $d -gt $((q*2)) ] && continue
  for a in "--models 20" "--models 3" "--models 3 --force-exchange"; do
    v=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py $a --steps 20 --warmup 3 --depth $d --frames-per-step 16 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "hwq=$q depth=$d $a -> $v"
  done
done; done
