#!/usr/bin/env bash
# usage: sweep.sh  -- prints frames/s for HW-queue / depth combinations
cd "$(dirname "$0")/.."
for q in 4 8 16; do
  for d in 4 8; do
    v=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --depth $d 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "hwq $q depth $d -> $v"
  done
done
