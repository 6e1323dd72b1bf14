#!/usr/bin/env bash
cd "$(dirname "$0")/.."
for q in 4 8; do for d in 3 4 6 8; do
  v=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --models 20 --steps 30 --warmup 3 --depth $d --frames-per-step 24 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  w=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python scripts/small_shard_probe.py 3 $d 2>/dev/null | tail -1)
  echo "hwq=$q depth=$d 20-model: $v | 3-model: $w"
done; done
