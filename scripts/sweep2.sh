#!/usr/bin/env bash
cd "$(dirname "$0")/.."
for g in 1 0; do for d in 4 8; do for m in 3 20; do
  v=$(MH_GRAPH=$g timeout -k 10 200 python bench.py --models $m --steps 30 --warmup 3 --depth $d --frames-per-step 16 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "no_graph=$g depth=$d models=$m -> $v"
done; done; done
