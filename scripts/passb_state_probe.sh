#!/bin/bash
# Does pass B's isolated time depend on what the chip did just before?  bench.py's roofline leg after a short and after a long
# timed region, product library.
for st in "1 0" "10 2" "30 2" "1 0"; do
  set -- $st
  python bench.py --no-secondary --no-cpu-baseline --h2d-steps 0 --steps $1 --warmup $2 2>/dev/null | grep "^{" \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('steps $1 warmup $2:', d['value'], 'frames/s; stage ms', r['match_stage']['kernels_ms'], 'frac', r['frac'])"
done
