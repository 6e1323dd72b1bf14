# frames per MATCH launch sequence x slots, config 1 (one box, back to back): bash scripts/batch_sweep.sh
for b in 4 8 4 8 12 16; do for d in 16; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --h2d-steps 0 --batch $b --depth $d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', d['config']['frames_per_match_launch'], 'depth', $d, d['value'], d['roofline']['frac'], d['roofline']['match_stage']['kernels_ms'], flush=True)"
done; done
for b in 8 16; do timeout -k 10 300 python bench.py --no-cpu-baseline --h2d-steps 0 --batch $b --depth 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', d['config']['frames_per_match_launch'], 'depth 8', d['value'], flush=True)"; done
