# pass A's sampling stride (every k-th tile) against pass B's hit count: config 1, one box, back to back
for k in 8 12 16 6 8; do
  MH_SCREEN_SAMPLE=$k timeout -k 10 300 python bench.py --no-cpu-baseline --h2d-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); m=d['roofline']['match_stage']; print('sample', $k, d['value'], 'stage', m['ms_per_launch'], m['kernels_ms'], 'cand/query', m['candidate_rows_per_query'], flush=True)"
done
