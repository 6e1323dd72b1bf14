# CLUSTER launch shape A/B on one box: MH_MS_GRID=0 = one workgroup per model (the old shape), default = a few workgroups
set -e
for rep in 1 2; do for g in default 0; do
  if [ $g = default ]; then unset MH_MS_GRID; else export MH_MS_GRID=$g; fi
  echo "== MH_MS_GRID=$g"
  timeout -k 10 200 python scripts/image_frame_bench.py 20 16 3000 2>&1 | grep "image->"
  timeout -k 10 300 python bench.py --no-cpu-baseline --h2d-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 1', d['value'])"
  timeout -k 10 300 python bench.py --no-cpu-baseline --h2d-steps 0 --models 200 --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2', d['value'])"
done; done
