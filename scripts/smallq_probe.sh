# the match stage at small query counts (one image's keypoints, one frame) and the headline shape, per kernel
for q in 600 3000; do timeout -k 10 200 python scripts/screen_probe.py 20 $q 30 2>&1 | grep -v amdgpu | tail -3; done
timeout -k 10 200 python scripts/image_frame_bench.py 20 16 3000 2>&1 | grep "image->"
timeout -k 10 300 python bench.py --no-cpu-baseline --h2d-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 1', d['value'], d['roofline']['frac'], d['roofline']['match_stage']['kernels_ms'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --h2d-steps 0 --batch 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 1 frame by frame', d['value'], d['roofline']['match_stage']['kernels_ms'])"
