# POSE: inliers re-collected under the refined pose + second refine (default) against one pass (MH_POSE_REPASS=0)
for r in 1 0 1 0; do
  MH_POSE_REPASS=$r timeout -k 10 300 python bench.py --no-cpu-baseline --h2d-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MH_POSE_REPASS=$r config 1', d['value'], d['config']['objects_per_frame'], flush=True)"
done
for r in 1 0; do for sc in 283 573 306; do echo "MH_POSE_REPASS=$r scene $sc"; MH_POSE_REPASS=$r FRAME_STRESS_ONLY=$sc timeout -k 10 300 python tests/tools/frame_stress.py 600 2 2>&1 | grep "device objects:\|oracle objects:\|MISMATCH\|^note\|^score"; done; done
