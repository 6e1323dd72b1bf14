#!/bin/bash
# A/B: the frames of a batch through group / CLUSTER / POSE / POSE2 in one launch per stage (default) vs frame after frame
for v in 1 0 1 0; do
  echo "MH_MERGE_BATCH=$v"
  MH_MERGE_BATCH=$v python3 bench.py --no-cpu-baseline --no-roofline --h2d-steps 0 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config'].get('host_issue_seconds'), d['config']['objects_per_frame'])"
done
