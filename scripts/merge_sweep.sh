#!/bin/bash
# merged rest launches: frames per MATCH launch sequence x slots (config 1, one box, back to back)
run() { echo -n "merge=$1 batch=$2 depth=$3: "; MH_MERGE_BATCH=$1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-roofline --h2d-steps 0 --batch $2 --depth $3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config'].get('host_issue_seconds'), d['config']['hbm_pipeline_mb'])"; }
run 1 4 16; run 0 4 16; run 1 8 16; run 0 8 16; run 1 8 8; run 1 4 8; run 1 8 12; run 1 4 16; run 0 4 16
