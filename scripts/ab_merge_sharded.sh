#!/bin/bash
# A/B on the per-rank load of a sharded DB (one rank's share of an 8-way split, exchange path on one GPU):
# the frames of a batch through the rest chain in one launch per stage (default) vs frame after frame
run() { echo -n "merge=$1 models=$2: "; MH_MERGE_BATCH=$1 timeout -k 10 200 python3 bench.py --models $2 --force-exchange --no-cpu-baseline --no-roofline --h2d-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config'].get('host_issue_seconds'), d['config']['objects_per_frame'])"; }
run 1 3; run 0 3; run 1 25; run 0 25; run 1 3; run 0 3
