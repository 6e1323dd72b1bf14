#!/usr/bin/env bash
# Which mode is this host in (on about one box in ten a process gets fewer hardware queues' worth of
# concurrency), and what does bench.py's slot calibration settle on?  usage (GPU box): bash scripts/slowdiag.sh
v() { python bench.py --models 3 --force-exchange --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'slots x frames =', d['config']['frames_in_flight'])"; }
echo "load: $(cat /proc/loadavg)"
echo "32 slots: $(v --depth 32)"
echo "16 slots: $(v --depth 16)"
echo "12 slots: $(v --depth 12)"
echo "default (calibrated): $(v)"
