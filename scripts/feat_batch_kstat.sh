#!/bin/bash
# per-kernel time per BATCH of 16 images (one batch on the chip at a time) for a list of library builds.
# usage: bash scripts/feat_batch_kstat.sh lib1.so lib2.so ...
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export MH_LIB_PATH=$root/moped_amd/$lib
  rm -rf /tmp/fbk
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fbk -- python3 $root/scripts/image_frame_bench.py 20 1 480 16 > /dev/null 2>&1
  python3 - $(find /tmp/fbk -name "*kernel_stats.csv" | head -1) $lib <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
det = [r for r in rows if "detect_kernel" in r["Name"]]
batches = int(det[0]["Calls"]) if det else 1
out = []
for r in rows:
    m = re.search(r"(\w+)_kernel", r["Name"])
    if m and m.group(1) in ("blur_jobs", "describe", "detect", "orient", "small_octaves", "prepare"):
        out.append(f"{m.group(1)} {float(r['TotalDurationNs']) / batches / 1e3:.0f}")
print(sys.argv[2], "| us per batch of 16:", ", ".join(out), flush=True)
PY
done
