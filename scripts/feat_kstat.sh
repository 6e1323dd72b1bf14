#!/bin/bash
# per-kernel table of SIFT alone (one image at a time) for a list of library builds.  usage: bash scripts/feat_kstat.sh IMAGE lib1.so lib2.so ...
im=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export MH_LIB_PATH=$root/moped_amd/$lib
  rm -rf /tmp/sp_k
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_k -- python3 $root/scripts/sift_size_probe.py 300 $im > /dev/null 2>&1
  python3 - $(find /tmp/sp_k -name "*kernel_stats.csv" | head -1) $lib <<'PY'
import csv, re, sys
row = []
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) < 100: continue
    m = re.search(r"(\w+)_kernel", r["Name"])
    if m: row.append(f"{m.group(1)} {float(r['TotalDurationNs']) / 304 / 1000:.1f}")
print(sys.argv[2], "|", ", ".join(row), flush=True)
PY
done
