#!/bin/bash
# FEAT after a kernel change (GPU box): the SIFT parity tests, SIFT alone / image -> objects at both keypoint counts, and
# the per-kernel table of one image alone.  usage: bash scripts/feat_probe.sh TAG [lib.so]
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$2" ] && export MH_LIB_PATH=$root/moped_amd/$2
out=$root/gpurun_out
python -m pytest tests/test_gpu_sift.py tests/test_gpu_image_frame.py -m gpu -x -q > $out/feat_${tag}_tests.log 2>&1; tail -3 $out/feat_${tag}_tests.log
timeout -k 10 300 python3 scripts/sift_size_probe.py 600 2>&1 | grep -v amdgpu.ids | tee $out/feat_${tag}_probe.txt
cd /tmp && export TMPDIR=/tmp
for im in textured bundled; do
  rm -rf /tmp/sp_$im
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$im -- python3 $root/scripts/sift_size_probe.py 300 $im > /dev/null 2>&1
  cp $(find /tmp/sp_$im -name "*kernel_stats.csv" | head -1) $out/feat_${tag}_${im}_kernel_stats.csv
  python3 - $out/feat_${tag}_${im}_kernel_stats.csv $im <<'PY'
import csv, re, sys
tot = 0
print("==", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) < 100: continue
    m = re.search(r"(\w+_kernel|fillBuffer\w*)", r["Name"])
    per = float(r["TotalDurationNs"]) / 304 / 1000
    tot += per
    print(f"  {m.group(1) if m else r['Name'][:20]:26s} x{int(r['Calls']) // 304:2d} per image {per:7.1f} us")
print(f"  sum {tot:.1f} us")
PY
done
