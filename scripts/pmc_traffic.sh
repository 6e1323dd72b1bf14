#!/usr/bin/env bash
# HBM traffic of the match kernel per launch (FETCH_SIZE / WRITE_SIZE passes), corrected as
# MI355X_MICROARCH.md prescribes (gfx950 FETCH_SIZE tallies 128-B requests at 64 B: doubled).
# usage (GPU box): bash scripts/pmc_traffic.sh   -> gpurun_out/traffic.json
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "{" > $root/gpurun_out/traffic.json
for m in 20 200; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr_$m$c
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/tr_$m$c -- python3 $root/scripts/match_only.py $m 3000 5 > /tmp/tr_$m$c.log 2>&1 || { tail -3 /tmp/tr_$m$c.log; exit 1; }
  done
  python3 - $m >> $root/gpurun_out/traffic.json <<'PY'
import csv, glob, sys, collections
m = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/tr_{m}{c}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "match_mfma_kernel" in r["Kernel_Name"] or ("match_kernel" in r["Kernel_Name"])]
    out[c] = sum(v) / len(v)
fetch = int(out["FETCH_SIZE"] * 1024 * 2)
print(f' "match_{m}m_3000q": {fetch},')
print(f' "_raw_{m}m": "FETCH_SIZE {out["FETCH_SIZE"]:.1f} KB raw (x2 on gfx950), WRITE_SIZE {out["WRITE_SIZE"]:.1f} KB per launch",')
PY
done
echo ' "_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per match launch (scripts/pmc_traffic.sh), KB*1024, FETCH doubled per MI355X_MICROARCH.md"' >> $root/gpurun_out/traffic.json
echo "}" >> $root/gpurun_out/traffic.json
cat $root/gpurun_out/traffic.json
