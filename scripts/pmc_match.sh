#!/usr/bin/env bash
# SQ counter passes over the match stage alone (scripts/match_only.py); one --pmc group per run.
# usage (on the GPU box): bash scripts/pmc_match.sh <out_dir> [n_models] [Q]
out=${1:-gpurun_out/pmc}; m=${2:-20}; Q=${3:-3000}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM" \
           "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc$i -- python3 "$root/scripts/match_only.py" $m $Q 3 > /tmp/pmc$i.log 2>&1 || { tail -5 /tmp/pmc$i.log; exit 1; }
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> "$root/$out/match_sq_counters.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "match_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:28s} per-launch mean {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
cat "$root/$out/match_sq_counters.txt"
