#!/usr/bin/env bash
# SQ counter passes over the two-stage MATCH (scripts/screen_probe.py); one --pmc group per run, never
# together with a trace.  usage (on the GPU box): bash scripts/pmc_screen.sh <out_file> [n_models] [Q] [kernel substring]
out=${1:-gpurun_out/screen_sq_counters.txt}; m=${2:-20}; Q=${3:-3000}; kern=${4:-screen_kernel<1>}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$(dirname "$root/$out")"; : > "$root/$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcs$i -- python3 "$root/scripts/screen_probe.py" $m $Q 3 > /tmp/pmcs$i.log 2>&1 || { echo "group $i failed: $grp"; tail -3 /tmp/pmcs$i.log; continue; }
  f=$(find /tmp/pmcs$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$kern" >> "$root/$out" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} per-launch mean {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
cat "$root/$out"
