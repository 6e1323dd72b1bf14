#!/usr/bin/env bash
# HBM traffic of the FEAT kernels in batches of 16 images (FETCH_SIZE / WRITE_SIZE in passes of their own, FETCH doubled on
# gfx950 as MI355X_MICROARCH.md prescribes) against their durations with one batch on the chip at a time.
# usage (GPU box): bash scripts/pmc_feat.sh [out=gpurun_out/feat_traffic.txt]
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-$root/gpurun_out/feat_traffic.txt}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ft_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/ft_$c -- python3 $root/scripts/image_frame_bench.py 20 1 480 16 > /tmp/ft_$c.log 2>&1 || { tail -3 /tmp/ft_$c.log; exit 1; }
done
rm -rf /tmp/ft_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ft_trace -- python3 $root/scripts/image_frame_bench.py 20 1 480 16 > /tmp/ft_trace.log 2>&1 || { tail -3 /tmp/ft_trace.log; exit 1; }
python3 - > $out <<'PY'
import csv, glob, re, collections
names = ("prepare_kernel", "blur_jobs_kernel", "small_octaves_kernel", "detect_kernel", "orient_kernel", "describe_kernel", "normalize_kernel")
def short(n):
    for k in names:
        if k in n: return k
    return None
by = {c: collections.defaultdict(list) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in by:
    f = glob.glob(f"/tmp/ft_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k: by[c][k].append(float(r["Counter_Value"]))
dur = {}
f = glob.glob("/tmp/ft_trace/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = short(r["Name"])
    if k: dur[k] = (float(r["TotalDurationNs"]), int(r["Calls"]))
print("FEAT in batches of 16 images (scripts/pmc_feat.sh: image_frame_bench.py 20 1 480 16, one batch on the chip at a time);")
print("bytes = FETCH_SIZE x 2 (gfx950 counts 128-byte requests as 64) + WRITE_SIZE, KB x 1024, summed over a stage's launches of one batch")
print(f"{'stage':22s} {'launches/batch':>14s} {'fetched MB':>11s} {'written MB':>11s} {'us/batch':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
tot_b = tot_t = 0
batches = dur["detect_kernel"][1]          # one detect launch per batch, warm-up batches included (every pass runs the same program)
for k in names:
    if k not in dur or not by["FETCH_SIZE"][k]: continue
    t_ns, calls = dur[k]
    fetch_b = sum(by["FETCH_SIZE"][k]) * 1024 * 2 / batches
    write_b = sum(by["WRITE_SIZE"][k]) * 1024 / batches
    t_b = t_ns / batches / 1e3
    gbs = (fetch_b + write_b) / (t_b * 1e-6) / 1e9
    tot_b += fetch_b + write_b; tot_t += t_b
    print(f"{k:22s} {calls / batches:14.1f} {fetch_b / 1e6:11.1f} {write_b / 1e6:11.1f} {t_b:9.1f} {gbs:8.0f} {gbs / 8000:9.3f}")
print(f"{'all FEAT stages':22s} {'':14s} {'':11s} {tot_b / 1e6:11.1f} {tot_t:9.1f} {tot_b / (tot_t * 1e-6) / 1e9:8.0f} {tot_b / (tot_t * 1e-6) / 1e9 / 8000:9.3f}")
PY
cat $out
