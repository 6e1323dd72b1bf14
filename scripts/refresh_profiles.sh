#!/usr/bin/env bash
# Regenerates the judged artifacts under gpurun_out/profiles_new (copy into profiles/ afterwards).
# usage (GPU box): bash scripts/refresh_profiles.sh
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/profiles_new
mkdir -p $out
cd $root
timeout -k 10 300 python bench.py > $out/r01_bench_20models.json 2>$out/bench20.err || exit 1
timeout -k 10 400 python bench.py --models 200 --steps 6 --warmup 2 > $out/r01_bench_200models.json 2>$out/bench200.err || exit 1
timeout -k 10 300 python bench.py --models 50 --depth-kind 1 > $out/r01_bench_50models_depth.json 2>$out/bench50.err || exit 1
timeout -k 10 300 python bench.py --models 50 --depth-kind 1 --moped3d-frontend > $out/r01_bench_50models_moped3d_frontend.json 2>$out/bench50f.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp20 -- python3 $root/bench.py --no-cpu-baseline > $out/r01_bench_20models_under_rocprof.json 2>/tmp/rp20.err || { tail /tmp/rp20.err; exit 1; }
cp $(find /tmp/rp20 -name "*kernel_stats.csv" | head -1) $out/r01_bench_20models_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp20d1 -- python3 $root/bench.py --depth 1 --no-cpu-baseline > $out/r01_bench_20models_depth1_under_rocprof.json 2>/tmp/rp20d1.err || { tail /tmp/rp20d1.err; exit 1; }
cp $(find /tmp/rp20d1 -name "*kernel_stats.csv" | head -1) $out/r01_bench_20models_depth1_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp200 -- python3 $root/bench.py --models 200 --depth 1 --steps 4 --warmup 1 --no-cpu-baseline > $out/r01_bench_200models_depth1_under_rocprof.json 2>/tmp/rp200.err || { tail /tmp/rp200.err; exit 1; }
cp $(find /tmp/rp200 -name "*kernel_stats.csv" | head -1) $out/r01_bench_200models_depth1_kernel_stats.csv
ls -la $out
# ---- round-1 additions: stage latency, per-rank load at N=8, CLUSTER / POSE phases, FEAT, image frames ----
cd $root
{ python scripts/stage_latency_probe.py 20 1; python scripts/stage_latency_probe.py 20 4; python scripts/stage_latency_probe.py 3 16; } > $out/r01_stage_latency.txt 2>&1
{ GPU_MAX_HW_QUEUES=16 python scripts/rest_only_probe.py 3 16;
  # the per-rank load of N = 8 / 4 / 2 GPUs on one GPU: a DB of 20/N models behind the exchange path
  # (defaults: 16 slots x 8 frames per MATCH launch), then 32 x 8, 16 x 4, frame by frame
  for a in "--models 3 --force-exchange" "--models 5 --force-exchange" "--models 10 --force-exchange" \
           "--models 3 --force-exchange --depth 32" "--models 3 --force-exchange --depth 16 --batch 4" \
           "--models 3 --force-exchange --depth 16 --batch 1" "--models 3 --depth 16"; do
    echo "bench.py $a"; python bench.py $a --no-cpu-baseline --no-roofline | tail -1; done; } > $out/r01_per_rank_load_n8.txt 2>&1
# the same for the 200-model DB (BASELINE configs[3]): per-rank load of 2 / 4 / 8 GPUs
{ for a in "--models 100 --steps 6" "--models 50 --steps 10" "--models 25"; do echo "bench.py $a --force-exchange"; python bench.py $a --force-exchange --no-cpu-baseline --no-roofline | tail -1; done; } > $out/r01_per_rank_load_200models.txt 2>&1
python tests/tools/ms_bench.py > $out/r01_meanshift_bench.txt 2>&1
python tests/tools/sift_probe.py > $out/r01_sift_probe.txt 2>&1
python scripts/image_frame_bench.py 20 4 > $out/r01_image_frame_bench.txt 2>&1
python scripts/host_step_timing.py > $out/r01_host_step_timing.txt 2>&1
python tests/tools/linkage_bench.py > $out/r01_linkage_bench.txt 2>&1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpsift -- python3 $root/tests/tools/sift_probe.py > /tmp/rpsift.log 2>&1 && cp $(find /tmp/rpsift -name "*kernel_stats.csv" | head -1) $out/r01_sift_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpimg -- python3 $root/scripts/image_frame_bench.py 20 4 > /tmp/rpimg.log 2>&1 && cp $(find /tmp/rpimg -name "*kernel_stats.csv" | head -1) $out/r01_image_frame_kernel_stats.csv
ls -la $out
