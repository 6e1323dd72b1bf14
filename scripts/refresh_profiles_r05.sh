#!/usr/bin/env bash
# Regenerates round 5's judged artifacts under gpurun_out/profiles_r05 (copy into profiles/ afterwards).
# usage (GPU box): bash scripts/refresh_profiles_r05.sh [part ...]   parts: bench prof traffic sq ranks latency feat stress misc (default: all)
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/profiles_r05
mkdir -p $out
parts=${*:-bench prof traffic sq ranks latency feat stress misc}
cd $root
has() { [[ " $parts " == *" $1 "* ]]; }
line() { grep "^{" | tail -1; }
if has bench; then
  timeout -k 10 600 python bench.py 2>$out/bench20.err | line > $out/r05_bench_20models.json || { tail -3 $out/bench20.err; exit 1; }
  echo "bench 20 done"
  timeout -k 10 600 python bench.py --models 200 --frames-per-step 256 --steps 10 --warmup 2 --no-secondary 2>$out/bench200.err | line > $out/r05_bench_200models.json || { tail -3 $out/bench200.err; exit 1; }
  echo "bench 200 done"
  timeout -k 10 500 python bench.py --models 50 --depth-kind 1 --frames-per-step 512 2>$out/bench50.err | line > $out/r05_bench_50models_depth.json || { tail -3 $out/bench50.err; exit 1; }
  timeout -k 10 500 python bench.py --models 50 --depth-kind 1 --moped3d-frontend --frames-per-step 512 --no-cpu-baseline 2>$out/bench50f.err | line > $out/r05_bench_50models_moped3d_frontend.json || { tail -3 $out/bench50f.err; exit 1; }
  timeout -k 10 500 python bench.py --models 50 --depth-kind 1 --moped3d-frontend --depthfill --frames-per-step 512 --no-cpu-baseline 2>$out/bench50d.err | line > $out/r05_bench_50models_moped3d_frontend_depthfill.json || { tail -3 $out/bench50d.err; exit 1; }
  echo "bench 50 done"
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  # the same command as the judged line under the profiler: default depth, and one launch sequence on the chip at a time
  for d in default 1; do
    a=""; [ $d = 1 ] && a="--depth 1 --frames-per-step 128"
    rm -rf /tmp/rp_$d
    timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$d -- python3 $root/bench.py --no-cpu-baseline --no-secondary --h2d-steps 0 $a 2>/tmp/rp_$d.err | line > $out/r05_bench_20models_depth${d}_under_rocprof.json || { tail -3 /tmp/rp_$d.err; exit 1; }
    cp $(grep -l screen $(find /tmp/rp_$d -name "*kernel_stats.csv") | head -1) $out/r05_bench_20models_depth${d}_kernel_stats.csv   # (the bench's own process: rocprofv3 also writes a file for the mfma_rate child)
    echo "rocprof depth $d done"
  done
  rm -rf /tmp/rp_200
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_200 -- python3 $root/bench.py --models 200 --depth 1 --frames-per-step 32 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --h2d-steps 0 2>/tmp/rp_200.err | line > $out/r05_bench_200models_depth1_under_rocprof.json || { tail -3 /tmp/rp_200.err; exit 1; }
  cp $(grep -l screen $(find /tmp/rp_200 -name "*kernel_stats.csv") | head -1) $out/r05_bench_200models_depth1_kernel_stats.csv
  cd $root
fi
if has traffic; then
  # HBM traffic of the two-stage MATCH's kernels per launch (separate --pmc passes, never with a trace); FETCH_SIZE
  # doubled as MI355X_MICROARCH.md prescribes for gfx950
  cd /tmp && export TMPDIR=/tmp
  echo "{" > $out/traffic_r05.json
  for mq in "20 48000" "200 48000" "20 24000" "200 24000" "20 3000" "50 12000"; do
    set -- $mq; m=$1; q=$2
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/trs_$m$c
      timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/trs_$m$c -- python3 $root/scripts/screen_probe.py $m $q 5 > /tmp/trs_$m$c.log 2>&1 || { tail -3 /tmp/trs_$m$c.log; exit 1; }
    done
    python3 - $m $q >> $out/traffic_r05.json <<'PY'
import csv, glob, sys
m, q = sys.argv[1], sys.argv[2]
names = {"screen_b": ("screen16_kernel<1", "screen_kernel<1"), "screen_a": ("screen16_kernel<0", "screen_kernel<0"),
         "rescore": ("rescore_kernel",), "match": ("match_mfma_kernel",)}
res, which = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/trs_{m}{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for k, pats in names.items():
        for pat in pats:
            v = [float(r["Counter_Value"]) for r in rows if pat in r["Kernel_Name"]]
            if v:
                res[(k, c)] = sum(v) / len(v)
                which[k] = pat
                break
for k in names:
    if (k, "FETCH_SIZE") in res:
        print(f' "{k}_{m}m_{q}q": {int(res[(k, "FETCH_SIZE")] * 1024 * 2)},')
        print(f' "_raw_{k}_{m}m_{q}q": "{which[k]}: FETCH_SIZE {res[(k, "FETCH_SIZE")]:.1f} KB raw (x2 on gfx950), WRITE_SIZE {res.get((k, "WRITE_SIZE"), 0):.1f} KB per launch",')
PY
  done
  echo ' "_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per launch over scripts/screen_probe.py (scripts/refresh_profiles_r05.sh traffic), KB*1024, FETCH doubled per MI355X_MICROARCH.md"' >> $out/traffic_r05.json
  echo "}" >> $out/traffic_r05.json
  cat $out/traffic_r05.json
  cd $root
fi
if has sq; then
  # SQ counters of pass B at the judged launch shape (20 models, 16 frames = 48 000 queries: screen16_kernel<1, 4>)
  bash scripts/pmc_screen.sh gpurun_out/profiles_r05/r05_screen_sq_counters.txt 20 48000 "screen16_kernel<1" > /dev/null 2>&1
  echo "sq done"; head -40 $out/r05_screen_sq_counters.txt
fi
if has ranks; then
  # one GPU carrying what a rank carries at N = 8 (20 models dealt round-robin: 3 or 2 models per rank; the frame's two
  # visible models land on two ranks -- a rank sees one of them or none), 4 and 2, behind the C-ABI RCCL exchange (world 1)
  { for a in "--models 3 --n-vis 1 --force-exchange" "--models 3 --n-vis 0 --force-exchange" "--models 2 --n-vis 1 --force-exchange" "--models 3 --n-vis 2 --force-exchange" \
             "--models 5 --n-vis 1 --force-exchange" "--models 10 --n-vis 1 --force-exchange" "--models 10 --n-vis 2 --force-exchange" "--models 3 --n-vis 1 --force-exchange --batch 8" \
             "--models 3 --n-vis 3 --force-exchange" "--models 5 --n-vis 5 --force-exchange" "--models 10 --n-vis 5 --force-exchange"; do
      echo "bench.py $a"; timeout -k 10 400 python bench.py $a --no-cpu-baseline --no-roofline --no-secondary --h2d-steps 0 2>/dev/null | line | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('   ', d['value'], 'frames/s;', 'frames per MATCH launch', c.get('frames_per_match_launch'), '; objects per frame', c.get('objects_per_frame'), '; exchange', c.get('exchange'))"; done; } > $out/r05_per_rank_load_n8.txt 2>&1
  { for a in "--models 25 --n-vis 1" "--models 25 --n-vis 0" "--models 50 --n-vis 1 --frames-per-step 512" "--models 100 --n-vis 1 --frames-per-step 256 --steps 10"; do
      echo "bench.py $a --force-exchange"; timeout -k 10 500 python bench.py $a --force-exchange --no-cpu-baseline --no-roofline --no-secondary --h2d-steps 0 2>/dev/null | line | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('   ', d['value'], 'frames/s;', 'frames per MATCH launch', c.get('frames_per_match_launch'), '; objects per frame', c.get('objects_per_frame'))"; done; } > $out/r05_per_rank_load_200models.txt 2>&1
  cat $out/r05_per_rank_load_n8.txt $out/r05_per_rank_load_200models.txt
fi
if has latency; then
  # one frame at a time (the reference's calling convention, moped.cpp:183-191): wall latency, the kernels of an isolated
  # frame from a rocprofv3 kernel trace, POSE's phases from the POSE_PROF build (if it travelled with the snapshot)
  { python3 scripts/single_frame_timeline.py run 20 2 2>&1 | grep -v amdgpu.ids
    python3 scripts/single_frame_timeline.py run 20 5 2>&1 | grep -v amdgpu.ids
    python3 scripts/single_frame_timeline.py run 20 10 2>&1 | grep -v amdgpu.ids
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sft
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sft -- python3 $root/scripts/single_frame_timeline.py run 20 2 > /tmp/sft.log 2>&1
    python3 $root/scripts/single_frame_timeline.py report /tmp/sft 2>&1 | grep -v amdgpu.ids
    cd $root
    if [ -f moped_amd/libmoped_hip_trace.so ]; then
      echo "# trace build: the workgroups of CLUSTER / POSE / POSE2 of one synchronous frame (scripts/single_frame_trace.py)"
      python3 scripts/single_frame_trace.py 20 2 20 2>&1 | grep -v amdgpu.ids
    fi
    if [ -f moped_amd/libmoped_hip_poseprof.so ]; then
      echo "# POSE_PROF build (cycles of thread 0 of the tasks that refine an object; scripts/pose_prof.py)"
      MH_LIB_PATH=$root/moped_amd/libmoped_hip_poseprof.so python3 scripts/pose_prof.py 2>&1 | grep -v amdgpu.ids
    fi; } > $out/r05_stage_latency.txt 2>&1
  cat $out/r05_stage_latency.txt
fi
if has feat; then
  # FEAT at the metric's keypoint count (synth.textured_image: ~3 200 keypoints) and on the bundled frame (~590)
  timeout -k 10 500 python3 scripts/sift_size_probe.py 2>&1 | grep -v amdgpu.ids > $out/r05_sift_size_probe.txt
  cd /tmp && export TMPDIR=/tmp
  for im in textured bundled; do
    rm -rf /tmp/sp_$im
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$im -- python3 $root/scripts/sift_size_probe.py 300 $im > /dev/null 2>&1
    cp $(find /tmp/sp_$im -name "*kernel_stats.csv" | head -1) $out/r05_sift_${im}_kernel_stats.csv
  done
  cd $root
  timeout -k 10 300 python3 scripts/image_frame_bench.py 20 16 2000 > $out/r05_image_frame_bench.txt 2>&1
  timeout -k 10 300 python3 scripts/image_frame_bench.py 20 16 2000 8 >> $out/r05_image_frame_bench.txt 2>&1
  timeout -k 10 300 python3 scripts/image_frame_bench.py 20 16 2000 16 >> $out/r05_image_frame_bench.txt 2>&1
  # the batch path's kernels (16 slots x 16 images) and the FEAT stages' HBM traffic (counter passes of their own)
  cd /tmp && rm -rf /tmp/ifb
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ifb -- python3 $root/scripts/image_frame_bench.py 20 16 2000 16 > /dev/null 2>&1
  cp $(find /tmp/ifb -name "*kernel_stats.csv" | head -1) $out/r05_image_batch16_kernel_stats.csv
  cd $root
  bash scripts/pmc_feat.sh $out/r05_feat_traffic.txt > /dev/null
  cat $out/r05_sift_size_probe.txt; grep -v amdgpu $out/r05_image_frame_bench.txt | grep "image->"; cat $out/r05_feat_traffic.txt
fi
if has stress; then
  # (the 2 x 1000-scene frame stress of round 5 runs as two calls of its own: gpurun_out/r05_frame_stress_{a,b}.txt)
  timeout -k 10 600 python3 tests/tools/depth_rules_stress.py 800 > $out/r05_depth_rules_stress.txt 2>&1; tail -1 $out/r05_depth_rules_stress.txt
  timeout -k 10 400 python3 scripts/screen_stress.py 200 11 > $out/r05_screen_stress.txt 2>&1; tail -1 $out/r05_screen_stress.txt
  timeout -k 10 600 python3 tests/tools/shard_stress.py 200 3 > $out/r05_shard_stress.txt 2>&1; tail -1 $out/r05_shard_stress.txt
  timeout -k 10 400 python3 tests/tools/depth_batch_stress.py 300 > $out/r05_depth_batch_stress.txt 2>&1; tail -1 $out/r05_depth_batch_stress.txt
fi
if has misc; then
  { for i in 1 2; do moped_amd/host/mfma_rate 2>/dev/null; done; } > $out/r05_mfma_shapes_rate.txt 2>&1
  # who holds the compute units (trace build) and an isolated pass B launch from the inside
  timeout -k 10 300 python scripts/cu_trace.py 20 16 16 64 full 2>&1 | grep -v amdgpu.ids > $out/r05_cu_trace_config1.txt
  timeout -k 10 300 python scripts/passb_timeline.py 20 16 2>&1 | grep -v amdgpu.ids > $out/r05_passb_timeline_20models.txt
  timeout -k 10 300 python scripts/passb_timeline.py 200 16 2>&1 | grep -v amdgpu.ids > $out/r05_passb_timeline_200models.txt
  # pass B's ablations at the judged shape (experiment build): MFMA shape, no records at all
  { echo "# scripts/ab_env.sh: bench.py --no-secondary, experiment build; per setting: frames/s, pass B ms (isolated, HIP events), frac of 2.5 PFLOP/s, the stage's kernels"
    timeout -k 10 500 scripts/ab_env.sh 2 "" - MH_SCREEN_SHAPE=1 MH_SCREEN_SHAPE=2 MH_SCREEN_NO_HITS=1 MH_SCREEN_SHAPE=1,MH_SCREEN_NW=4 2>&1 | cut -c1-300
    echo "# eight frames per launch sequence (the default until the end of round 5)"
    timeout -k 10 500 scripts/ab_env.sh 1 "--batch 8" - MH_SCREEN_SHAPE=1 MH_SCREEN_NO_HITS=1 2>&1 | cut -c1-300
    echo "# config 2 (200 models)"
    timeout -k 10 500 scripts/ab_env.sh 1 "--models 200 --frames-per-step 256 --steps 5" - MH_SCREEN_SHAPE=1 MH_SCREEN_NO_HITS=1 2>&1 | cut -c1-300; } > $out/r05_passb_ablations.txt
  # the C++ hosts
  timeout -k 10 200 scripts/cpp_host_probe.sh 2>&1 | grep -v amdgpu.ids > $out/r05_cpp_streaming_host.txt
  timeout -k 10 300 python scripts/host_step_timing.py > $out/r05_host_step_timing.txt 2>&1
  timeout -k 10 300 bash scripts/plugin_probe.sh 2>&1 | grep -v amdgpu.ids > $out/r05_plugin_probe.txt
fi
ls -la $out
