#!/usr/bin/env python3
"""bench.py -- detections/sec of libmoped's MATCH -> CLUSTER -> POSE -> FILTER -> POSE2 ->
FILTER2 path on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W

A step = one batch of `--frames-per-step` synthetic 640x480 frames (~3k SIFT
keypoints, 2 planted objects; `--frame-pool` distinct frames in turn) against an
N-model database, inputs resident in HBM, `--depth` frame slots in flight per GPU.

N > 1 = one process per GPU.  Started under torch.distributed.run the ranks are the
launcher's; started as plain `python bench.py --gpus N` this process starts the N
ranks itself (a child torch.distributed.run, BEFORE anything here touches the GPU)
and relays rank 0's line -- `n_gpus` is always the number of ranks that ran.  The
partition at N > 1 is BASELINE.json north_star's -- the model database sharded by
model, one all-gather of SURVEY.md 8(e) per batch of frames over RCCL -- as a
models x frames GRID: G model shards x R frame groups (G R = N).  The G ranks of a
frame group hold the whole DB between them and exchange over communicators of their
own; the R groups work on different frames and never talk.  R = 1 is pure model
sharding (total work fixed: "strong" scaling), G = 1 pure frame splitting;
`--parallelism auto` takes the largest G whose projected rate (committed per-rank
loads, profiles/per_rank_load.json) reaches 0.75 N times the single GPU's, else the
best G >= 2 (20 models at N = 8: 2 x 4; 200 models: 8 x 1).  Secondary measurements
ride in the same line: `pure_model_shard` (when auto took R > 1), `replicated_frames`
(DB replicated, frames split, no collective) and `sharded_200_models` (BASELINE
configs[3]: the 200-model DB sharded N ways).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector = FP32-input MFMA peak
PEAK_F16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense BF16/F16 MFMA (the F16 forms take the same cycles)
# What the matrix pipes sustain on RANDOM operands out of registers, two wavefronts per SIMD, every CU (the clock the power
# budget allows under matrix load): v_mfma_f32_16x16x32_f16 (pass B's instruction) 1 842-1 994 TFLOP/s over the boxes of the
# pool, v_mfma_f32_32x32x16_f16 1 625-1 771 (profiles/r03_mfma_shapes_rate.txt).  bench.py measures the box it runs on
# (moped_amd/host/mfma_rate --json) and falls back to this figure.
SUSTAINED_F16_TFLOPS = 1900.0
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E spec
N_CU = 256
DELIVER_CAP = 32           # objects per frame a delivery record carries (mh_frame_fetch_batch_async); n_objects says if there were more


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--models", type=int, default=20, help="models in the DB (5000 points each)")
    ap.add_argument("--queries", type=int, default=3000)
    ap.add_argument("--frames-per-step", type=int, default=1024,
                    help="frames per step (20 steps = 20,480 frames: a timed region of seconds, not milliseconds)")
    ap.add_argument("--frame-pool", type=int, default=100,
                    help="distinct synthetic frames the steps cycle through (SURVEY 8(d): 100 frames, seeds 0..99)")
    ap.add_argument("--depth", type=int, default=0,
                    help="frame slots (streams) per GPU (default 16: one stream per hardware queue; a frame is a chain of "
                         "~13 short launches, most of them latency bound -- 32 are faster on most boxes and 2.4x slower on some)")
    ap.add_argument("--no-adaptive", action="store_true",
                    help="POSE evaluates all 1024 hypotheses of every (cluster, replica) task instead of stopping after 256 "
                         "when the inlier ratio allows it")
    ap.add_argument("--h2d-steps", type=int, default=3,
                    help="steps of the secondary measurement with the descriptors in pinned host memory (0 = skip)")
    ap.add_argument("--n-vis", type=int, default=2)
    ap.add_argument("--depth-kind", type=int, default=0,
                    help="0 = moped2 residuals; 1/2 = moped3d back-projection / reprojection+depth (config 5)")
    ap.add_argument("--moped3d-frontend", action="store_true",
                    help="with --depth-kind: the frame takes the depth map itself and runs moped3d's shipped front end on "
                         "the device (DEPTHFILTER, depth-adaptive ratio, DEPTHFILTER2, DEPTHMAP_PROP, CLUSTER_LINKAGE; "
                         "moped3d/libmoped/src/config.hpp:41-45) instead of per-query depth attributes + mean shift")
    ap.add_argument("--depthfill", action="store_true",
                    help="with --moped3d-frontend: the sensor's depth map arrives with holes (z < 0 on ~12%% of the pixels) and "
                         "moped3d's DEPTHFILL step (DEPTH_FILL_EXACT_CPU(8, false), config.hpp:39) runs on the device for every "
                         "frame inside the timed region: a 4.9 MB working copy, the fill, the distance map")
    ap.add_argument("--parallelism", choices=("auto", "models", "frames", "grid"), default="auto",
                    help="N > 1: 'models' (the north-star design) shards the DB by model over all N ranks with one "
                         "all-gather per batch of frames; 'frames' replicates the DB and gives every rank its own frames (no "
                         "exchange: SURVEY 8(e)'s alternative for DBs too small to shard); 'grid' = G model shards x R frame "
                         "groups (--grid, or the table's choice); 'auto' = the grid the committed per-rank loads project "
                         "fastest under the rule in choose_partition (pure model sharding wherever that reaches 0.75 N)")
    ap.add_argument("--grid", default="",
                    help="'GxR' for --parallelism grid: G model shards per frame group x R frame groups, G R = the ranks")
    ap.add_argument("--assign", choices=("block", "round-robin"), default="round-robin",
                    help="model -> rank assignment of a sharded DB: round-robin (rank r owns models r, r + N, ...; SURVEY 8(e): spreads the "
                         "visible models' POSE work over the ranks) or contiguous blocks")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per MATCH launch and exchange (default: the library's choice for the partition; 1 = every "
                         "frame on its own)")
    ap.add_argument("--comms", type=int, default=4,
                    help="RCCL communicators per rank for the sharded path (slot i uses communicator i %% comms)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="single rank, but run the N > 1 code path (match_local -> RCCL all-gather -> rest)")
    ap.add_argument("--lane", default="",
                    help="'streams,reserve,low_priority': the chip-filling MATCH passes of all slots on `streams` shared "
                         "streams that leave `reserve` compute units of each XCD to the small kernels (CU mask), or (reserve "
                         "0, low_priority 1) of the lowest stream priority; '' or 'off' = everything on the slot's own stream")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements (replicated_frames, sharded_200_models, single-frame latency, "
                         "the C++ hosts)")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--watchdog", type=int, default=-1,
                    help="seconds after which a rank that is still running dumps every thread's Python stack to stderr and "
                         "exits with an error instead of hanging (default: 900 for N > 1, off for N = 1; 0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks, relay rank 0's line
# ------------------------------------------------------------------------------------------------------------
def count_devices() -> int:
    """GPUs this process could use, WITHOUT a HIP call (torch.cuda.device_count() falls back to hipGetDeviceCount on
    builds without amdsmi, which initialises the runtime in this parent): the visible-devices lists if set, else the KFD
    topology's nodes with SIMDs.  The topology may show more devices than a container may open; a rank that cannot get
    its device then fails at start-up, which is an error exit like the one below.  Unknown -> torch's count."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip()])
    n, base = 0, "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        n += int(line.split()[1]) > 0
                        break
    except (OSError, ValueError):
        n = 0
    if n:
        return n
    import torch
    return torch.cuda.device_count()


def arm_watchdog(seconds: int):
    """(Re-)arm the hang watchdog for the leg that starts now: a rank still inside it after `seconds` dumps every
    thread's stack and exits with an error.  Per leg, not per job: a long --steps or a slow host must not be killed for
    the sum of its legs."""
    import faulthandler
    faulthandler.cancel_dump_traceback_later()   # (seconds <= 0: off -- a timer armed for an earlier leg must not outlive it)
    if seconds > 0:
        faulthandler.dump_traceback_later(seconds, exit=True)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: one rank per GPU as a child
    `python -m torch.distributed.run`, started before this process has made any GPU call (a process that has
    initialised the GPU must not be replaced or forked into ranks: the ranks are a fresh child process, never an exec of
    this one, and the device count below comes from sysfs, not from HIP).  Returns the exit code to leave with; rank 0's
    JSON line is the only thing written to stdout.  Fewer devices than ranks is an error (never a line that names
    more GPUs than ran) unless MH_BENCH_REHEARSE=1 puts every rank on cuda:0 on purpose."""
    import socket
    n_dev = count_devices()
    rehearse = os.environ.get("MH_BENCH_REHEARSE") == "1"
    if n_dev < args.gpus and not rehearse:
        print(f"bench.py: --gpus {args.gpus} but this host shows {n_dev} device(s); not printing a line for ranks that "
              f"did not run", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MH_BENCH_LAUNCHED_BY="bench.py")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    line = None
    for out in child.stdout:
        if out.startswith("{") and line is None:
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks ended without a result line", file=sys.stderr)
        return 3
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc


def host_cpu():
    """The box the CPU figure was taken on (north_star: "core count stated"): model name, logical CPUs the OS shows, CPUs
    this process may run on."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = None
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "usable_cpus": usable}


def cpu_baseline(db, frames, args):
    """Bounded CPU sample on the host cores (rank 0, N=1): the shipped-default
    matcher of the reference (ANN kd-tree, eps=5) when oracle/_ref was built,
    else the oracle's exact matcher on a query subset; CLUSTER..FILTER2 = oracle
    port with the reference's constants and OpenMP structure."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orclib
    from moped_amd import synth
    cores = min(os.cpu_count() or 1, 16)
    dbn = orclib.normalize(db.desc)
    use_ref = orclib.ref_available(fast=True)
    ann = orclib.RefAnn(dbn, fast=True) if use_ref else None
    # bounded sample: frames of the workload in turn until ~30 s of CPU work (at least 3, at most 600 frames; only
    # 3 without the reference matcher: the exact matcher's untimed full search per frame is what takes long then).
    # Round 4's 10 s / ~170 frames moved 27% between boxes; the line now also carries the median and the fastest frame.
    budget_s, max_frames = 30.0, (600 if use_ref else 3)
    t_match, t_rest, n_obj = 0.0, 0.0, 0
    best_threads = None
    n_frames = 0
    per_frame = []
    while n_frames < max_frames and (n_frames < 3 or t_match + t_rest < budget_s):
        fi = n_frames
        fr = frames[fi % len(frames)]
        n_frames += 1
        t0 = time.perf_counter()
        qn = orclib.normalize(fr.desc)
        if use_ref:
            idx2, d = ann.search2(qn, 5.0)   # Quality = 5 (config.hpp:83); serial like the omp critical
            idx, d1, d2 = idx2[:, 0].copy(), d[:, 0].copy(), d[:, 1].copy()
            tm = time.perf_counter() - t0
            t_match += tm
        else:
            sub = 300
            idx_s, d1_s, d2_s = orclib.match_2nn(dbn, qn[:sub], n_threads=cores)
            tm = (time.perf_counter() - t0) * (qn.shape[0] / sub)
            t_match += tm
            idx, d1, d2 = orclib.match_2nn(dbn, qn, n_threads=cores)  # untimed: inputs for the rest
        # the reference drivers use 4 threads (moped_test.cpp:244); take the best of {1, 4}
        trial = {}
        for th in ((1, 4) if fi == 0 else (best_threads,)):
            t0 = time.perf_counter()
            om, op, osc, cnt = orclib.frame_rest(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models,
                                                 synth.K_DEFAULT, synth.CAM_IDENTITY, n_threads=th, seed=fi)
            trial[th] = time.perf_counter() - t0
        if fi == 0:
            best_threads = min(trial, key=trial.get)
        t_rest += trial[best_threads]
        per_frame.append(tm + trial[best_threads])
        n_obj += len(om)
    if ann:
        ann.close()
    fps = n_frames / (t_match + t_rest)
    pf = np.sort(np.array(per_frame))
    return {
        "value": round(fps, 3), "unit": "frames/s", "cores": max(1 if use_ref else cores, best_threads or 1), "kind": "port",
        # the sample's spread: frames/s of the median frame, of the fastest frame, of the slowest tenth
        "median_frame": round(1.0 / float(np.median(pf)), 3), "fastest_frame": round(1.0 / float(pf[0]), 3),
        "slowest_decile": round(1.0 / float(pf[int(0.9 * (len(pf) - 1))]), 3), "seconds": round(t_match + t_rest, 2),
        "host": host_cpu(),
        "sample": (f"{n_frames} frames of the same workload; MATCH = "
                   + ("reference ANN 1.1.1 kd-tree eps=5 via oracle/_ref (shipped default, 1 thread: omp critical)"
                      if use_ref else f"oracle exact matcher, {cores} threads, 300-query subset scaled")
                   + f"; CLUSTER..FILTER2 = oracle port, {best_threads} OpenMP thread(s)"),
        "match_ms": round(1e3 * t_match / n_frames, 2), "rest_ms": round(1e3 * t_rest / n_frames, 2),
        "objects_per_frame": n_obj / n_frames,
    }


PER_RANK_LOAD = os.path.join(ROOT, "profiles", "per_rank_load.json")


def per_rank_us(models_per_rank: float, table=None) -> float:
    """Microseconds per frame of ONE rank that owns `models_per_rank` models (5 000 points each; one of them visible
    in a frame), from the committed measurements (profiles/per_rank_load.json: bench.py --models m --n-vis 1
    --force-exchange on one MI355X, i.e. behind the real exchange entry points at world 1), piecewise linear in
    between, the last segment's slope beyond."""
    if table is None:
        with open(PER_RANK_LOAD) as f:
            table = json.load(f)["frames_per_s_by_models_per_rank"]
    pts = sorted((float(m), 1e6 / float(v)) for m, v in table.items())
    if models_per_rank <= pts[0][0]:
        return pts[0][1]
    for (m0, u0), (m1, u1) in zip(pts, pts[1:]):
        if models_per_rank <= m1:
            return u0 + (u1 - u0) * (models_per_rank - m0) / (m1 - m0)
    (m0, u0), (m1, u1) = pts[-2], pts[-1]
    return u1 + (u1 - u0) / (m1 - m0) * (models_per_rank - m1)


def projected_speedup(n_models: int, G: int, R: int, table=None) -> float:
    """R frame groups, each at the rate of a rank that owns n_models / G models, over the single GPU's rate."""
    return R * per_rank_us(n_models, table) / per_rank_us(n_models / G, table)


def choose_partition(n_models: int, world: int, parallelism: str = "auto", grid: str = "", table=None):
    """(G, R): G model shards per frame group x R frame groups, G R = world.
    'models' -> (world, 1): north_star's partition as it is written.  'frames' -> (1, world).  'grid' -> --grid, else
    like 'auto'.  'auto': model sharding stays the design (G >= 2 whenever world >= 2); a rank's cost per frame does not
    shrink with its shard below ~20 us (every rank still screens every query), so a small DB sharded 8 ways
    projects 2.7-2.9x where north_star asks >= 6x.  Rule: the LARGEST G dividing world whose projection reaches 0.75
    world; none does -> the G >= 2 with the best projection.  (20 models: 2 x 4 at N = 8, 2 x 2 at 4; 200 models: 8 x 1.)"""
    if world <= 1:
        return 1, 1
    if parallelism == "models":
        return world, 1
    if parallelism == "frames":
        return 1, world
    if grid:
        G, R = (int(x) for x in grid.lower().split("x"))
        if G < 1 or R < 1 or G * R != world:
            raise SystemExit(f"bench.py: --grid {grid} does not multiply to the {world} ranks that run")
        return G, R
    cands = [g for g in range(world, 1, -1) if world % g == 0 and g <= max(n_models, 2)]
    proj = {g: projected_speedup(n_models, g, world // g, table) for g in cands}
    for g in cands:   # descending
        if proj[g] >= 0.75 * world:
            return g, world // g
    g = max(cands, key=lambda k: proj[k])
    return g, world // g


def partition_label(G: int, R: int) -> str:
    world = G * R
    if world <= 1:
        return "single GPU"
    if R == 1:
        return f"model-shard x{world}"
    if G == 1:
        return f"frame-parallel x{world} (DB replicated)"
    return f"grid: {G} model shards x {R} frame groups"


def parse_lane(text: str):
    """'streams,reserve,low_priority' -> tuple, '' / 'off' -> None (moped_amd.pipeline.FramePipeline(lane=...))."""
    if not text or text == "off":
        return None
    v = [int(x) for x in text.split(",")]
    return (v[0], v[1] if len(v) > 1 else 2, bool(v[2]) if len(v) > 2 else False)


def scaling_label(G: int, R: int) -> str:
    """The contract's `scaling` key: nothing scales at N = 1; pure model sharding (R = 1) keeps the total work fixed;
    frame groups each bring their own frames -- the job's frames grow with R ("weak": with G = 1 the per-GPU work is
    fixed outright, in a grid each rank's share of a frame is 1/G of the single GPU's)."""
    if G * R <= 1:
        return "n/a"
    return "strong" if R == 1 else "weak"


def default_batch(args, sharded: bool, shards: int = 0) -> int:
    from moped_amd import capi
    if args.batch > 0:
        return min(args.batch, capi.MAX_BATCH)
    if args.depth_kind and sharded:
        return 1    # (the sharded batch path carries no per-frame depth attributes)
    if sharded:
        # a shard of a few thousand rows does not fill the chip for one frame's queries, and every launch of the rest
        # chain is shared by the frames of a batch: as many as the library takes
        return capi.MAX_BATCH if args.models // max(shards or args.gpus, 1) < 25 else 16
    # sixteen frames per MATCH launch sequence (48 000 queries: three rounds of pass B workgroups that sweep 49 tiles each
    # instead of two rounds of 38 for eight frames -- fewer prologues and a fuller last query block per frame: +2.7% on
    # config 1, +1.6% on config 2); a batch of plain frames also shares the launches of its rest chain (one group /
    # CLUSTER / POSE / POSE2 launch for all of them).  Round 4: frames with depth ATTRIBUTES share their launches like
    # plain frames, and so do frames that bring the depth map itself and run moped3d's front end on it (a depth map per
    # frame of the batch, DepthMaps): sixteen everywhere (front end: 5 270 / 5 920 / 6 290 frames/s at 4 / 8 / 16).
    return 16


class Job:
    """One partition of one workload on this rank: the pipeline, the resident inputs, the step loop."""

    def __init__(self, args, env, db, n_models_total, by_frames, sharded, batch, frames_per_step, seeds_base=0, part=None):
        """part = (G, R): G model shards per frame group x R frame groups (G R = the ranks; rank = r G + g).  Default: what
        by_frames / sharded said before the grid existed -- (1, world) / (world, 1)."""
        import torch
        from moped_amd import capi, synth
        from moped_amd.pipeline import FramePipeline, ShardedDB
        self.args, self.env, self.db = args, env, db
        self.by_frames, self.sharded, self.B = by_frames, sharded, batch
        rank, world, dev = env["rank"], env["world"], env["dev"]
        if part is None:
            part = (1, world) if by_frames else (world, 1)
        self.G, self.R = G, R = part
        assert G * R == world or (world == 1 and G == 1 and R == 1), (G, R, world)
        self.shard_rank, self.frame_group = rank % G, rank // G
        Q = args.queries
        n_frames = max(frames_per_step, 1)
        n_pool = max(1, min(args.frame_pool, n_frames))
        B = batch
        if B > 1:
            # a batch = B consecutive frames of the pool, the pool taken cyclically: lcm(pool, B) / B distinct batches
            # (100 frames, B = 16: 25 batches, every frame in four of them); a step = n_frames / B batches
            n_frames = max(B, n_frames // B * B)
            n_pool = max(1, min(n_pool, n_frames))
        self.n_frames, self.n_pool = n_frames, n_pool
        self.seeds_base = seeds_base
        # every frame group works on its own frames (group 0: the single GPU's); its G ranks hold the DB between them
        self.seeds_base = seeds_base + 1000 * self.frame_group
        if G == 1:
            self.shard = ShardedDB(db.desc, db.xyz, db.model_of, db.n_models, 0, 1)
        else:
            self.shard = ShardedDB(db.desc, db.xyz, db.model_of, db.n_models, self.shard_rank, G, assign=args.assign)
        params = capi.default_frame_params()
        if args.no_adaptive:
            params.pose1.n_hypotheses = -abs(params.pose1.n_hypotheses)
            params.pose2.n_hypotheses = -abs(params.pose2.n_hypotheses)
        # MH_BENCH_ABLATE (experiments; recorded under env_overrides, the line is then NOT the metric): what the steps after
        # MATCH cost the pipeline -- lm0: no LM refine, rep1: one replica per cluster, nostage2: stop after POSE
        for knob in os.environ.get("MH_BENCH_ABLATE", "").split(","):
            if knob == "lm0":
                params.pose1.lm_iters_l2 = params.pose1.lm_iters_l4 = params.pose2.lm_iters_l2 = params.pose2.lm_iters_l4 = 0
            elif knob == "rep1":
                params.pose1.max_objects_per_cluster = params.pose2.max_objects_per_cluster = 1
            elif knob == "nostage2":
                params.run_stage2 = 0
        if args.depth_kind:
            # moped3d's shipped constants (moped3d/libmoped/src/config.hpp:46-49)
            params.pose1.error_threshold = 8.0
            params.f1_min_points, params.f1_feature_distance, params.f1_min_score = 6, 4096.0, 2.0
            params.f2_min_points, params.f2_feature_distance, params.f2_min_score = 8, 8192.0, 1e-4
        self.params = params
        free0, _ = torch.cuda.mem_get_info(dev)
        # a grid's frame groups exchange among their own G ranks: RCCL ids from the group's shard 0 (id_leader), or -- the
        # rehearsal on one GPU -- the host transport over the group's gloo subgroup
        grid = G > 1 and R > 1
        self.pipe = FramePipeline(env["local_rank"], self.shard, depth=args.depth, max_queries=Q * B, params=params,
                                  force_exchange=args.force_exchange and sharded, n_comms=args.comms, batch=B,
                                  lane=parse_lane(args.lane),
                                  group=env["subgroups"](G)[self.frame_group] if grid and env.get("rehearse") else None,
                                  id_leader=self.frame_group * G if grid and not env.get("rehearse") else None)
        torch.cuda.synchronize(dev)
        self.hbm_pipeline_mb = (free0 - torch.cuda.mem_get_info(dev)[0]) / 2 ** 20   # the DB (one copy, shared by all slots) + every slot's frame buffers
        self.work = [torch.empty((Q, 128), dtype=torch.float32, device=dev) for _ in range(args.depth)]
        if B > 1:
            self.work_b = [torch.empty((Q * B, 128), dtype=torch.float32, device=dev) for _ in range(args.depth)]
        if args.depth_kind and args.moped3d_frontend:
            from moped_amd import moped3d
            if args.depthfill:   # per slot: the B working maps DEPTHFILL fills in place + the distance maps it writes
                self.fill_work = [[(torch.empty((480, 640, 4), dtype=torch.float32, device=dev),
                                    torch.empty((480, 640), dtype=torch.float32, device=dev)) for _ in range(max(B, 1))]
                                  for _ in range(args.depth)]
            table = moped3d.ratio_table(db.xyz, db.model_of, db.n_models, synth.K_DEFAULT)
            for c in self.pipe.ctxs:
                c.frame_set_depth_rules(synth.K_DEFAULT, 64, 0.05, 0.01, table)      # config.hpp:41-44
                c.frame_set_cluster_linkage(capi.default_linkage_params())          # config.hpp:45
        self.active_slots = args.depth   # slots in use (the calibration may settle on fewer)
        # Delivery (the reference's loop hands every frame's objects to its caller, moped2/libmoped/src/moped.cpp:166-194):
        # behind every batch one stream-ordered copy of its B result heads into the slot's pinned host block; the host
        # reads the block -- object counts, the planted models among the objects -- before it reuses the slot.
        self.pipe.attach_delivery(DELIVER_CAP, max(B, 1))
        self.slot_pending = {}           # slot -> pool indices of the frames of the delivery in flight (None: not counted)
        self.slot_last = {}              # sharded: slot -> pool indices of the batch whose objects ride on the slot's NEXT exchange
        self.load_frames(args.n_vis)

    def load_frames(self, n_vis):
        """The workload's frames (`n_vis` planted objects each, seeds seeds_base ..) resident in HBM; the
        pipeline stays.  Nothing may be in flight."""
        import math
        import torch
        from moped_amd import capi, synth
        args, db, dev, B, n_pool = self.args, self.db, self.env["dev"], self.B, self.n_pool
        Q = args.queries
        self.n_vis = n_vis
        self.frames = frames = [synth.make_frame(db, n_vis=n_vis, seed=self.seeds_base + s, Q=Q) for s in range(n_pool)]
        self.pristine = [torch.from_numpy(f.desc).to(dev) for f in frames]
        self.uvs = [torch.from_numpy(f.uv).to(dev) for f in frames]
        self.depths = None
        self.maps = None
        if args.depth_kind and args.moped3d_frontend:
            self.maps = []
            for i, f in enumerate(frames):
                img, fill = synth.depth_image(db, f, seed=i, fill_max=0.3)
                if args.depthfill:   # sensor-like holes (blobs + a dead border), none on a planted keypoint's pixel
                    rng_h = np.random.default_rng([0xD0F1, i])
                    hole = np.zeros((480, 640), bool)
                    yy, xx = np.ogrid[:480, :640]
                    for _ in range(18):
                        cy, cx, r = rng_h.integers(0, 480), rng_h.integers(0, 640), rng_h.integers(6, 45)
                        hole |= (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
                    hole[:, :8] = True
                    rows = np.nonzero((f.src_point >= 0) & ~f.is_outlier)[0]
                    hole[np.clip(f.uv[rows, 1].astype(np.int32), 0, 479), np.clip(f.uv[rows, 0].astype(np.int32), 0, 639)] = False
                    img[hole, 2] = -1.0
                self.maps.append((torch.from_numpy(img).to(dev), torch.from_numpy(fill).to(dev)))
        elif args.depth_kind:
            self.depths = []
            for i, f in enumerate(frames):
                wpts, fill = synth.frame_depth(db, f, seed=i)
                f32 = np.float32
                wgt = (1.0 / (1.0 + (fill / f32(0.1 if args.depth_kind == 1 else 25.0)) ** 2)).astype(f32)  # getCauchyWeight
                d = capi.pack_depth(wpts, wgt)
                self.depths.append(torch.from_numpy(d.view(np.float32).reshape(-1, 4)).to(dev))
        # group_frames[g]: the pool frames of batch g (one frame per "batch" without batches)
        if B > 1:
            self.groups = self.n_frames // B
            self.pool_groups = n_pool // math.gcd(n_pool, B)
            self.group_frames = [np.array([(g * B + f) % n_pool for f in range(B)]) for g in range(self.pool_groups)]
            self.pristine_b = [torch.cat([self.pristine[i] for i in gf]) for gf in self.group_frames]
            self.uv_b = [torch.cat([self.uvs[i] for i in gf]) for gf in self.group_frames]
            self.depths_b = None if self.depths is None else [torch.cat([self.depths[i] for i in gf]) for gf in self.group_frames]
        else:
            self.group_frames = [np.array([i]) for i in range(n_pool)]
        # the inputs were put together on torch's default stream (torch.cat); the slots' streams do not wait for it
        torch.cuda.synchronize(dev)
        self.host_desc = None            # h2d measurement: the same descriptors in pinned host memory
        nv = max(1, max(len(f.visible) for f in frames))
        self.planted = np.full((n_pool, nv), -1, np.int32)
        for i, f in enumerate(frames):
            self.planted[i, :len(f.visible)] = f.visible
        self.reset_delivered()

    # ---- delivered results ----------------------------------------------------------------------------------
    def reset_delivered(self):
        self.dl = {"frames": 0, "objects": 0, "hist": np.zeros(64, np.int64), "planted_missed": 0, "frames_missing": 0}

    def _count(self, which, n_per_frame, models, valid):
        """n_per_frame [B], models / valid [B][cap]: the delivered objects of the pool frames `which` [B]."""
        st = self.dl
        B = len(n_per_frame)
        st["frames"] += B
        st["objects"] += int(n_per_frame.sum())
        st["hist"] += np.bincount(np.minimum(n_per_frame, 63), minlength=64)
        pl = self.planted[which]
        found = ((models[:, None, :] == pl[:, :, None]) & valid[:, None, :]).any(-1) | (pl < 0)
        st["planted_missed"] += int((~found).sum())
        st["frames_missing"] += int((~found.all(-1)).sum())

    def _consume(self, slot, count=True):
        recs = self.pipe.take_delivery(slot)    # waits for the delivery's event only; raises on capacity / exchange flags
        if recs is None:
            return
        first = self.slot_pending.pop(slot, None)
        if first is None or not count:
            return
        n = recs["head"]["n_objects"]
        self._count(first, n, recs["objects"]["model"], np.arange(DELIVER_CAP)[None, :] < n[:, None])

    def _deliver(self, slot, first, tag):
        """Behind the batch just enqueued in `slot` (pool frames `first`, an index array): its delivery.  With a sharded DB what arrives
        is the slot's PREVIOUS batch, all ranks' objects (they rode on this batch's exchange)."""
        pipe = self.pipe
        pipe.deliver(slot, tag)
        if pipe.exchange:
            self.slot_pending[slot] = self.slot_last.get(slot)
            self.slot_last[slot] = first
        else:
            self.slot_pending[slot] = first

    def drain(self, count=True):
        """Every delivery in flight reaches the host; with a sharded DB the slots' last batches too (exchange 2 on its own:
        nothing follows to carry them).  Every rank calls this at the same point."""
        pipe, B = self.pipe, max(self.B, 1)
        for slot in sorted(self.slot_pending):
            self._consume(slot, count)
        if pipe.exchange:
            for slot in sorted(self.slot_last):
                first = self.slot_last.pop(slot)
                per = pipe.flush_objects_batch(slot, B)
                if count:
                    n = np.array([len(o) for o in per], np.int64)
                    models = np.full((B, max(1, int(n.max()))), -1, np.int32)
                    for f, o in enumerate(per):
                        models[f, :len(o)] = o["model"]
                    self._count(first, n, models, models >= 0)

    def delivered_summary(self):
        st = self.dl
        hist = st["hist"]
        top = int(np.nonzero(hist)[0].max()) + 1 if hist.any() else 0
        return {"frames": int(st["frames"]), "objects": int(st["objects"]),
                "objects_per_frame_histogram": hist[:top].tolist(),
                "planted_objects_missed": int(st["planted_missed"]), "frames_missing_a_planted_object": int(st["frames_missing"])}

    # ---- one step -----------------------------------------------------------------------------------------
    def _run_step_batched(self, step, from_host=False):
        import torch
        a, pipe, B = self.args, self.pipe, self.B
        from moped_amd import synth
        for g in range(self.groups):
            slot = (step * self.groups + g) % self.active_slots
            pg = (step * self.groups + g) % self.pool_groups
            self._consume(slot)   # the slot's previous batch is on the host before the slot is reused
            with torch.cuda.stream(pipe.streams[slot]):
                self.work_b[slot].copy_(self.host_desc[pg] if from_host else self.pristine_b[pg], non_blocking=True)
            if self.depths_b is not None:
                pipe.ctxs[slot].frame_set_depth(self.depths_b[pg].data_ptr(), a.depth_kind, 0.5)
            if self.maps is not None:   # the B frames' own depth and distance maps
                mm = [self.maps[i] for i in self.group_frames[pg]]
                if a.depthfill:
                    for j, m in enumerate(mm):
                        wd, wf = self.fill_work[slot][j]
                        with torch.cuda.stream(pipe.streams[slot]):
                            wd.copy_(m[0], non_blocking=True)
                        pipe.ctxs[slot].depth_fill_dev(wd.data_ptr(), 640, 480, synth.K_DEFAULT, wf.data_ptr(), 8)
                    mm = self.fill_work[slot][:B]
                pipe.ctxs[slot].frame_set_depth_image_batch([m[0].data_ptr() for m in mm], [m[1].data_ptr() for m in mm], 640, 480,
                                                            a.depth_kind, 0.5, 0.1 if a.depth_kind == 1 else 25.0)
            pipe.enqueue_batch(slot, self.work_b[slot], self.uv_b[pg], B, [1000 * step + g * B + f + 1 for f in range(B)])
            self._deliver(slot, self.group_frames[pg], step * self.groups + g)

    def run_step(self, step, from_host=False):
        import torch
        if self.B > 1:
            return self._run_step_batched(step, from_host)
        a, pipe = self.args, self.pipe
        from moped_amd import synth
        world = self.env["world"]
        for f in range(self.n_frames):
            b = f % self.n_pool
            slot = f % a.depth
            s = pipe.streams[slot]
            self._consume(slot)
            with torch.cuda.stream(s):
                # fresh raw descriptors (normalise is in place): from HBM (the headline) or over PCIe from pinned memory
                self.work[slot].copy_(self.host_desc[b] if from_host else self.pristine[b], non_blocking=True)
            if self.depths is not None:
                pipe.ctxs[slot].frame_set_depth(self.depths[b].data_ptr(), a.depth_kind, 0.5)
            if self.maps is not None:
                mb = self.maps[b]
                if a.depthfill:
                    mb = self.fill_work[slot][0]
                    with torch.cuda.stream(s):
                        mb[0].copy_(self.maps[b][0], non_blocking=True)
                    pipe.ctxs[slot].depth_fill_dev(mb[0].data_ptr(), 640, 480, synth.K_DEFAULT, mb[1].data_ptr(), 8)
                pipe.ctxs[slot].frame_set_depth_image(mb[0].data_ptr(), mb[1].data_ptr(), 640, 480,
                                                      a.depth_kind, 0.5, 0.1 if a.depth_kind == 1 else 25.0)
            pipe.enqueue(slot, self.work[slot], self.uvs[b], seed=1000 * step + f + 1)
            self._deliver(slot, self.group_frames[b], step * self.n_frames + f)

    def sync_all(self):
        import torch
        import torch.distributed as dist
        self.pipe.synchronize()
        torch.cuda.synchronize(self.env["dev"])
        if self.env["world"] > 1:
            dist.barrier()

    def max_over_ranks(self, seconds):
        import torch
        import torch.distributed as dist
        if self.env["world"] > 1:
            t = torch.tensor([seconds], dtype=torch.float64, device=self.env["red_dev"])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seconds = float(t.item())
        return seconds

    def calibrate_slots(self):
        """Untimed: how many of the slots to use.  One stream per hardware queue (16) is the rule, but on some hosts a
        process gets fewer queues' worth of concurrency and 12 slots run 10-45% faster than 16: eight
        steps with each, all ranks together, the faster setting stays.  Only for the default slot count."""
        timing = {}
        for cand in (16, 12):
            self.active_slots = cand
            self.run_step(-100)
            self.drain(count=False)
            self.sync_all()
            t0c = time.perf_counter()
            for k in range(8):
                self.run_step(-101 - k)
            self.drain(count=False)
            self.sync_all()
            timing[cand] = self.max_over_ranks(time.perf_counter() - t0c)
        self.active_slots = 16 if timing[16] <= timing[12] * 1.03 else 12

    def timed(self, steps, warmup, from_host=False, step_base=0):
        """The contract's timed region: `warmup` untimed steps, then exactly `steps` steps bracketed by a barrier +
        synchronize on both sides, max over ranks.  Inside it every batch's objects are delivered to the host (one
        stream-ordered copy of the batch's result heads into pinned memory; read and counted before the slot's next
        batch) and the clock stops only when the last batch's have arrived.  Returns (seconds, host issue seconds);
        self.dl = what was delivered."""
        for w in range(warmup):
            self.run_step(step_base - 1 - w, from_host=from_host)
        self.drain(count=False)
        self.sync_all()
        self.reset_delivered()
        t0 = time.perf_counter()
        for k in range(steps):
            self.run_step(step_base + k, from_host=from_host)
        t_issue = time.perf_counter() - t0   # the host's share: the enqueue loop alone (behind full queues it waits for the GPU)
        self.drain(count=True)
        self.sync_all()
        dt = self.max_over_ranks(time.perf_counter() - t0)
        return dt, t_issue

    def total_frames(self, steps):
        return steps * self.n_frames * self.R

    def detections_per_frame(self):
        """Objects per frame over EVERY frame of the last timed region, as delivered to the host inside it."""
        st = self.dl
        self.detections_detail = self.delivered_summary()
        return st["objects"] / st["frames"] if st["frames"] else 0.0

    def close(self):
        self.pipe.close()


def measure_pose(job, out_cfg):
    """occupancy and hypotheses/s of the RANSAC kernel, next to every GPU figure.  Live: one isolated
    batch on slot 0 with the library's stage timing (HIP events on the launching stream around every stage)."""
    import torch
    pipe, B, args = job.pipe, job.B, job.args
    c, s = pipe.ctxs[0], pipe.streams[0]
    info = c.pose_kernel_info()
    if info is None:
        return None
    reps = 10
    tot = np.zeros(2)
    hyp = tasks = 0
    for r in range(reps):
        with torch.cuda.stream(s):
            (job.work_b[0] if B > 1 else job.work[0]).copy_(job.pristine_b[0] if B > 1 else job.pristine[0], non_blocking=True)
        c.enable_timing(True)
        if B > 1:
            pipe.enqueue_batch(0, job.work_b[0], job.uv_b[0], B, [7000 + r * B + f for f in range(B)])
        else:
            pipe.enqueue(0, job.work[0], job.uvs[0], seed=7000 + r)
        s.synchronize()
        t = c.timing()
        c.enable_timing(False)
        if r == 0:
            continue   # first timed pass warms the event pool
        tot += np.array([t.get("pose1_ms", 0.0), t.get("pose2_ms", 0.0)])
        k = c.frame_counters()
        hyp += k["hypotheses"]
        tasks += k["pose_tasks"]
    n = reps - 1
    pose_ms = float(tot.sum() / n)            # both POSE launches of one batch
    return {"pose_ms": pose_ms, "hyp": hyp / n, "tasks": tasks / n, "info": info}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))     # nothing above or inside touched the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world                    # n_gpus = the ranks that run, whatever --gpus said
    wd = args.watchdog if args.watchdog >= 0 else (900 if world > 1 else 0)
    arm_watchdog(wd)   # a collective that never completes must end the job with a stack, not hold the node; re-armed per leg
    G, R = choose_partition(args.models, world, args.parallelism, args.grid)
    if args.force_exchange and world == 1:
        G, R = 1, 1
    by_frames = G == 1 and world > 1 and not args.force_exchange
    sharded = G > 1 or args.force_exchange
    depth_given = args.depth > 0
    if args.depth <= 0:
        args.depth = 16
    if args.depth > 4:
        # one HW queue per frame in flight (+ RCCL's); the HIP runtime reads this when it initialises
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(args.depth, 16)))
    # The C++ hosts (moped_hip_bench, the STEP-plugin harness) are measured FIRST, as child processes, while this
    # process has not touched the GPU: a C++ host runs alone in production, and one that shares the chip's hardware
    # queues with this file's 16 streams loses a third of its pinned-host rate (8 300 against 12 400 frames/s).
    host_figs = None
    if rank == 0 and world == 1 and not args.no_secondary and not (args.depth_kind or args.moped3d_frontend):
        from moped_amd import synth as _synth
        _db = _synth.make_db(args.models, 5000)
        _pool = max(1, min(args.frame_pool, args.frames_per_step))
        host_figs = host_side_figures(args, _db, [_synth.make_frame(_db, n_vis=args.n_vis, seed=s, Q=args.queries) for s in range(_pool)])
    import torch
    import torch.distributed as dist
    from moped_amd import capi, synth
    B = default_batch(args, sharded, G)

    # MH_BENCH_REHEARSE=1: the N > 1 code paths on a one-GPU box -- all ranks on cuda:0, gloo for the timing contract's
    # barrier / max (RCCL refuses two ranks on one device); with a sharded DB the frames' exchange then runs over the
    # library's host transport.  The numbers of such a run mean nothing; it is recorded under env_overrides.
    rehearse = os.environ.get("MH_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    red_dev = torch.device("cpu") if rehearse else dev
    if world > 1:
        # torch.distributed: the barrier / max-over-ranks of the timing contract and the hand-over of rank 0's
        # communicator id; the frames' collectives are issued by libmoped_hip.so itself (csrc/comm.hip)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    _subgroups = {}

    def subgroups(g):
        """The gloo subgroups of a G = g grid (rehearsal: the host transport's all-gather runs over them), made once:
        every rank creates every group, in the same order."""
        if g not in _subgroups:
            _subgroups[g] = [dist.new_group(list(range(r * g, (r + 1) * g)), backend="gloo") for r in range(world // g)]
        return _subgroups[g]
    env = {"rank": rank, "local_rank": local_rank, "world": world, "dev": dev, "red_dev": red_dev, "rehearse": rehearse,
           "subgroups": subgroups}

    Q = args.queries
    db = synth.make_db(args.models, 5000)
    job = Job(args, env, db, args.models, by_frames, sharded, B, args.frames_per_step, part=(G, R))
    pipe, params = job.pipe, job.params
    if B > 1 and not depth_given and args.depth == 16:
        job.calibrate_slots()
    arm_watchdog(wd + 2 * args.steps if wd > 0 else 0)   # (wd = 0 is "off", not "2 s per step": the N = 1 default)
    dt, t_issue = job.timed(args.steps, args.warmup)
    fps = job.total_frames(args.steps) / dt    # every one of these frames' objects reached the host inside dt (checked below)
    det_per_frame = job.detections_per_frame()
    delivered = job.detections_detail

    # what POSE actually evaluated (device-side counters of the last frame of every slot)
    ctr = [c.frame_counters() for c in pipe.ctxs[:job.active_slots if B > 1 else args.depth]]
    R_ = params.pose1.max_objects_per_cluster
    hyp_per_task = float(np.sum([c["hypotheses"] for c in ctr]) / max(1, np.sum([c["pose_tasks"] for c in ctr])))
    ms = pipe.ctxs[0].match_stats(Q * B)
    overrides = {k: v for k, v in sorted(os.environ.items()) if k.startswith("MH_") or k == "GPU_MAX_HW_QUEUES"}
    comm_info = pipe.comm_info() if pipe.exchange else None
    n_frames, n_pool = job.n_frames, job.n_pool

    out = {
        "metric": "detections/sec (frames/s) 640x480 ~3k SIFT vs N models",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": scaling_label(G, R), "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "dtype_detail": "every delivered result is fp32 arithmetic (bit-identical to the exact f32 kernels and the oracle); "
                        "the screen that decides which rows get it multiplies in f16 on the matrix pipe with a proved margin",
        "config": {"workload": f"{args.models}-model DB ({db.n} descriptors), 640x480 frames, "
                               f"{Q} SIFT-like keypoints, {args.n_vis} planted objects, "
                               f"up to {abs(params.pose1.n_hypotheses)} P3P hypotheses x {R_} replicas per cluster "
                               f"({'all evaluated' if args.no_adaptive else 'adaptive stop after 256'}: {hyp_per_task:.0f} evaluated per "
                               f"(cluster, replica) task), MATCH->CLUSTER->POSE->FILTER->POSE2->FILTER2"
                               + ("" if not args.depth_kind else f", moped3d depth residuals kind {args.depth_kind}")
                               + ("" if not (args.depth_kind and args.moped3d_frontend) else
                                  ", moped3d front end on the device (" + ("DEPTHFILL of a map with holes, " if args.depthfill else "") +
                                  "DEPTHFILTER x2, adaptive ratio, DEPTHMAP_PROP, CLUSTER_LINKAGE)"),
                   "frames_per_step": n_frames, "distinct_frames": n_pool, "timed_seconds": round(dt, 3),
                   "host_issue_seconds": round(t_issue, 3),
                   "frames_in_flight": (job.active_slots if B > 1 else args.depth) * B, "frames_per_match_launch": B,
                   "lane": args.lane or None,
                   "parallelism": partition_label(G, R),
                   "partition": {"model_shards": G, "frame_groups": R,
                                 "models_per_rank": -(-args.models // G),
                                 "chosen_by": (args.parallelism if args.parallelism != "auto" else
                                               "auto: largest G whose projection from profiles/per_rank_load.json reaches 0.75 N, "
                                               "else the best G >= 2"),
                                 "projected_speedup": (None if world == 1 or not os.path.exists(PER_RANK_LOAD) else
                                                       round(projected_speedup(args.models, G, R), 2)),
                                 "communicators": (None if G == 1 else
                                                   f"{R} disjoint sets of {args.comms if not rehearse else 1} communicator(s), one set per "
                                                   f"frame group, each over the group's {G} ranks (ranks r G .. r G + G - 1)")},
                   "model_assignment": (args.assign if sharded else None),
                   "ranks_launched_by": os.environ.get("MH_BENCH_LAUNCHED_BY", "torch.distributed.run" if world > 1 else "none"),
                   "exchange": comm_info,
                   "results_delivered": "every frame",
                   "results_delivery": (f"one stream-ordered copy of each batch's result heads ({B if B > 1 else 1} frames x up to {DELIVER_CAP} objects) "
                                        "into pinned host memory behind the batch (mh_frame_fetch_batch_async"
                                        + (" / mh_frame_fetch_previous_async: all ranks' objects ride on the slot's next exchange; the "
                                           "slots' last batches by mh_frame_gather_objects" if pipe.exchange else "")
                                        + "), read and counted on the host before the slot's next batch; the clock stops after the last one"),
                   "objects_per_frame": det_per_frame, "objects_detail": delivered,
                   "hypotheses_per_task": round(hyp_per_task, 1), "hypotheses_per_frame": round(float(np.mean([c["hypotheses"] for c in ctr])), 1),
                   "match": ("two-stage: f16 MFMA screen + canonical f32 arithmetic on the candidates (bit-identical to the exact kernels)"
                             if ms["two_stage"] else "exact f32 kernels"),
                   "hbm_pipeline_mb": round(job.hbm_pipeline_mb, 1),
                   "env_overrides": overrides},
    }
    # sanity of the line itself: a pipeline that stops finding the planted objects must not print a clean metric
    if delivered["frames"] != args.steps * n_frames:
        out["suspect"] = f"{delivered['frames']} frames were delivered to the host, {args.steps * n_frames} were timed"
    elif delivered["frames_missing_a_planted_object"]:
        out["suspect"] = (f"{delivered['frames_missing_a_planted_object']} of {delivered['frames']} delivered frames miss a planted "
                          f"object ({delivered['planted_objects_missed']} objects)")
    elif det_per_frame < args.n_vis - 0.5:
        out["suspect"] = f"only {det_per_frame:.2f} objects per frame found of {args.n_vis} planted"

    # ---- secondary: the same frames with the descriptors in pinned HOST memory (1.5 MB over PCIe per frame, the copy
    # on the frame's own stream, overlapped with the other frames in flight).  Never `value`.
    if world == 1 and args.h2d_steps > 0:
        if B > 1:
            host_desc = [torch.from_numpy(np.concatenate([job.frames[i].desc for i in gf])).pin_memory() for gf in job.group_frames]
        else:
            host_desc = [torch.from_numpy(f.desc).pin_memory() for f in job.frames]
        job.host_desc = host_desc
        dth, _ = job.timed(args.h2d_steps, 1, from_host=True, step_base=-600)
        out["h2d_inclusive"] = {"value": round(args.h2d_steps * n_frames / dth, 2), "unit": "frames/s", "steps": args.h2d_steps,
                                "objects_per_frame": job.detections_per_frame(),
                                "note": "descriptors start in pinned host memory: one 1.5 MB hipMemcpyAsync per frame on the frame's "
                                        "stream (keypoint coordinates resident), objects delivered to the host like the headline's; "
                                        "a reported figure, not the metric's `value`"}

    # ---- roofline of the dominant kernel, measured live with HIP events on the stream it is launched on ----
    arm_watchdog(wd)
    if rank == 0 and not args.no_roofline:
        out["roofline"] = roofline(job, fps, out)
    if world > 1:
        dist.barrier()

    # ---- POSE: occupancy, hypotheses/s, share of the chip's CU time ----
    # (N = 1 only: with a sharded DB a batch on rank 0's pipeline is a collective the other ranks would have to join)
    if rank == 0 and world == 1 and not args.no_roofline and "roofline" in out and not (args.depth_kind or args.moped3d_frontend):
        try:
            pm = measure_pose(job, out["config"])
        except Exception as e:   # a reported extra, never a reason to lose the line
            pm = None
            out["roofline"]["pose"] = {"error": str(e)[:200]}
        if pm:
            info = pm["info"]
            frames_per_batch = B
            cu_s_per_frame = pm["pose_ms"] * 1e-3 / frames_per_batch   # wall time of the two launches per frame, isolated
            out["roofline"]["pose"] = {
                "kernel": "pose_kernel<0> (POSE and POSE2 of one batch, isolated, HIP events by the library; stage timing keeps the "
                          "steps apart, so this is the one-launch form -- the pipeline runs hypotheses and refines as two launches)",
                "vgprs": info["vgprs"], "lds_bytes": info["lds_bytes"], "threads_per_workgroup": info["threads"],
                "waves_per_simd": info["waves_per_simd"], "workgroups_per_cu": info["workgroups_per_cu"],
                "tasks_per_frame": round(pm["tasks"], 1), "hypotheses_per_frame": round(pm["hyp"], 1),
                "ms_per_batch_both_stages": round(pm["pose_ms"], 4),
                "hypotheses_per_s_isolated": round(pm["hyp"] * frames_per_batch / (pm["pose_ms"] * 1e-3), 0) if pm["pose_ms"] > 0 else None,
                "hypotheses_per_s_pipeline": round(float(np.mean([c["hypotheses"] for c in ctr])) * fps, 0),
                # POSE's share of the chip's CU time in the pipeline, an upper bound: every task charged the whole isolated
                # launch of its stage (half the two launches' time) on the half compute unit a POSE workgroup occupies,
                # over the CU time the chip has in the period the pipeline takes for one batch
                "cu_time_share": round((pm["tasks"] * frames_per_batch * (pm["pose_ms"] / 2.0) / max(info["workgroups_per_cu"], 1))
                                       / (N_CU * (frames_per_batch / fps * 1e3)), 4) if fps > 0 else None,
                "cu_time_share_note": "upper bound: (tasks of a frame x frames of a batch) x the isolated launch time of a stage x "
                                      "1/workgroups_per_cu of a compute unit, over 256 CUs x the pipeline's time per batch",
            }
    if world > 1:
        dist.barrier()

    # ---- the workload breadth SURVEY 8(d) defines, same DB, same pipeline, same launch shape (all ranks take part):
    # 5 and 10 visible objects per frame (POSE's cost grows with what is visible,
    # POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:188-211), and every one of the 4 x 1024 hypotheses evaluated ----
    if not args.no_secondary and not (args.depth_kind or args.moped3d_frontend) and not args.no_adaptive:
        sec_steps = max(1, args.secondary_steps)

        def side(n_vis, no_adaptive):
            if n_vis != job.n_vis:
                job.load_frames(n_vis)
            h1, h2 = params.pose1.n_hypotheses, params.pose2.n_hypotheses
            if no_adaptive:
                params.pose1.n_hypotheses, params.pose2.n_hypotheses = -abs(h1), -abs(h2)
            try:
                dts, _ = job.timed(sec_steps, 1, step_base=-2000)
            finally:
                params.pose1.n_hypotheses, params.pose2.n_hypotheses = h1, h2
            det = job.detections_per_frame()
            cs = [c.frame_counters() for c in pipe.ctxs[:job.active_slots if B > 1 else args.depth]]
            r = {"value": round(job.total_frames(sec_steps) / dts, 2), "unit": "frames/s", "steps": sec_steps,
                 "planted_objects": n_vis, "objects_per_frame": round(det, 3), "objects_detail": job.detections_detail,
                 "hypotheses_per_task": round(float(np.sum([c["hypotheses"] for c in cs]) / max(1, np.sum([c["pose_tasks"] for c in cs]))), 1),
                 "pose_tasks_per_frame": round(float(np.mean([c["pose_tasks"] for c in cs])), 1)}
            d = job.detections_detail
            if d["frames"] != sec_steps * job.n_frames or d["frames_missing_a_planted_object"]:
                r["suspect"] = (f"{d['frames']} of {sec_steps * job.n_frames} frames delivered, "
                                f"{d['frames_missing_a_planted_object']} miss a planted object")
            return r
        for label, nv, na in (("no_adaptive", args.n_vis, True), ("n_vis_5", 5, False), ("n_vis_10", 10, False)):
            arm_watchdog(wd)
            try:
                out[label] = side(nv, na)
            except Exception as e:    # a reported extra: the line survives, the failure is in it
                if world > 1:
                    raise
                out[label] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out["no_adaptive"]["note"] = "all 1024 hypotheses of every (cluster, replica) task evaluated (bench.py --no-adaptive as a whole line)"
        if job.n_vis != args.n_vis:
            job.load_frames(args.n_vis)

    # ---- image in, objects out (the next row in front of the path: FEAT on the device) ----
    if world == 1 and not args.no_secondary and not (args.depth_kind or args.moped3d_frontend) and not args.no_adaptive:
        arm_watchdog(wd)
        try:
            out["image_to_objects"] = image_to_objects_leg(args)
            arm_watchdog(wd)
            t = image_to_objects_leg(args, frames=1024, image="textured")   # the config's keypoint count
            out["image_to_objects"]["at_3240_keypoints"] = {k: t[k] for k in ("value", "unit", "frames", "keypoints_per_image", "sift_alone_ms",
                                                                               "objects_per_frame_last_batch", "image")}
        except Exception as e:    # a reported extra: the line survives, the failure is in it
            out.setdefault("image_to_objects", {})["error"] = f"{type(e).__name__}: {e}"[:300]

    # ---- secondary partitions / workloads in the same line (all ranks take part) ----
    if not args.no_secondary and not (args.depth_kind or args.moped3d_frontend):
        sec_steps = max(1, args.secondary_steps)
        if world > 1 and G > 1 and R > 1:
            # the north star's partition as it is written -- every rank a shard, ONE frame stream -- beside the grid
            job.close()
            arm_watchdog(wd)
            a1 = argparse.Namespace(**vars(args))
            a1.parallelism = "models"
            j1 = Job(a1, env, db, args.models, False, True, default_batch(a1, True, world), args.frames_per_step, part=(world, 1))
            dt1, _ = j1.timed(sec_steps, 1)
            det1 = j1.detections_per_frame()
            out["pure_model_shard"] = {"value": round(j1.total_frames(sec_steps) / dt1, 2), "unit": "frames/s", "steps": sec_steps,
                                       "parallelism": partition_label(world, 1), "scaling": "strong",
                                       "models_per_rank": -(-args.models // world), "objects_per_frame": det1,
                                       "objects_detail": j1.detections_detail,
                                       "note": "north_star's partition with no frame groups: the DB sharded over all ranks, one "
                                               "all-gather per batch over all of them; a reported figure, not `value`"}
            j1.close()
            job = None
        if world > 1 and not by_frames:
            if job is not None:
                job.close()
            arm_watchdog(wd)
            a2 = argparse.Namespace(**vars(args))
            a2.parallelism = "frames"
            j2 = Job(a2, env, db, args.models, True, False, default_batch(a2, False), args.frames_per_step, part=(1, world))
            dt2, _ = j2.timed(sec_steps, 1)
            det2 = j2.detections_per_frame()
            out["replicated_frames"] = {"value": round(j2.total_frames(sec_steps) / dt2, 2), "unit": "frames/s", "steps": sec_steps,
                                        "parallelism": f"frame-parallel x{world} (DB replicated)", "scaling": "weak",
                                        "objects_per_frame": det2, "objects_detail": j2.detections_detail,
                                        "note": "SURVEY 8(e)'s alternative for a DB too small to shard: no collective, every rank "
                                                "its own frames; a reported figure, not `value`"}
            j2.close()
            job = None
        if args.models != 200:
            if job is not None:
                job.close()
                job = None
            arm_watchdog(wd)
            a3 = argparse.Namespace(**vars(args))
            a3.models, a3.parallelism = 200, "models"
            db200 = synth.make_db(200, 5000)
            sh3 = world > 1
            j3 = Job(a3, env, db200, 200, False, sh3, default_batch(a3, sh3, world), 256, part=(world, 1))
            dt3, _ = j3.timed(sec_steps, 1)
            det3 = j3.detections_per_frame()
            out["sharded_200_models"] = {"value": round(j3.total_frames(sec_steps) / dt3, 2), "unit": "frames/s", "steps": sec_steps,
                                         "frames_per_step": j3.n_frames, "frames_per_match_launch": j3.B,
                                         "parallelism": f"model-shard x{world}" if world > 1 else "single GPU",
                                         "models_per_rank": 200 // world, "objects_per_frame": det3,
                                         "note": "BASELINE configs[2] (N = 1) / configs[3] (N > 1): the 200-model DB, 1 M "
                                                 "descriptors, sharded by model; a reported figure, not `value`"}
            j3.close()

    if host_figs:
        out.update(host_figs)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if job is None:
            frames_cpu = [synth.make_frame(db, n_vis=args.n_vis, seed=s, Q=Q) for s in range(n_pool)]
        else:
            frames_cpu = job.frames
        out["cpu_baseline"] = cb = cpu_baseline(db, frames_cpu, args)
        # the GPU figures of this line over the CPU figure of the same box (north_star: >= 50x at 1 GPU).  A reported
        # ratio, not a kernel-quality claim: the roofline fraction is that.
        if cb.get("value"):
            ratios = {"value": out["value"]}
            for key, src in (("h2d_inclusive", out.get("h2d_inclusive", {}).get("value")), ("cpp_host_pinned", (out.get("cpp_host") or {}).get("fps_pinned_host")),
                             ("plugin_resident", out.get("plugin_resident_fps")), ("plugin_path_six_steps", out.get("plugin_path_fps"))):
                if src:
                    ratios[key] = src
            out["vs_cpu_baseline"] = {k: round(v / cb["value"], 1) for k, v in ratios.items()}
            out["vs_cpu_baseline"]["against"] = "cpu_baseline.value (whole sample); the median frame gives x%.2f of these" % (cb["value"] / cb["median_frame"])
    if rank == 0:
        print(json.dumps(out), flush=True)
    if job is not None:
        job.close()
    if world > 1:
        dist.destroy_process_group()


def image_to_objects_leg(args, frames=2048, slots=16, batch=16, image="bundled"):
    """SURVEY 8(f) N2 in front of the path: 8-bit 640x480 frames in, objects out -- FEAT (SIFT, csrc/sift.hip: what
    FEAT_SIFT_CPU::process does, moped2/libmoped/src/feat/FEAT_SIFT_CPU.hpp:78-112), MATCH .. FILTER2 on the device,
    `batch` images per launch sequence (mh_frame_enqueue_image_batch), `slots` contexts in flight.  The frames are the
    reference's bundled test frames (tests/golden/sift_ref_frames.npz, ~590 keypoints each); the DB is the synthetic
    one plus frame 0's own keypoints as a planar model, so every frame of the pool that shows it yields an object.
    Images resident in HBM, like the descriptors of the judged line.  image="textured": synth.textured_image -- ~3 240
    keypoints, the keypoint count BASELINE.json's config names -- with the keypoints of its centre as the planar model (all
    3 240 in one model are past what CLUSTER / POSE reserve per model)."""
    import torch
    from moped_amd import capi, synth
    gold = np.load(os.path.join(ROOT, "tests", "golden", "sift_ref_frames.npz"))
    K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY
    dev = torch.device("cuda:0")
    db = synth.make_db(args.models, 5000)
    c0 = capi.Context(0)
    textured = image == "textured"
    pool = [synth.textured_image(0)] if textured else [gold[f"gray{int(f)}"] for f in gold["frames"]]
    cap = 4096 if textured else 1024
    xy, _, desc = c0.sift(pool[0])
    if textured:
        hh, ww = pool[0].shape
        keep = (np.abs(xy[:, 0] - ww / 2) < 110) & (np.abs(xy[:, 1] - hh / 2) < 90)
        xy, desc = xy[keep], desc[keep]
    z = np.float32(0.8)
    xyz = np.stack([(xy[:, 0] - K[2]) / K[0] * z, (xy[:, 1] - K[3]) / K[1] * z, np.full(len(xy), z)], 1).astype(np.float32)
    all_desc = c0.normalize(np.concatenate([db.desc, desc]))
    all_xyz = np.concatenate([db.xyz, xyz])
    model_of = np.concatenate([db.model_of, np.full(len(xy), args.models, np.int32)])
    c0.close()
    ctxs, streams = [], []
    try:
        for i in range(slots):
            c = capi.Context(0)
            st = torch.cuda.Stream(device=dev)
            c.set_stream(st.cuda_stream)
            if i == 0:
                c.db_upload(all_desc, model_of, all_xyz, args.models + 1)
            else:
                c.db_share(ctxs[0])
            c.reserve(cap * batch)
            ctxs.append(c)
            streams.append(st)
        imgs = [torch.from_numpy(np.ascontiguousarray(g)).to(dev) for g in pool]
        h, w = pool[0].shape
        prm = capi.default_frame_params()
        cam = capi.make_cam(K, CAM0)
        torch.cuda.synchronize()

        def go(k):
            for g in range(k // batch):
                ptrs = [imgs[(g * batch + j) % len(imgs)].data_ptr() for j in range(batch)]
                ctxs[g % slots].frame_enqueue_image_batch(ptrs, w, h, True, cap, K, CAM0, prm,
                                                         [g * batch + j + 1 for j in range(batch)], _cam_struct=cam)
        go(2 * slots * batch)
        for c in ctxs:
            c.frame_fetch_slot(0)
        frames = max(slots * batch, frames // (slots * batch) * slots * batch)
        t0 = time.perf_counter()
        go(frames)
        for st in streams:
            st.synchronize()
        dt = time.perf_counter() - t0
        n_obj, best = [], []
        for j in range(batch):    # the last batch's frames: frame 0's object where frame 0 is the image
            objs, _ = ctxs[(frames // batch - 1) % slots].frame_fetch_slot(j)
            n_obj.append(len(objs))
            best.append(int(objs[np.argmax(objs["score"])]["model"]) if len(objs) else -1)
        # SIFT alone, one image at a time
        c = ctxs[0]
        cap = 4096
        d_, xy_, cnt = (torch.empty((cap, 128), device=dev), torch.empty((cap, 2), device=dev),
                        torch.zeros(1, dtype=torch.int32, device=dev))
        for _ in range(20):
            c.sift_dev(imgs[0].data_ptr(), w, h, True, d_.data_ptr(), xy_.data_ptr(), 0, cap, cnt.data_ptr())
        streams[0].synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            c.sift_dev(imgs[0].data_ptr(), w, h, True, d_.data_ptr(), xy_.data_ptr(), 0, cap, cnt.data_ptr())
        streams[0].synchronize()
        sift_ms = (time.perf_counter() - t0) / 200 * 1e3
        return {"value": round(frames / dt, 1), "unit": "frames/s", "frames": frames, "slots": slots,
                "images_per_launch_sequence": batch,
                "image": "640x480 8-bit, doubled (ScaleOrigin -1): " + ("synth.textured_image" if textured else "the reference's bundled test frames"),
                "keypoints_per_image": int(cnt.item()), "objects_per_frame_last_batch": n_obj,
                "frames_of_the_last_batch_whose_best_object_is_the_planted_model": int(sum(b == args.models for b in best)),
                "sift_alone_ms": round(sift_ms, 4),
                "what": "FEAT(SIFT) + MATCH + CLUSTER + POSE + FILTER + POSE2 + FILTER2 on the device from the image; "
                        "DB = the synthetic models + the first image's keypoints (the textured image: those of its centre) as a planar model"}
    finally:
        for c in ctxs[1:] + ctxs[:1]:
            c.close()


def host_side_figures(args, db, frames):
    """Figures of the C++ hosts (no Python, no torch in the measured process; each a child process on a scene file
    written here): `moped_hip_bench` -- the streaming C++ host that drives mh_frame_enqueue_batch the way this file
    does, descriptors from pinned host memory -- and the literal drop-in, one synchronous frame at a time through the
    STEP plugins with host FrameData between the steps (`moped_hip_test`: the loop of
    moped2/libmoped/src/moped.cpp:183-191).  Absent binaries are reported as such."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_scene
    out = {}
    host = os.path.join(ROOT, "moped_amd", "host")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(host, "moped_hip_bench")
        if os.path.exists(exe):
            try:
                path = os.path.join(tmp, "frames.bin")
                dump_scene.dump_frames(path, db, frames)
                cmd = [exe, path, "--json", "--steps", "5", "--batch", str(default_batch(args, False))]
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode == 0 and lines:
                    d = json.loads(lines[-1])
                    out["cpp_host"] = d
                    out["single_frame_latency_ms"] = d["single_frame_latency_ms"]   # descriptors from pinned host memory: PCIe inside the clock
                    if "single_frame_latency_resident_ms" in d:
                        out["single_frame_latency_resident_ms"] = d["single_frame_latency_resident_ms"]   # descriptors in HBM, like `value`
                else:
                    out["cpp_host"] = {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as e:
                out["cpp_host"] = {"error": str(e)[:300]}
        else:
            out["cpp_host"] = {"error": "moped_amd/host/moped_hip_bench not built"}
        exe = os.path.join(host, "moped_hip_test")
        if os.path.exists(exe):
            try:
                path = os.path.join(tmp, "scene.bin")
                dump_scene.dump(path, db, frames[0])
                r = subprocess.run([exe, path, "30"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
                times = {l.split()[1]: float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("TIME ")}
                n_obj = sum(1 for l in r.stdout.splitlines() if l.startswith("OBJ "))
                if r.returncode == 0 and times:
                    tot = sum(times.values())
                    out["plugin_path_fps"] = round(1.0 / tot, 1)
                    out["plugin_path"] = {"ms_per_frame": round(1e3 * tot, 4), "objects": n_obj,
                                          "steps_ms": {k: round(1e3 * v, 4) for k, v in times.items()},
                                          "note": "moped_hip_test: MopedPipeline -> six STEP plugins -> C ABI, ONE synchronous frame at "
                                                  "a time, features in pageable host FrameData, every step's outputs written into "
                                                  "FrameData; consecutive HIP steps hand the frame over on the device (mh_step_*: a step "
                                                  "that finds FrameData as the HIP step before left it does not upload it again)"}
                else:
                    out["plugin_path"] = {"error": (r.stderr or r.stdout)[-300:]}
                # ... and with the device-side hand-over between the steps off: every step uploads its inputs (round 4's path)
                r = subprocess.run([exe, path, "30"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(env, MH_STEP_HANDOVER="0"))
                times = {l.split()[1]: float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("TIME ")}
                if r.returncode == 0 and times and "plugin_path" in out and "error" not in out["plugin_path"]:
                    out["plugin_path"]["upload_paths"] = {"fps": round(1.0 / sum(times.values()), 1),
                                                          "steps_ms": {k: round(1e3 * v, 4) for k, v in times.items()},
                                                          "note": "MH_STEP_HANDOVER=0: every step carries its inputs over PCIe again"}
                # the same frame through ONE step (FRAME_RESIDENT_HIP -> mh_frame_run_host): the frame stays on the device
                # between MATCH and FILTER2
                r = subprocess.run([exe, "--resident", path, "30"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
                times = {l.split()[1]: float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("TIME ")}
                n_obj = sum(1 for l in r.stdout.splitlines() if l.startswith("OBJ "))
                if r.returncode == 0 and times:
                    tot = sum(times.values())
                    out["plugin_resident_fps"] = round(1.0 / tot, 1)
                    out["plugin_resident"] = {"ms_per_frame": round(1e3 * tot, 4), "objects": n_obj,
                                              "note": "moped_hip_test --resident: MopedPipeline with FRAME_RESIDENT_HIP as its one "
                                                      "step (MATCH_SIFT .. FILTER2 in one call of the C ABI), one synchronous "
                                                      "frame at a time, features in pageable host memory, objects back on the host"}
            except Exception as e:
                out["plugin_path"] = {"error": str(e)[:300]}
        else:
            out["plugin_path"] = {"error": "moped_amd/host/moped_hip_test not built"}
    return out


def sustained_mfma(ach_tf):
    """achieved / what a register-only loop of pass B's MFMA sustains on THIS box right now (a child process: our own
    streams are idle while it runs)."""
    tool = os.path.join(ROOT, "moped_amd", "host", "mfma_rate")
    tf, src = SUSTAINED_F16_TFLOPS, "profiles/r03_mfma_shapes_rate.txt (typical; moped_amd/host/mfma_rate not built)"
    shapes = None
    if os.environ.get("MH_BENCH_REHEARSE") == "1" and int(os.environ.get("WORLD_SIZE", "1")) >= 4:
        # four ranks rehearsing on ONE GPU + their launcher + a test runner that has used the GPU are the six processes a
        # GPU box lets hold its card: no child process on top of them (the figure below is then the typical one)
        src = "profiles/r03_mfma_shapes_rate.txt (typical; no child process in a rehearsal of four ranks on one GPU)"
    elif os.path.exists(tool):
        try:
            res = subprocess.run([tool, "--json"], capture_output=True, text=True, timeout=60)
            shapes = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
            tf, src = float(shapes["f16_16x16x32_tflops"]), "moped_amd/host/mfma_rate --json, this box, after the timed region"
        except Exception as e:   # the figure is a secondary one: never fail the line for it
            src += f" ({type(e).__name__})"
    out = {"tflops": round(tf, 1), "frac": round(ach_tf / tf, 4), "source": src,
           "what": "v_mfma_f32_16x16x32_f16 on random operands out of registers, two wavefronts per SIMD, all CUs"}
    if shapes:
        out["f16_32x32x16_tflops"] = shapes["f16_32x32x16_tflops"]
    return out


def roofline(job, fps, out):
    import torch
    args, pipe, B = job.args, job.pipe, job.B
    dev = job.env["dev"]
    Q = args.queries
    c, s = pipe.ctxs[0], pipe.streams[0]
    Qr = Q * B
    qn = (job.pristine_b[0] if B > 1 else job.pristine[0]).clone()
    qnorm = torch.empty(Qr, dtype=torch.float32, device=dev)
    idx = torch.empty(Qr, dtype=torch.int32, device=dev)
    d1 = torch.empty(Qr, dtype=torch.float32, device=dev)
    d2 = torch.empty(Qr, dtype=torch.float32, device=dev)
    n_local = job.shard.desc.shape[0]
    flops = 2.0 * 128 * Qr * n_local                # one multiply-add per (query, row, coordinate)
    reps = 20
    two_stage = c.match_stats(Qr)["two_stage"]
    c.enable_timing(True)
    stage = np.zeros(5)
    with torch.cuda.stream(s):
        c.normalize_dev(qn.data_ptr(), qnorm.data_ptr(), Qr)
        for _ in range(3):
            c.match_local_dev(qn.data_ptr(), qnorm.data_ptr(), Qr, idx.data_ptr(), d1.data_ptr(), d2.data_ptr())
    s.synchronize()
    c.match_stats(reset=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):   # 20 launch sequences back to back; the library records its own events around every kernel
        e0.record(s)
        for _ in range(reps):
            c.match_local_dev(qn.data_ptr(), qnorm.data_ptr(), Qr, idx.data_ptr(), d1.data_ptr(), d2.data_ptr())
        e1.record(s)
    s.synchronize()
    t_stage = e0.elapsed_time(e1) / reps
    if two_stage:
        stage = np.array(list(c.match_timing().values()))
    c.enable_timing(False)
    st = c.match_stats()
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{'screen_b' if two_stage else 'match'}_{args.models}m_{Qr}q")
        except Exception:
            traffic = None
    if two_stage:
        t_b = float(stage[3])
        ach_tf = flops / (t_b * 1e-3) / 1e12
        rec_bytes = 8.0 * st["candidates"] / max(st["queries"], 1) * Qr / 1.3   # ~1.3 rows per record
        b_alg = 256.0 * n_local + 256.0 * Qr + rec_bytes      # f16 DB once + f16 queries once + candidate records
        return {
            "kernel": "screen16_kernel<1, 4> = pass B of the two-stage MATCH: f16 x f16 -> f32 (v_mfma_f32_16x16x32_f16) over "
                      "every (query, row) pair",
            "bound": "mfma", "achieved": round(ach_tf, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach_tf / PEAK_F16_TFLOPS, 4), "traffic": traffic,
            "ms_per_launch": round(t_b, 4),
            "algorithmic_flops": int(flops),
            "note": "compute bound (SURVEY F10): 2*128*Q*N flops per launch against the dense F16 MFMA peak of the dtype the "
                    "kernel multiplies in.  The match STAGE delivers exact f32 results (bit-identical to the f32 kernels and the "
                    "oracle); its five kernels per launch, ms: "
                    + ", ".join(f"{k} {v:.4f}" for k, v in zip(("query image", "pass A", "thresholds", "pass B", "pass C"), stage)),
            "match_stage": {"ms_per_launch": round(t_stage, 4), "kernels_ms": [round(float(v), 4) for v in stage],
                            "candidate_rows_per_query": round(st["candidates"] / max(st["queries"], 1), 2),
                            "brute_force_queries": st["brute_force_queries"],
                            "f32_equivalent_tflops": round(flops / (t_stage * 1e-3) / 1e12, 1)},
            "measured": "HIP events recorded by libmoped_hip.so on the launching stream around every kernel of the stage "
                        "(mh_match_timing), 20 launch sequences after the timed region, one kernel on the chip at a time; "
                        "profiles/r03_*_depth1_kernel_stats.* is the same command under rocprofv3 with --depth 1",
            "whole_pipeline_tflops_per_gpu": round(flops / B * fps / 1e12, 2),   # F_alg of this rank's shard x frames/s
            # `peak` is the data sheet's 2.5 PFLOP/s (2.4 GHz); a loop of nothing but this MFMA out of registers on every
            # SIMD sustains 74-80% of it (the clock the power budget allows under matrix load), measured on this box:
            "sustained_mfma_only": sustained_mfma(ach_tf),
            "hbm": {"achieved": round(b_alg / (t_b * 1e-3) / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(b_alg / (t_b * 1e-3) / 1e9 / PEAK_HBM_GBS, 5), "algorithmic_bytes": int(b_alg)},
        }
    ach_tf = flops / (t_stage * 1e-3) / 1e12
    b_alg = 512.0 * n_local + 512.0 * Qr + 12.0 * Qr    # DB once + queries once + (idx,d1,d2)
    return {
        "kernel": "match_mfma_kernel / match_kernel (+ combine_splits_kernel, <1% of the time): the exact f32 search",
        "bound": "mfma", "achieved": round(ach_tf, 2), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
        "frac": round(ach_tf / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
        "ms_per_launch": round(t_stage, 4),
        "note": "compute bound (SURVEY F10): 2*128*Q*N fp32 FMA flops vs the 157.3 TFLOP/s dense FP32 MFMA peak",
        "measured": "HIP events around 20 launches on one stream after the timed region (one kernel on the chip at a time)",
        "whole_pipeline_tflops_per_gpu": round(flops / B * fps / 1e12, 2),
        "hbm": {"achieved": round(b_alg / (t_stage * 1e-3) / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(b_alg / (t_stage * 1e-3) / 1e9 / PEAK_HBM_GBS, 5), "algorithmic_bytes": int(b_alg)},
    }


if __name__ == "__main__":
    main()
