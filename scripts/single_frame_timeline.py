"""One frame at a time (the reference's calling convention, moped2/libmoped/src/moped.cpp:183-191): the kernels of an
isolated frame, in launch order, with their start offsets, durations and the gaps between them.

  on the GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/sft -- python3 $REPO/scripts/single_frame_timeline.py run [models] [n_vis]
                   python3 $REPO/scripts/single_frame_timeline.py report /tmp/sft
`run` also prints the wall latency (enqueue -> objects on the host) without the profiler's help."""
import csv, glob, os, sys, time
import numpy as np

def run(models=20, n_vis=2, frames=80):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from moped_amd import synth
    from moped_amd.pipeline import FramePipeline, ShardedDB
    db = synth.make_db(models, 5000)
    frs = [synth.make_frame(db, n_vis=n_vis, seed=s) for s in range(8)]
    dev = torch.device("cuda:0")
    from moped_amd import capi
    prm = capi.default_frame_params()
    if "TIMELINE_L2" in os.environ:   # experiment: cap of POSE's plain-residual phase
        prm.pose1.lm_iters_l2 = prm.pose2.lm_iters_l2 = int(os.environ["TIMELINE_L2"])
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000, params=prm)
    q = [torch.from_numpy(f.desc).to(dev) for f in frs]
    uv = [torch.from_numpy(f.uv).to(dev) for f in frs]
    work = torch.empty_like(q[0])
    torch.cuda.synchronize()
    lat = []
    for i in range(frames):
        with torch.cuda.stream(pipe.streams[0]):
            work.copy_(q[i % 8], non_blocking=True)
        pipe.streams[0].synchronize()
        t0 = time.perf_counter()
        pipe.enqueue(0, work, uv[i % 8], seed=i + 1)
        objs, counts = pipe.fetch(0)
        lat.append(time.perf_counter() - t0)
        assert len(objs) == n_vis, (i, len(objs))
    lat = np.array(lat[20:]) * 1e3
    print(f"models {models} n_vis {n_vis}: wall latency enqueue -> objects on the host, ms: median {np.median(lat):.3f} min {lat.min():.3f} p90 {np.percentile(lat, 90):.3f}")
    pipe.close()

def report(d):
    f = [p for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True)][0]
    rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: r[1])
    # frames start at the normalise kernel of a 3000-query frame; the copy kernels of torch are skipped
    starts = [i for i, r in enumerate(rows) if r[0].startswith("void mh::normalize_kernel") or "normalize_kernel<true>" in r[0]]
    frames = [rows[a:b] for a, b in zip(starts, starts[1:])]
    frames = [[k for k in fr if "mh::" in k[0] or "mh_" in k[0]] for fr in frames]
    n = max(set(len(fr) for fr in frames), key=[len(fr) for fr in frames].count)
    frames = [fr for fr in frames if len(fr) == n][-40:]
    print(f"{len(frames)} isolated frames of {n} kernels each (medians, us)")
    print(f"{'#':>2s} {'kernel':60s} {'start':>8s} {'dur':>8s} {'gap before':>10s}")
    span = []
    tot_k = tot_g = 0.0
    for j in range(n):
        name = frames[0][j][0].replace("void mh::", "").replace("(anonymous namespace)::", "")[:60]
        st = np.median([fr[j][1] - fr[0][1] for fr in frames]) / 1e3
        du = np.median([fr[j][2] - fr[j][1] for fr in frames]) / 1e3
        gap = np.median([fr[j][1] - fr[j - 1][2] for fr in frames]) / 1e3 if j else 0.0
        tot_k += du
        tot_g += gap
        print(f"{j:2d} {name:60s} {st:8.1f} {du:8.1f} {gap:10.1f}")
    span = np.median([fr[-1][2] - fr[0][1] for fr in frames]) / 1e3
    print(f"span first start -> last end {span:.1f} us; kernels {tot_k:.1f} us, gaps {tot_g:.1f} us")

if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(*(int(a) for a in sys.argv[2:]))
    else:
        report(sys.argv[2])
