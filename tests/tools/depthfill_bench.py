"""mh_depth_fill on a device-resident 640x480 depth map: ms per call (hole patterns of tests/test_gpu_depthfill.py)
beside the oracle's CPU restatement.  usage: depthfill_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import orclib
from moped_amd import capi
import test_gpu_depthfill as T
ctx = capi.Context(0)
dev = torch.device("cuda:0")
for kind in ("sparse", "blobs", "dense", "grid"):
    d = T.holes(kind, 480, 640, np.random.default_rng(0))
    src = torch.from_numpy(d).to(dev)
    work = torch.empty_like(src)
    fill = torch.empty((480, 640), dtype=torch.float32, device=dev)
    def once():
        work.copy_(src)
        ctx.depth_fill_dev(work.data_ptr(), 640, 480, T.K, fill.data_ptr(), 8)
    for _ in range(3): once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): once()
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter(); orclib.depth_fill(d, T.K, 8); cpu = time.perf_counter() - t0
    print(f"{kind}: holes {(d[..., 2] < 0).mean():.0%}  gpu {gpu*1e3:.3f} ms per map (incl. a 4.9 MB device copy)  oracle cpu {cpu*1e3:.2f} ms", flush=True)
