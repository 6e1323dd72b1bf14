"""SIFT extraction alone with D frames in flight (D contexts / streams): do the pyramid's kernels of different frames
overlap?  usage: sift_depth_probe.py [frames=2000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from moped_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
gold = np.load(os.path.join(ROOT, "tests", "golden", "sift_ref_frames.npz"))
dev = torch.device("cuda:0")
img = torch.from_numpy(gold["gray0"]).to(dev)
h, w = gold["gray0"].shape
for D in (1, 2, 4, 8, 16):
    ctxs, streams, bufs = [], [], []
    for i in range(D):
        c = capi.Context(0); s = torch.cuda.Stream(device=dev); c.set_stream(s.cuda_stream)
        ctxs.append(c); streams.append(s)
        bufs.append((torch.empty((1024, 128), device=dev), torch.empty((1024, 2), device=dev), torch.zeros(1, dtype=torch.int32, device=dev)))
    def go(k):
        for i in range(k):
            d, xy, cnt = bufs[i % D]
            ctxs[i % D].sift_dev(img.data_ptr(), w, h, True, d.data_ptr(), xy.data_ptr(), 0, 1024, cnt.data_ptr())
    go(4 * D); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(n); th = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"depth {D}: {1e3*dt/n:.3f} ms per frame ({n/dt:.0f} frames/s); host enqueue {1e3*th/n:.3f} ms per frame; keypoints {int(bufs[0][2].item())}", flush=True)
    for c in ctxs: c.close()
