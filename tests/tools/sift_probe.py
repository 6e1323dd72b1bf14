"""Agreement report HIP SIFT vs oracle on a golden frame + timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import orclib
from moped_amd import capi
from scipy.spatial import cKDTree
G = np.load(os.path.join(ROOT, "tests", "golden", "sift_ref_frames.npz"))
c = capi.Context(0)
for f in (0, 3):
    gray = G[f"gray{f}"]
    wx, ws, wd = G[f"xy{f}"], G[f"scale_ori{f}"], G[f"desc{f}"]
    gx, gs, gd = c.sift(gray)
    print(f"frame {f}: reference {len(wx)} keypoints, HIP {len(gx)}")
    if len(gx) == 0: continue
    t = cKDTree(np.concatenate([wx, ws], 1)); dist, j = t.query(np.concatenate([gx, gs], 1)); ok = dist < 1e-2
    print(f"  paired {ok.sum()}  max|dxy| {np.abs(gx[ok]-wx[j[ok]]).max():.2e}  max|dscale/scale| {np.abs(gs[ok,0]/ws[j[ok],0]-1).max():.2e}"
          f"  max|dori| {np.abs(gs[ok,1]-ws[j[ok],1]).max():.2e}")
    dd = np.abs(gd[ok] - wd[j[ok]]).max(1)
    print(f"  descriptor max|d|: median {np.median(dd):.2e} 99% {np.quantile(dd,0.99):.2e} max {dd.max():.2e}; bit-identical descriptors: {(dd==0).sum()}")
    print(f"  same order: {np.array_equal(j[ok], np.sort(j[ok]))}; unpaired HIP: {np.nonzero(~ok)[0][:10]}")
dev = torch.device("cuda:0")
g = torch.from_numpy(G["gray0"]).to(dev)
cap = 8192
desc = torch.empty(cap, 128, device=dev); xy = torch.empty(cap, 2, device=dev); n = torch.zeros(1, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(); c.set_stream(s.cuda_stream)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10): c.sift_dev(g.data_ptr(), 640, 480, 1, desc.data_ptr(), xy.data_ptr(), 0, cap, n.data_ptr())
    e1.record(s); s.synchronize()
    print(f"HIP SIFT 640x480 (doubled): {e0.elapsed_time(e1)/10:.3f} ms per frame, {int(n.item())} keypoints")
t0 = time.perf_counter(); orclib.sift(G["gray0"]); print(f"oracle (1 CPU thread): {1e3*(time.perf_counter()-t0):.0f} ms")
