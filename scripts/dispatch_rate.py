"""How many small dependent kernel dispatches per second does the GPU front end retire,
over S streams?  usage: dispatch_rate.py [streams] [launches_per_stream]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moped_amd import capi
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda:0")
ctxs, streams, bufs = [], [], []
for i in range(S):
    c = capi.Context(0); s = torch.cuda.Stream(); c.set_stream(s.cuda_stream)
    q = torch.rand(64, 128, device=dev); qn = torch.empty(64, device=dev)
    ctxs.append(c); streams.append(s); bufs.append((q, qn))
def go(k):
    for i in range(k):
        for j in range(S):
            ctxs[j].normalize_dev(bufs[j][0].data_ptr(), bufs[j][1].data_ptr(), 64)
go(50); torch.cuda.synchronize()
t0 = time.perf_counter(); go(n); th = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"streams {S}: {S*n/dt/1e3:.1f} k dispatches/s ({1e6*dt/(S*n):.2f} us each; host {1e6*th/(S*n):.2f} us each)")
