"""Latency of a cross-stream event hand-off vs an in-stream dependent launch (tiny kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev); y = torch.zeros(64, device=dev)
a, b = torch.cuda.Stream(), torch.cuda.Stream()
n = 2000
def chain_same():
    with torch.cuda.stream(a):
        for _ in range(n): x.add_(1.0); x.add_(1.0)
def chain_hop():
    for _ in range(n):
        with torch.cuda.stream(a): x.add_(1.0)
        e = torch.cuda.Event(); e.record(a); b.wait_event(e)
        with torch.cuda.stream(b): x.add_(1.0)
        f = torch.cuda.Event(); f.record(b); a.wait_event(f)
for name, fn in (("same stream", chain_same), ("ping-pong over 2 streams", chain_hop)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); th = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:28s}: {1e6*dt/(2*n):7.2f} us per kernel (host {1e6*th/(2*n):.2f} us)")
