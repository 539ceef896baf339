"""gnnops_edge_reduce on a graph with heavy destinations: N = 1M, E = 10M uniform + one destination with 1M edges and ten with
100k each; piecewise hub passes (gnnops_edge_reduce_hubs) against leaving each hub to one lane group (plain entry point)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd")]
import torch
import gnnops
from gnnops import conv, _lib

gnnops.load_library()
n, e, d = 1_000_000, 10_000_000, 128
g = torch.Generator(device="cuda").manual_seed(1)
src = torch.randint(0, n, (e,), generator=g, device="cuda")
dst = torch.randint(0, n, (e,), generator=g, device="cuda")
dst[:1_000_000] = 5
for h in range(10):
    dst[1_000_000 + h * 100_000: 1_100_000 + h * 100_000] = 1000 + h
ei = torch.stack([src, dst])[:, torch.randperm(e, device="cuda")].contiguous()
x = torch.empty(n, d, dtype=torch.float16, device="cuda").normal_()
pq = torch.empty(n, 4 * d, dtype=torch.float16, device="cuda").normal_()


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / reps


L = _lib.load()
real = L.gnnops_edge_reduce_hub_workspace_bytes
for name, fn in (("copy (gather-sum) D=128 fp16", lambda: conv.edge_reduce("copy", x, ei, n, add=x)),
                 ("cgconv D=128 fp16", lambda: conv.edge_reduce("cgconv", pq[:, 2 * d:], ei, n, p=pq[:, :2 * d], add=x))):
    with_hubs = timed(fn)
    ref = fn().float()

    class NoHubs:   # same library, hub workspace size reported as 0: the plain path
        def __getattr__(self, k):
            return (lambda E, K: 0) if k == "gnnops_edge_reduce_hub_workspace_bytes" else getattr(L, k)
    orig = _lib.load
    _lib.load = lambda: NoHubs()
    try:
        without = timed(fn, reps=1)
        diff = (fn().float() - ref).abs().max().item() / ref.abs().max().item()
    finally:
        _lib.load = orig
    print(f"{name:32s} hubs piecewise {with_hubs:8.3f} ms   one lane group per hub {without:8.3f} ms   max rel diff {diff:.1e}", flush=True)
