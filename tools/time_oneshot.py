"""Cold scatter at config 2: one-shot form (bucket.hip) vs plan build + segment reduce, interleaved in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops
from gnnops import ops

N, E, D = 10_000_000, 50_000_000, 128
g = torch.Generator(device="cuda").manual_seed(42)
src = torch.rand(E, D, generator=g, device="cuda")
idx = torch.randint(0, N, (E,), generator=g, device="cuda")


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        o = fn()
        del o
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def plan_path(reduce):
    p = gnnops.Plan(idx, N)
    return gnnops.scatter(src, p, 0, dim_size=N, reduce=reduce)


gnnops.set_plan_cache(False)
for rep in range(3):
    for r in ("sum", "min", "mean"):
        a = timed(lambda: gnnops.scatter(src, idx, 0, dim_size=N, reduce=r))
        b = timed(lambda: plan_path(r))
        print(f"{r:5s} oneshot {a:7.3f} ms   plan+segment {b:7.3f} ms", flush=True)

table = torch.rand(N, D, generator=g, device="cuda")
for rep in range(3):
    gnnops.set_plan_cache(False)
    a = timed(lambda: gnnops.index_select(table, 0, idx))
    b = timed(lambda: gnnops.index_select(table, 0, idx, plan=gnnops.Plan(idx, N)))
    print(f"index_select oneshot {a:7.3f} ms   plan+push {b:7.3f} ms", flush=True)
