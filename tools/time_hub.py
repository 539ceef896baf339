"""How slow is a hub? scatter_add / scatter_max with a fraction of all edges on ONE destination (power-law graphs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops

N, E, D = 1_000_000, 5_000_000, 128
g = torch.Generator(device="cuda").manual_seed(0)
src = torch.rand(E, D, generator=g, device="cuda")


def timed(fn, iters=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for hub in (0, 10_000, 100_000, 1_000_000):
    idx = torch.randint(0, N, (E,), generator=g, device="cuda")
    if hub:
        idx[torch.randperm(E, device="cuda")[:hub]] = 7
    for cache in (False, True):
        gnnops.set_plan_cache(cache)
        a = timed(lambda: gnnops.scatter_add(src, idx, 0, dim_size=N))
        m = timed(lambda: gnnops.scatter_max(src, idx, 0, dim_size=N))
        print(f"hub degree {hub:>8d}  {'plan (cached)' if cache else 'one-shot     '}  scatter_add {a:8.3f} ms   scatter_max {m:8.3f} ms", flush=True)
