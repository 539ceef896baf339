"""The reference's '>95 % memory' layout-F scatter shapes (data/scatter_{min,max,mean}.csv:2-3): timings per call, for
rocprofv3 --kernel-trace --stats. usage: python tools/prof_big_scatter.py [which]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch

import gnnops
import torch_scatter

gnnops.set_plan_cache(False)
dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "all"


def timed(name, fn, iters=2):
    out = fn()
    del out
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
        del out
    torch.cuda.synchronize()
    print(f"{name:50s} {(time.perf_counter() - t0) / iters * 1e3:9.2f} ms", flush=True)


if which in ("all", "1d"):
    n = 1_472_353_280
    s, i = torch.rand(n, device=dev), torch.randint(0, n, (n,), device=dev)
    timed("scatter_min 1-D 1.47e9 fp32", lambda: torch_scatter.scatter_min(s, i, 0))
    timed("scatter_mean 1-D 1.47e9 fp32", lambda: torch_scatter.scatter_mean(s, i, 0))
    del s, i
    torch.cuda.empty_cache()
if which in ("all", "2d"):
    L = 38000
    s, i = torch.rand(L, L, device=dev), torch.randint(0, L, (L, L), device=dev)
    timed("scatter_max (38000)^2 fp32 dim 0", lambda: torch_scatter.scatter_max(s, i, 0))
    timed("scatter_max (38000)^2 fp32 dim 1", lambda: torch_scatter.scatter_max(s, i, 1))
    L = 36400
    s, i = torch.rand(L, L, device=dev), torch.randint(0, L, (L, L), device=dev)
    timed("scatter_mean (36400)^2 fp32 dim 0", lambda: torch_scatter.scatter_mean(s, i, 0))
    timed("scatter_mean (36400)^2 fp32 dim 1", lambda: torch_scatter.scatter_mean(s, i, 1))
