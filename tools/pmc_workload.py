"""Workload for the PMC passes: a few launches of each hot kernel at the config-2 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops

N, E, D = 10_000_000, 50_000_000, 128
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(42)
src = torch.rand(E, D, generator=g, device=dev)
idx = torch.randint(0, N, (E,), generator=g, device=dev)
gnnops.set_plan_cache(False)
for _ in range(3):
    plan = gnnops.Plan(idx, N)
    out = gnnops.scatter_add(src, plan, 0)
    mn = gnnops.scatter_min(src, plan, 0)
    del out, mn
    out = gnnops.scatter_add(src, idx, 0, dim_size=N)   # plan cache off: the one-shot form (bucket_reduce_kernel)
    mn = gnnops.scatter_min(src, idx, 0, dim_size=N)
    del out, mn
del src
table = torch.rand(N, D, generator=g, device=dev)
for _ in range(3):
    o = gnnops.index_select(table, 0, idx, plan=plan)
    del o
torch.cuda.synchronize()
print("done")
