"""rocprofv3 target: the long 1-D scatters (csrc/scatter1d.hip) at the reference's 1.47e9-element shape, three calls each.
usage: prof_scatter1d.py [n] [min|mean]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_472_353_280
what = sys.argv[2] if len(sys.argv) > 2 else "min"
gnnops.set_plan_cache(False)
s = torch.rand(n, device="cuda"); i = torch.randint(0, n, (n,), device="cuda")
for _ in range(3):
    o = gnnops.scatter_min(s, i, 0, dim_size=n) if what == "min" else gnnops.scatter_mean(s, i, 0, dim_size=n); del o
torch.cuda.synchronize()
