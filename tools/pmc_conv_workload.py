"""Workload for the PMC / kernel-trace passes of the fused edge pass (csrc/conv.hip) at BASELINE config 2's graph size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd")]
import torch
import gnnops
from gnnops import conv

gnnops.load_library()
n, e, d = 10_000_000, 50_000_000, 128
g = torch.Generator(device="cuda").manual_seed(5)
ei = torch.randint(0, n, (2, e), generator=g, device="cuda")
pq = torch.empty(n, 4 * d, dtype=torch.float16, device="cuda").normal_()
x = torch.empty(n, d, dtype=torch.float16, device="cuda").normal_()
for _ in range(3):
    conv.edge_reduce("copy", x, ei, n, add=x)                                                   # 16-B lanes
    conv.edge_reduce("film", pq[:, 2 * d:3 * d], ei, n, p=pq[:, :2 * d], add=x, aggr=("mean",))   # 16-B lanes
    conv.edge_reduce("cgconv", pq[:, 2 * d:], ei, n, p=pq[:, :2 * d], add=x)                    # 4-B lanes, one row per wave
del pq
pq32 = torch.empty(n, 4 * d, dtype=torch.float32, device="cuda").normal_()
x32 = x.float()
for _ in range(3):
    conv.edge_reduce("cgconv", pq32[:, 2 * d:], ei, n, p=pq32[:, :2 * d], add=x32)              # 8-B lanes, one row per wave
torch.cuda.synchronize()
print("done")
