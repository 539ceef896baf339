"""cgconv edge pass only (D = 128 fp16, N = 10M, E = 50M), three launches: workload for the SQ counter pass."""
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
    conv.edge_reduce("cgconv", pq[:, 2 * d:], ei, n, p=pq[:, :2 * d], add=x)
torch.cuda.synchronize()
print("done")
