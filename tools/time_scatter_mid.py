"""Layout-F scatter (the reference's (L, L) full-shape index) at mid sizes: wide vs narrowed LDS strips (separate processes:
GNNOPS_LDS_NARROW is read once)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch

import gnnops

g = torch.Generator(device="cuda").manual_seed(1)
for L in (500, 1000, 1500, 2000, 2500, 3500):
    src = torch.rand(L, L, generator=g, device="cuda").half()
    line = f"L={L:5d}"
    for dim in (0, 1):
        idx = torch.randint(0, L, (L, L), generator=g, device="cuda")
        for red in ("sum", "min"):
            f = (lambda: gnnops.scatter(src, idx, dim=dim, dim_size=L, reduce=red))
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(30):
                f()
            e.record()
            torch.cuda.synchronize()
            line += f"  {red} dim{dim} {s.elapsed_time(e) / 30 * 1e3:7.1f} us"
    print(line, flush=True)
