"""Layout-F scatters at the reference's largest published shape ((6708, 6708) fp16, index int64, RF 1 / 8): the int64 index
(cold: what a first call runs) against the narrowed copy a repeated call streams.
usage (GPU box): python tools/time_narrow_index.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch

import gnnops


def ev(fn, iters=20):
    fn(); fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


dev = "cuda"
for L in (6708, 2000):
    src = torch.rand(L, L, device=dev).half()
    for rf in (1, 8):
        idx = torch.randint(0, L // rf, (L, L), device=dev)
        for red in ("sum", "mean", "min"):
            for dim in (0, 1):
                gnnops.set_plan_cache(False)
                cold = ev(lambda: gnnops.scatter(src, idx, dim, dim_size=L // rf, reduce=red))
                gnnops.set_plan_cache(True)
                warm = ev(lambda: gnnops.scatter(src, idx, dim, dim_size=L // rf, reduce=red))
                print(f"L={L} RF{rf} {red:4s} dim{dim}: int64 index {cold * 1e3:7.1f} us   narrowed copy {warm * 1e3:7.1f} us   ({cold / warm:.2f}x)", flush=True)
gnnops.set_plan_cache(False)
