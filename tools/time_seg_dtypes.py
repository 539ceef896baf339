"""Segment reduce (plan reused) across dtypes and row lengths at N=10M, E=50M."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops
N, E = 10_000_000, 50_000_000
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(42)
idx = torch.randint(0, N, (E,), generator=g, device=dev)
plan = gnnops.Plan(idx, N)
for dt, D in [(torch.float32, 128), (torch.float32, 64), (torch.float32, 32), (torch.bfloat16, 128), (torch.bfloat16, 256), (torch.float16, 64), (torch.float32, 256)]:
    src = torch.empty(E, D, device=dev, dtype=dt).uniform_(0, 1)
    for red in ("sum", "max"):
        fn = lambda: gnnops.scatter(src, plan, 0, reduce=red)
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 3
        sz = src.element_size()
        alg = E * D * sz + E * 8 + N * D * sz + (N * D * 8 if red == "max" else 0)
        print(f"{str(dt):15s} D={D:4d} {red:4s} {ms:8.3f} ms  {alg/ms/1e6:8.1f} GB/s  ({alg/ms/1e6/80:.1f} % of peak)", flush=True)
    del src
