"""Host-side cost per call at the reference's SMALLEST shapes ((223, 223) fp16, full-shape int64 index — benchmark_scatter_add.py:40-46):
these calls are launch / host bound, so what the Timer reports is how long Python takes to enqueue one. cProfile of 3000 calls each."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops, torch_scatter
gnnops.install()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
cache = (sys.argv[2] if len(sys.argv) > 2 else "cold") == "warm"
gnnops.set_plan_cache(cache)
L = 223
src = torch.rand(L, L, device="cuda").half()
idx = torch.randint(0, L, (L, L), device="cuda")
row = torch.randint(0, L, (L,), device="cuda")
cases = {
    "torch_scatter.scatter_add(src, idx, dim=0)": lambda: torch_scatter.scatter_add(src, idx, dim=0),
    "zeros_like + scatter_add_ (dim 0)": lambda: torch.zeros_like(src).scatter_add_(0, idx, src),
    "torch_scatter.scatter_min(src, idx, 0)": lambda: torch_scatter.scatter_min(src, idx, 0),
    "torch.index_select(src, 0, row)": lambda: torch.index_select(src, 0, row),
    "src.index_add_(1, row, src)": lambda: src.index_add_(1, row, src),
    "torch.gather(src, 0, idx)": lambda: torch.gather(src, 0, idx),
}
for name, fn in cases.items():
    if which != "all" and which not in name:
        continue
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3000
    print(f"=== {name}: {dt * 1e6:.1f} us per call ({'warm' if cache else 'cold'})", flush=True)
    if which != "all":
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3000):
            fn()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
