"""A launch-bound step (small graph: scatter_add + scatter_min + index_select + addmm, ~25 kernels) eager vs replayed as one
HIP graph (torch.cuda.CUDAGraph over our C-ABI launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops

gnnops.set_plan_cache(False)
for N, E, D in ((3000, 20000, 64), (100_000, 500_000, 64), (1_000_000, 5_000_000, 128)):
    gen = torch.Generator(device="cuda").manual_seed(0)
    src = torch.rand(E, D, generator=gen, device="cuda")
    idx = torch.randint(0, N, (E,), generator=gen, device="cuda")
    w = torch.rand(D, D, generator=gen, device="cuda").to(torch.bfloat16)

    def step():
        agg = gnnops.scatter_add(src, idx, 0, dim_size=N)
        mn, arg = gnnops.scatter_min(src, idx, 0, dim_size=N)
        sel = gnnops.index_select(agg, 0, idx)
        y = gnnops.addmm(agg.to(torch.bfloat16), agg.to(torch.bfloat16), w)
        return agg, mn, arg, sel, y

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = step()

    def timed(fn, iters=50):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    print(f"N={N} E={E} D={D}: eager {timed(step):.3f} ms/step, graph replay {timed(graph.replay):.3f} ms/step", flush=True)
