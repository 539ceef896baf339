"""Time gnnops_index_select (pull) and _planned (push) at the config-2 shape under env tuning knobs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops

N, E, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (10_000_000, 50_000_000, 128)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(42)
table = torch.rand(N, D, generator=g, device=dev)
idx = torch.randint(0, N, (E,), generator=g, device=dev)
plan = gnnops.Plan(idx, N)

def timeit(fn, iters=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
alg = N * D * 4 + E * 8 + E * D * 4
which = os.environ.get("WHICH", "both")
msg = f"sel_mode={os.environ.get('GNNOPS_SEL_MODE','0')} push_mode={os.environ.get('GNNOPS_PUSH_MODE','0')} grid={os.environ.get('GNNOPS_SEL_GRID','32')}"
if which in ("both", "pull"):
    ms = timeit(lambda: gnnops.index_select(table, 0, idx))
    msg += f" pull_ms={ms:.3f} ({alg/ms/1e6:.0f} GB/s)"
if which in ("both", "push"):
    ms = timeit(lambda: gnnops.index_select(table, 0, idx, plan=plan))
    msg += f" push_ms={ms:.3f} ({alg/ms/1e6:.0f} GB/s)"
print(msg)
