"""The same op on the same box: stock PyTorch-ROCm kernels (what the reference's native-op scripts would run on an MI355X
without this package) against gnnops. Config-2 sizes for the row ops (N = 10M, E = 50M, D = 128 fp32), config-3 for spmm."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd")]
import torch
import gnnops

gnnops.load_library()
gnnops.set_plan_cache(False)   # cold: every call pays its own partition


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def row(name, stock, ours):
    print(f"{name:58s} stock torch {stock:9.3f} ms   gnnops {ours:9.3f} ms   {stock / ours:6.2f}x", flush=True)


g = torch.Generator(device="cuda").manual_seed(42)
N, E, D = 10_000_000, 50_000_000, 128
src = torch.rand(E, D, generator=g, device="cuda")
idx = torch.randint(0, N, (E,), generator=g, device="cuda")
out = torch.zeros(N, D, device="cuda")
row("index_add_ (dim 0) N=10M E=50M D=128 fp32", timed(lambda: out.index_add_(0, idx, src), 3), timed(lambda: gnnops.index_add_(out, 0, idx, src)))
row("scatter_add rows (zeros + index_add_ vs scatter_add)", timed(lambda: torch.zeros(N, D, device="cuda").index_add_(0, idx, src), 3),
    timed(lambda: gnnops.scatter_add(src, idx, 0, dim_size=N)))
del src
table = out
row("index_select (dim 0) N=10M E=50M D=128 fp32", timed(lambda: torch.index_select(table, 0, idx), 3), timed(lambda: gnnops.index_select(table, 0, idx)))
del table, out
torch.cuda.empty_cache()
keys = torch.rand(50_000_000, generator=g, device="cuda")
row("sort 50M fp32 (stable)", timed(lambda: torch.sort(keys, stable=True), 3), timed(lambda: gnnops.sort(keys, 0, False, True)))
del keys
L = 6324
inp = torch.rand(L, L, generator=g, device="cuda").half()
gi = torch.randint(0, L, (L, L), generator=g, device="cuda")
row("gather dim 0 (6324)^2 fp16", timed(lambda: torch.gather(inp, 0, gi), 10), timed(lambda: gnnops.gather(inp, 0, gi), 10))
row("gather dim 1 (6324)^2 fp16", timed(lambda: torch.gather(inp, 1, gi), 10), timed(lambda: gnnops.gather(inp, 1, gi), 10))
L = 6708
s16 = torch.rand(L, L, generator=g, device="cuda").half()
fi = torch.randint(0, L, (L, L), generator=g, device="cuda")
row("scatter_add_ full index dim 0 (6708)^2 fp16", timed(lambda: torch.zeros_like(s16).scatter_add_(0, fi, s16), 10),
    timed(lambda: gnnops.scatter_add_(torch.zeros_like(s16), 0, fi, s16), 10))
row("scatter_add_ full index dim 1 (6708)^2 fp16", timed(lambda: torch.zeros_like(s16).scatter_add_(1, fi, s16), 10),
    timed(lambda: gnnops.scatter_add_(torch.zeros_like(s16), 1, fi, s16), 10))
del inp, gi, s16, fi
torch.cuda.empty_cache()
M, nnz, D3 = 2_000_000, 40_000_000, 256
r = torch.randint(0, M, (nnz,), generator=g, device="cuda").sort().values
c = torch.randint(0, M, (nnz,), generator=g, device="cuda")
rowptr = torch.zeros(M + 1, dtype=torch.int64, device="cuda")
rowptr[1:] = torch.bincount(r, minlength=M).cumsum(0)
val = torch.rand(nnz, generator=g, device="cuda")
B = torch.rand(M, D3, generator=g, device="cuda")
try:
    csr = torch.sparse_csr_tensor(rowptr, c, val, (M, M))
    stock = timed(lambda: torch.sparse.mm(csr, B), 3)
except Exception as exc:   # noqa: BLE001
    stock = float("nan")
    print("stock sparse.mm failed:", type(exc).__name__, exc)
row("spmm CSR 2M x 2M nnz 40M, D = 256 fp32", stock, timed(lambda: gnnops.spmm_csr(rowptr.to(torch.int32), c, val, B), 3))
