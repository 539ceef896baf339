"""Workload for the PMC passes: a few launches of each hot kernel at the config-2 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops

N, E, D = 10_000_000, 50_000_000, 128
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(42)
src = torch.rand(E, D, generator=g, device=dev)
idx = torch.randint(0, N, (E,), generator=g, device=dev)
gnnops.set_plan_cache(False)
for _ in range(3):
    plan = gnnops.Plan(idx, N)
    out = gnnops.scatter_add(src, plan, 0)
    mn = gnnops.scatter_min(src, plan, 0)
    del out, mn
    out = gnnops.scatter_add(src, idx, 0, dim_size=N)   # plan cache off: the one-shot form (bucket_reduce_kernel)
    mn = gnnops.scatter_min(src, idx, 0, dim_size=N)
    del out, mn
del src
table = torch.rand(N, D, generator=g, device=dev)
for _ in range(3):
    o = gnnops.index_select(table, 0, idx, plan=plan)
    del o
del table, idx, plan
torch.cuda.empty_cache()
# config 3: spmm over CSR 2M x 2M, nnz 40M, D = 256 bf16 (spmm_rows_kernel<bf16>): compulsory 2.464 GB, gathered 20.48 GB
M, nnz, D3 = 2_000_000, 40_000_000, 256
row = torch.randint(0, M, (nnz,), generator=g, device=dev).sort().values
col = torch.randint(0, M, (nnz,), generator=g, device=dev)
rowptr = torch.zeros(M + 1, dtype=torch.int32, device=dev)
rowptr[1:] = torch.bincount(row, minlength=M).cumsum(0).to(torch.int32)
del row
val = torch.rand(nnz, generator=g, device=dev).to(torch.bfloat16)
Bm = torch.rand(M, D3, generator=g, device=dev).to(torch.bfloat16)
for _ in range(3):
    o = gnnops.spmm_csr(rowptr, col, val, Bm)
    del o
del rowptr, col, val, Bm
torch.cuda.empty_cache()
# config 4: fused index_select + sum, E = N = 100M rows of D = 128 fp16 (select_sum_rows_kernel<half>): 26.4 GB
E4 = 100_000_000
table = torch.empty(E4, 128, device=dev, dtype=torch.float16).uniform_(0, 1, generator=g)
index = torch.randint(0, E4, (E4,), generator=g, device=dev)
for _ in range(2):
    o = gnnops.index_select_sum(table, 0, index)
    del o
torch.cuda.synchronize()
print("done")
