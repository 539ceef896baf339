"""Config 3 SpMM (CSR 2M x 2M, nnz 40M, D=256 bf16): time spmm_rows_kernel with the dense operand swept in column slices
(GNNOPS_SPMM_GSHIFT: lanes per row = 2^g, slice = 2^g * 16 B of every 512-B row), nontemporal vs cached gathers.
usage (GPU box): python tools/time_spmm.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnn-ops-benchmark_amd"))
import torch

import gnnops


def ev_ms(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(42)
    M, nnz = 2_000_000, 40_000_000
    row = torch.randint(0, M, (nnz,), generator=g, device=dev).sort().values
    col = torch.randint(0, M, (nnz,), generator=g, device=dev)
    rowptr = torch.zeros(M + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.bincount(row, minlength=M).cumsum(0).to(torch.int32)
    del row
    val = torch.rand(nnz, generator=g, device=dev).to(torch.bfloat16)
    for D in (256, 128):
        Bm = torch.rand(M, D, generator=g, device=dev).to(torch.bfloat16)
        ref = None
        for gs in ("", "4", "3", "2", "1"):
            if gs:
                os.environ["GNNOPS_SPMM_GSHIFT"] = gs
            else:
                os.environ.pop("GNNOPS_SPMM_GSHIFT", None)
            out = gnnops.spmm_csr(rowptr, col, val, Bm)
            if ref is None:
                ref = out
            same = bool(torch.equal(out, ref))
            ms = ev_ms(lambda: gnnops.spmm_csr(rowptr, col, val, Bm))
            print(f"D={D} gshift={gs or 'auto'}: {ms:.3f} ms  gathered {nnz * D * 2 / ms / 1e6:.0f} GB/s  identical={same}", flush=True)
        os.environ.pop("GNNOPS_SPMM_GSHIFT", None)
        del Bm


if __name__ == "__main__":
    main()
