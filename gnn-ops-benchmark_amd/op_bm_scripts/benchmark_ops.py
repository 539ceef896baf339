#!/usr/bin/env python3
"""One parametrised counterpart of the reference's 17 op_bm_scripts/benchmark_*.py.

Same protocol as the reference: inputs built on the device, `op_*` body handed to
torch.utils.benchmark.Timer(stmt, setup, globals).timeit(n) — 2 warm-ups, one synchronised block of n
calls, mean seconds per call — one CSV row per (op, shape, dim, reduce factor). The op bodies are the
reference's, calling the same names (`torch_scatter.scatter_add`, `torch.index_select`, ...) — which resolve
to the gfx950 kernels through the shim packages and `gnnops.install()`.

    python benchmark_ops.py --ops scatter_add,index_select --point ref_max      # the reference's largest published shapes
    python benchmark_ops.py --ops all --point ref_min --csv out.csv

The A100-40GB numbers printed beside ours are the reference's own (file:line given there).
"""
import argparse
import csv
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import torch
import torch.utils.benchmark as benchmark

import gnnops
import torch_scatter
import torch_sparse
from graph_benchmark.benchmark.util import setup_seed

# ---- op bodies: the reference's, verbatim in meaning (file:line under the reference root) ----------------------
def op_scatter_add(src, idx, dim):            # benchmark_scatter_add.py:15-19
    return torch_scatter.scatter_add(src, idx, dim=dim)
def op_native_scatter_add_(src, idx, dim):    # benchmark_scatter_add.py:22-25
    temp = torch.zeros_like(src)
    temp.scatter_add_(dim, idx, src)
def op_scatter_min(src, idx, dim):            # benchmark_scatter_min.py:15-18
    return torch_scatter.scatter_min(src, idx, dim)
def op_scatter_max(src, idx, dim):            # benchmark_scatter_max.py:15-18
    return torch_scatter.scatter_max(src, idx, dim)
def op_scatter_mean(src, idx, dim):           # benchmark_scatter_mean.py:15-18
    return torch_scatter.scatter_mean(src, idx, dim)
def op_native_scatter_multiply_(src, idx):    # benchmark_scatter_multiply.py:42-45
    temp = torch.zeros_like(src)
    temp.scatter_(-1, idx, src, reduce="multiply")
def op_native_index_select(input, dim, index):  # benchmark_native_index_select.py:12-15
    return torch.index_select(input, dim, index)
def op_native_index_add_(input, dim, index, source):  # benchmark_native_index_add_.py:13-16
    input.index_add_(dim, index, source)
def op_native_gather(input, dim, index):      # benchmark_native_gather.py:14-17
    return torch.gather(input, dim, index)
def op_native_sort(input, dim, stable):       # benchmark_native_sort.py:28-30
    return torch.sort(input, dim=dim, stable=stable)
def op_native_smm(matA, matB):                # benchmark_sparse_spmm.py:12-14 / benchmark_sparse_spspmm.py:12-14
    return torch.sparse.mm(matA, matB)
def op_sparse_coalesce(index, value, m, n):   # benchmark_sparse_coalesce.py:35-37
    return torch_sparse.coalesce(index=index, value=value, m=m, n=n)
def op_native_transpose(matA):                # benchmark_sparse_transpose.py:13-16
    return gnnops.transpose_contiguous(matA)
def op_fused_index_select_reduce(input, dim, index):     # benchmark_fused_index_select_reduce.py:12-20
    return gnnops.index_select_sum(input, dim, index)
def op_unfused_index_select_reduce(input, dim, index):
    return torch.index_select(input, dim, index).float().sum()
def op_fused_index_add_reduce(input, dim, index, other):  # benchmark_fused_index_add_reduce.py:12-20
    return gnnops.index_add_select_sum(input, dim, index, other)
def op_native_addmm(input, mat1, mat2):       # benchmark_native_addmm.py:13-16
    return torch.addmm(input, mat1, mat2)
def op_native_matmul(input, other):           # benchmark_native_matmul.py:13-16
    return torch.matmul(input, other)


DEV = "cuda"
H = torch.float16


def _rand(shape, dtype=H):
    return torch.rand(shape, device=DEV, dtype=torch.float32).to(dtype)


def _sq(L, rf=1, dtype=H):
    src = _rand((L, L), dtype)
    idx = torch.randint(0, max(L // rf, 1), (L, L), device=DEV, dtype=torch.int64)
    return src, idx


# (name, reference max L, reference min L, A100 ms at max L by case, builder(L) -> list of (case, stmt, globals, alg_bytes))
def cases_scatter(name, fn, has_arg=False):
    def build(L):
        out = []
        for rf in (1, 8):
            src, idx = _sq(L, rf)
            for dim in (0, 1):
                alg = src.numel() * 2 + idx.numel() * 8 + (L // rf) * L * (2 + (8 if has_arg else 0))
                out.append((f"RF{rf} dim{dim}", f"{fn.__name__}(src, idx, {dim})", {"src": src, "idx": idx, fn.__name__: fn}, alg))
        return out
    return build


def build_index_select(L):
    out = []
    inp = _rand((L, L))
    for rf in (1, 8):
        for dim in (0, 1):
            index = torch.randint(0, L, (L // rf,), device=DEV)
            alg = inp.numel() * 2 + index.numel() * 8 + (L // rf) * L * 2
            out.append((f"RF{rf} dim{dim}", f"op_native_index_select(input, {dim}, index)",
                        {"input": inp, "index": index, "op_native_index_select": op_native_index_select}, alg))
    return out


def build_index_add(L):
    inp, source = _rand((L, L)), _rand((L, L))
    index = torch.randint(0, L, (L,), device=DEV)
    alg = 3 * L * L * 2 + L * 8
    return [("dim1", "op_native_index_add_(input, 1, index, source)",
             {"input": inp, "index": index, "source": source, "op_native_index_add_": op_native_index_add_}, alg)]


def build_gather(L):
    inp = _rand((L, L))
    idx = torch.randint(0, L, (L, L), device=DEV)
    return [(f"dim{d}", f"op_native_gather(input, {d}, index)",
             {"input": inp, "index": idx, "op_native_gather": op_native_gather}, L * L * (2 + 8 + 2)) for d in (0, 1)]


def build_sort(L):
    out = []
    x = torch.nn.functional.dropout(torch.rand((L, L), device=DEV), p=0.5)
    for d in (0, 1):
        out.append((f"2d dim{d} stable", f"op_native_sort(input, {d}, True)", {"input": x, "op_native_sort": op_native_sort},
                    L * L * (4 + 4 + 8)))
    x1 = torch.rand((L * L,), device=DEV)
    out.append(("1d stable", "op_native_sort(input, 0, True)", {"input": x1, "op_native_sort": op_native_sort}, L * L * 16))
    return out


def _sparse(L, sparsity, dtype=torch.float32):
    dense = torch.nn.functional.dropout(torch.rand((L, L), device=DEV, dtype=dtype), p=sparsity)
    return dense, dense.to_sparse()


def build_spmm(L):
    _, A = _sparse(L, 0.999)
    Bd, _ = _sparse(L, 0.999)
    nnz = A._nnz()
    return [("coo x dense s=.999", "op_native_smm(matA, matB)", {"matA": A, "matB": Bd, "op_native_smm": op_native_smm},
             nnz * 20 + 2 * L * L * 4)]


def build_spspmm(L):
    _, A = _sparse(L, 0.995)
    _, B = _sparse(L, 0.995)
    return [("coo x coo s=.995", "op_native_smm(matA, matB)", {"matA": A, "matB": B, "op_native_smm": op_native_smm},
             (A._nnz() + B._nnz()) * 20)]


def build_coalesce(L):
    out = []
    _, A = _sparse(L, 0.5)
    idx, val = A._indices(), A._values()
    for rf in (1, 8):
        index = torch.cat([idx] * rf, dim=1)
        index = index[:, torch.randperm(index.shape[1], device=DEV)]
        value = torch.cat([val] * rf)
        out.append((f"dup x{rf}", "op_sparse_coalesce(index, value, m, n)",
                    {"index": index, "value": value, "m": L * rf, "n": L * rf, "op_sparse_coalesce": op_sparse_coalesce},
                    index.shape[1] * (16 + 4) * 2))
    return out


def build_transpose(L):
    x = torch.nn.functional.dropout(_rand((L, L)).float(), p=0.995).half()
    return [("dense fp16", "op_native_transpose(matA)", {"matA": x, "op_native_transpose": op_native_transpose}, 2 * L * L * 2)]


def build_fused_select(L):
    out = []
    inp = _rand((L, L))
    index = torch.randint(0, L, (L,), device=DEV)
    for d in (0, 1):
        g = {"input": inp, "index": index, "op_fused_index_select_reduce": op_fused_index_select_reduce,
             "op_unfused_index_select_reduce": op_unfused_index_select_reduce}
        out.append((f"fused dim{d}", f"op_fused_index_select_reduce(input, {d}, index)", g, L * L * 2 + L * 8))
        out.append((f"unfused dim{d}", f"op_unfused_index_select_reduce(input, {d}, index)", g, 3 * L * L * 2 + L * 8))
    return out


def build_fused_add(L):
    inp = _rand((L, L))
    other = inp.clone()
    index = torch.randint(0, L, (L,), device=DEV)
    return [(f"dim{d}", f"op_fused_index_add_reduce(input, {d}, index, other)",
             {"input": inp, "index": index, "other": other, "op_fused_index_add_reduce": op_fused_index_add_reduce},
             2 * L * L * 2 + L * 8) for d in (0, 1)]


def build_addmm(L):
    a, b, c = _rand((L, L)), _rand((L, L)), _rand((L, L))
    return [("fp16", "op_native_addmm(input, mat1, mat2)", {"input": c, "mat1": a, "mat2": b, "op_native_addmm": op_native_addmm},
             4 * L * L * 2)]


def build_matmul(L):
    a, b = _rand((L, L)), _rand((L, L))
    return [("fp16", "op_native_matmul(input, other)", {"input": a, "other": b, "op_native_matmul": op_native_matmul}, 3 * L * L * 2)]


def build_multiply(L):
    src, idx = _sq(L, 1, torch.float32)
    return [("2d fp32", "op_native_scatter_multiply_(src, idx)", {"src": src, "idx": idx,
             "op_native_scatter_multiply_": op_native_scatter_multiply_}, L * L * (4 + 8 + 8))]


# name -> (L at the reference's largest published point, smallest point, builder, {case: A100 ms}, source in BASELINE.md)
OPS = {
    "scatter_add": (6708, 223, cases_scatter("scatter_add", op_scatter_add), {"RF1 dim0": 6.688, "RF1 dim1": 3.678, "RF8 dim0": 2.542, "RF8 dim1": 6.937}),
    "native_scatter_add_": (6708, 223, cases_scatter("native_scatter_add_", op_native_scatter_add_), {"RF1 dim0": 6.388, "RF1 dim1": 3.394}),
    "scatter_min": (6708, 223, cases_scatter("scatter_min", op_scatter_min, True), {"RF1 dim0": 14.53, "RF1 dim1": 6.69}),
    "scatter_max": (6708, 223, cases_scatter("scatter_max", op_scatter_max, True), {"RF1 dim0": 14.53, "RF1 dim1": 6.70}),
    "scatter_mean": (6708, 223, cases_scatter("scatter_mean", op_scatter_mean), {"RF1 dim0": 13.60, "RF1 dim1": 7.62}),
    "scatter_multiply": (6708, 223, build_multiply, {}),
    "native_index_select": (14142, 2738, build_index_select, {"RF1 dim0": 2.156, "RF8 dim0": 0.273, "RF1 dim1": 3.444}),
    "native_index_add_": (10000, 1581, build_index_add, {"dim1": 7.061}),
    "native_gather": (6324, 1224, build_gather, {"dim0": 2.493, "dim1": 0.408}),
    "native_sort": (7071, 1000, build_sort, {}),
    "sparse_spmm": (7071, 1414, build_spmm, {"coo x dense s=.999": 4.788}),
    "sparse_spspmm": (7071, 1414, build_spspmm, {"coo x coo s=.995": 3.156}),
    "sparse_coalesce": (3000, 500, build_coalesce, {}),
    "sparse_transpose": (7071, 2000, build_transpose, {"dense fp16": 0.709}),
    "fused_index_select_reduce": (14142, 2738, build_fused_select, {"unfused dim0": 2.920, "fused dim0": 2.921}),
    "fused_index_add_reduce": (6708, 223, build_fused_add, {"dim0": 8.814, "dim1": 24.70}),
    "native_addmm": (8164, 1581, build_addmm, {"fp16": 7.230}),
    "native_matmul": (8164, 1581, build_matmul, {"fp16": 8.796}),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="all")
    ap.add_argument("--point", default="ref_max", choices=["ref_max", "ref_min"])
    ap.add_argument("--runs", type=int, default=20)
    ap.add_argument("--csv", default=None)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise Exception("Benchmarking only supported for CUDA")  # the reference's guard (benchmark_scatter_add.py:52-54)
    setup_seed(42)
    gnnops.install()
    names = list(OPS) if args.ops == "all" else args.ops.split(",")
    rows = []
    for name in names:
        Lmax, Lmin, build, a100 = OPS[name]
        L = Lmax if args.point == "ref_max" else Lmin
        torch.cuda.empty_cache()
        for case, stmt, g, alg in build(L):
            t = benchmark.Timer(stmt=stmt, globals=g).timeit(args.runs)
            ms = t.median * 1e3
            ref = a100.get(case) if args.point == "ref_max" else None
            rows.append([name, case, f"({L}, {L})", f"{ms:.4f}", f"{alg / ms / 1e6:.1f}", "" if ref is None else ref,
                         "" if ref is None else f"{ref / ms:.2f}"])
            print(f"{name:28s} {case:22s} L={L:6d}  {ms:9.4f} ms  {alg / ms / 1e6:8.1f} GB/s alg"
                  + ("" if ref is None else f"   A100 {ref} ms  ({ref / ms:.2f}x)"), flush=True)
            del t
    if args.csv:
        with open(args.csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["op", "case", "Input size", "GPU clock time (ms, mean per call)", "algorithmic GB/s", "A100-40GB ms (reference)", "speedup vs A100"])
            w.writerows(rows)


if __name__ == "__main__":
    main()
