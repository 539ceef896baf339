"""Generate the golden vectors in this directory with the PyTorch CPU ops the reference's op_* bodies
call (or, for the torch_scatter family — absent everywhere — torch.scatter_reduce_ / index_add_ as the
independent formulation named in SURVEY.md §8c). Run in the build container:

    python tests/golden/make_golden.py

Inputs are seeded (42/43) CPU tensors; every case stores inputs AND expected outputs, so the GPU box
needs neither this script nor the reference. bf16 arrays are stored as uint16 bit patterns.
16-bit expectations are "accumulate in fp32, round once" (SURVEY.md §8c), i.e. torch run on
src.float() and cast back — not torch's fp16 CPU accumulation order.
arg_out expectations (first position of the extremum) are computed by brute force here: the reference
holds nothing that pins them (PARITY UNPINNED for arg tie-breaking).
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DTYPES = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def to_np(t):
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    return t.numpy().copy()


def expand_index(idx, src, dim):
    if idx.dim() == src.dim():
        return idx
    shape = [1] * src.dim()
    shape[dim] = -1
    return idx.view(shape).expand_as(src)


def brute_arg(src32, full_idx, dim, N, want_min):
    """first position e along dim attaining the extremum per destination element; E where empty."""
    s = src32.movedim(dim, 0).numpy()
    ix = full_idx.movedim(dim, 0).numpy()
    E = s.shape[0]
    rest = s.shape[1:]
    best = np.full((N,) + rest, np.inf if want_min else -np.inf, dtype=np.float32)
    arg = np.full((N,) + rest, E, dtype=np.int64)
    for e in range(E):
        for pos in np.ndindex(*rest):
            n = ix[(e,) + pos]
            v = s[(e,) + pos]
            if (v < best[(n,) + pos]) if want_min else (v > best[(n,) + pos]):
                best[(n,) + pos] = v
                arg[(n,) + pos] = e
    return torch.from_numpy(arg).movedim(0, dim).contiguous()


def scatter_cases():
    cases = {}
    g = torch.Generator()
    shapes = [((7, 5), 4), ((61, 37), 23), ((3, 17, 4), 6)]
    for seed, (shape, nmax) in zip((42, 43, 42), shapes):
        for dim in range(len(shape)):
            for layout in ("R", "F"):
                for dname, dt in DTYPES.items():
                    g.manual_seed(seed + dim)
                    src = (torch.rand(shape, generator=g) * 4 - 2).to(dt)
                    E = shape[dim]
                    N = min(nmax, E)
                    if layout == "R":
                        idx = torch.randint(0, N, (E,), generator=g)
                    else:
                        idx = torch.randint(0, N, shape, generator=g)
                    if layout == "R" and shape == (61, 37):
                        idx[idx == 3] = 4  # leave destination 3 empty
                    full = expand_index(idx, src, dim).contiguous()
                    src32 = src.float()
                    oshape = list(shape)
                    oshape[dim] = N
                    exp = {}
                    exp["sum"] = torch.zeros(oshape).scatter_add_(dim, full, src32).to(dt)
                    exp["mean"] = torch.zeros(oshape).scatter_reduce_(dim, full, src32, "mean", include_self=False).to(dt)
                    exp["mul"] = torch.ones(oshape).scatter_reduce_(dim, full, src32, "prod", include_self=True).to(dt)
                    exp["min"] = torch.zeros(oshape).scatter_reduce_(dim, full, src32, "amin", include_self=False).to(dt)
                    exp["max"] = torch.zeros(oshape).scatter_reduce_(dim, full, src32, "amax", include_self=False).to(dt)
                    key = f"scatter_{'x'.join(map(str, shape))}_d{dim}_{layout}_{dname}"
                    cases[key + "_src"] = to_np(src)
                    cases[key + "_idx"] = idx.numpy()
                    cases[key + "_N"] = np.int64(N)
                    for r, t in exp.items():
                        cases[f"{key}_{r}"] = to_np(t)
                    cases[key + "_argmin"] = brute_arg(src32, full, dim, N, True).numpy()
                    cases[key + "_argmax"] = brute_arg(src32, full, dim, N, False).numpy()
    # hand-computable known answer: scatter_min([[3,1,2]], idx [[0,0,2]]) -> [1,0,2], arg [1,3,2]
    cases["known_min_src"] = np.array([[3.0, 1.0, 2.0]], dtype=np.float32)
    cases["known_min_idx"] = np.array([[0, 0, 2]], dtype=np.int64)
    cases["known_min_out"] = np.array([[1.0, 0.0, 2.0]], dtype=np.float32)
    cases["known_min_arg"] = np.array([[1, 3, 2]], dtype=np.int64)
    # all indices equal (one destination takes everything); index.max()+1 < rows
    g.manual_seed(42)
    src = torch.rand(33, 8, generator=g)
    idx = torch.full((33,), 2, dtype=torch.int64)
    cases["allsame_src"] = src.numpy()
    cases["allsame_idx"] = idx.numpy()
    cases["allsame_sum"] = torch.zeros(3, 8).index_add_(0, idx, src).numpy()
    # the reference's first sweep length (benchmark_scatter_add.py:40-41 -> 223), fp16, layout F, RF=2
    g.manual_seed(42)
    src = torch.rand(223, 223, generator=g).half()
    idx = torch.randint(0, 223 // 2, (223, 223), generator=g)
    for dim in (0, 1):
        oshape = [223, 223]
        oshape[dim] = 223 // 2
        cases[f"ref223_d{dim}_sum"] = torch.zeros(oshape).scatter_add_(dim, idx, src.float()).half().numpy()
    cases["ref223_src"] = src.numpy()
    cases["ref223_idx"] = idx.numpy().astype(np.int16)  # values < 111; widened to int64 by the tests
    return cases


def native_cases():
    cases = {}
    g = torch.Generator()
    for shape in [(7, 5), (61, 37), (3, 17, 4)]:
        for dim in range(len(shape)):
            for dname, dt in DTYPES.items():
                g.manual_seed(43 + dim)
                inp = (torch.rand(shape, generator=g) * 4 - 2).to(dt)
                Nn = shape[dim]
                E = Nn + 3
                idx = torch.randint(0, Nn, (E,), generator=g)
                key = f"native_{'x'.join(map(str, shape))}_d{dim}_{dname}"
                cases[key + "_in"] = to_np(inp)
                cases[key + "_idx"] = idx.numpy()
                cases[key + "_index_select"] = to_np(torch.index_select(inp, dim, idx))
                gshape = list(shape)
                gshape[dim] = E
                gidx = torch.randint(0, Nn, gshape, generator=g)
                cases[key + "_gidx"] = gidx.numpy()
                cases[key + "_gather"] = to_np(torch.gather(inp, dim, gidx))
                sshape = list(shape)
                sshape[dim] = E
                source = (torch.rand(sshape, generator=g) * 2 - 1).to(dt)
                cases[key + "_source"] = to_np(source)
                # fp32 accumulate from the rounded input, round once
                cases[key + "_index_add"] = to_np(inp.float().index_add_(dim, idx, source.float()).to(dt))
    return cases


def sparse_sort_cases():
    """torch.sort(stable=True), Tensor.coalesce(), transposed coalesce, torch.sparse.mm on CPU."""
    cases = {}
    g = torch.Generator()
    # sort: tie-heavy inputs as benchmark_native_sort.py:95-97 builds them (dropout p=.9 -> exact zeros)
    for name, shape in (("1d", (5000,)), ("2d", (37, 53)), ("3d", (6, 11, 9))):
        g.manual_seed(42)
        x = torch.rand(shape, generator=g) * 2 - 1
        x = torch.where(torch.rand(shape, generator=g) < 0.9, torch.zeros(()), x)
        cases[f"sort_{name}_in"] = x.numpy()
        for dim in range(len(shape)):
            v, i = torch.sort(x, stable=True, dim=dim)
            cases[f"sort_{name}_d{dim}_values"] = v.numpy()
            cases[f"sort_{name}_d{dim}_indices"] = i.numpy()
    # coalesce / transpose: duplicated entries as benchmark_sparse_coalesce.py:129-159 builds them
    g.manual_seed(43)
    m, n, nnz = 40, 30, 500
    idx = torch.stack([torch.randint(0, m, (nnz,), generator=g), torch.randint(0, n, (nnz,), generator=g)])
    idx = torch.cat([idx, idx[:, :200]], dim=1)  # explicit duplicates
    val = torch.rand(idx.shape[1], generator=g)
    A = torch.sparse_coo_tensor(idx, val, (m, n))
    Ac = A.coalesce()
    At = A.t().coalesce()
    cases["coo_index"] = idx.numpy()
    cases["coo_value"] = val.numpy()
    cases["coo_mn"] = np.array([m, n], dtype=np.int64)
    cases["coalesce_index"] = Ac.indices().numpy()
    cases["coalesce_value"] = Ac.values().numpy()
    cases["transpose_index"] = At.indices().numpy()
    cases["transpose_value"] = At.values().numpy()
    # spmm: torch.sparse.mm(COO, dense) (benchmark_sparse_spmm.py:12-14); torch's summation order is its own,
    # so this vector is checked with a tolerance (1e-5 relative), not bit for bit
    B = torch.rand(n, 24, generator=g)
    cases["spmm_B"] = B.numpy()
    cases["spmm_out"] = torch.sparse.mm(A, B).numpy()
    # dense transpose copy (benchmark_sparse_transpose.py:13-16), fp16
    d = torch.rand(45, 70, generator=g).half()
    cases["dense_in"] = d.numpy()
    cases["dense_T"] = torch.transpose(d, 0, 1).contiguous().numpy()
    return cases


if __name__ == "__main__":
    torch.set_num_threads(1)
    np.savez_compressed(os.path.join(HERE, "scatter_golden.npz"), **scatter_cases())
    np.savez_compressed(os.path.join(HERE, "native_golden.npz"), **native_cases())
    np.savez_compressed(os.path.join(HERE, "sparse_sort_golden.npz"), **sparse_sort_cases())
    for f in ("scatter_golden.npz", "native_golden.npz", "sparse_sort_golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
