"""GPU parity for the sort / sparse rows of SURVEY.md §8(a): a10 sort, a11 spmm, a13 coalesce, a14 transpose.
Bars: sort values+indices, coalesce/transpose indices and the dense transpose are bit-exact; spmm and the
coalesced values follow the oracle's sequential fp32 order and are required bit-exact against it (and within
1e-5 of torch's own CPU result in the golden file)."""
import numpy as np
import pytest
import torch

from helpers import TORCH_DT, assert_bits_equal, f32_of, from_np, load_golden, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gnnops():
    import gnnops as g

    g.load_library()
    g.set_plan_cache(False)
    yield g
    g.set_plan_cache(True)


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o

    return o


def test_golden_sparse_sort(gnnops):
    g = load_golden("sparse_sort_golden.npz")
    for name in ("1d", "2d", "3d"):
        x = torch.from_numpy(g[f"sort_{name}_in"]).cuda()
        for dim in range(x.dim()):
            for stable in (True, False):
                v, i = gnnops.sort(x, dim=dim, stable=stable)
                assert_bits_equal(v.cpu().numpy(), g[f"sort_{name}_d{dim}_values"], f"sort {name} d{dim} values")
                assert_bits_equal(i.cpu().numpy(), g[f"sort_{name}_d{dim}_indices"], f"sort {name} d{dim} indices")
    m, n = (int(v) for v in g["coo_mn"])
    idx = torch.from_numpy(g["coo_index"]).cuda()
    val = torch.from_numpy(g["coo_value"]).cuda()
    import torch_sparse

    ci, cv = torch_sparse.coalesce(idx, val, m, n)
    assert_bits_equal(ci.cpu().numpy(), g["coalesce_index"], "coalesce index")
    np.testing.assert_allclose(cv.cpu().numpy(), g["coalesce_value"], rtol=1e-6)
    ti, tv = torch_sparse.transpose(idx, val, m, n)
    assert_bits_equal(ti.cpu().numpy(), g["transpose_index"], "transpose index")
    np.testing.assert_allclose(tv.cpu().numpy(), g["transpose_value"], rtol=1e-6)
    out = torch_sparse.spmm(idx, val, m, n, torch.from_numpy(g["spmm_B"]).cuda())
    np.testing.assert_allclose(out.cpu().numpy(), g["spmm_out"], rtol=1e-5, atol=1e-6)
    dT = gnnops.transpose_contiguous(torch.from_numpy(g["dense_in"]).cuda())
    assert_bits_equal(dT.cpu().numpy(), g["dense_T"], "dense transpose")


@pytest.mark.parametrize("shape,dim", [((100_000,), 0), ((1 << 20,), 0), ((300, 1000), 1), ((300, 1000), 0),
                                       ((20, 50, 30), 1), ((20, 50, 30), 2), ((8193,), 0), ((1,), 0), ((3, 1), 1)])
@pytest.mark.parametrize("sparsity", [0.0, 0.9])
def test_sort_vs_oracle(gnnops, oracle, shape, dim, sparsity):
    g = torch.Generator().manual_seed(42)
    x = torch.rand(shape, generator=g) * 4 - 2
    if sparsity:
        x = torch.nn.functional.dropout(x, p=sparsity)  # exact zeros: massive ties (benchmark_native_sort.py:95-97)
    v, i = gnnops.sort(x.cuda(), dim=dim, stable=True)
    ev, ei = oracle.sort(x.numpy(), dim)
    assert_bits_equal(v.cpu().numpy(), ev, "values")
    assert_bits_equal(i.cpu().numpy(), ei, "indices")


def test_sort_special_values(gnnops, oracle):
    x = torch.tensor([0.0, -0.0, float("nan"), -1.0, float("inf"), -float("inf"), 0.0, 1e-40, -1e-40])
    v, i = gnnops.sort(x.cuda(), 0)
    ev, ei = oracle.sort(x.numpy(), 0)
    assert_bits_equal(i.cpu().numpy(), ei, "indices")
    assert torch.isnan(v[-1]) and np.array_equal(v[:-1].cpu().numpy(), ev[:-1])


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16, torch.float64])
@pytest.mark.parametrize("descending", [False, True])
def test_sort_values_keep_the_input_bits(gnnops, dt, descending):
    """values == input.gather(dim, indices) BIT FOR BIT, like torch.sort: -0.0 stays -0.0 (1 / value must not flip sign),
    NaNs keep sign and payload. Checked on the bit patterns, in every sort form (1-D passes, on-chip rows, rows in two
    halves + merge, segmented 64-bit keys, transposed dims)."""
    g = torch.Generator().manual_seed(31)
    bits = {torch.float32: torch.int32, torch.float16: torch.int16, torch.bfloat16: torch.int16, torch.float64: torch.int64}[dt]
    shapes = [(50_000,)] if dt == torch.float64 else [(50_000,), (40, 3000), (3, 30000), (6, 50, 40), (2, 45000)]
    for shape in shapes:
        x = (torch.randn(shape, generator=g) * 2).to(dt)
        r = torch.rand(shape, generator=g)
        x = torch.where(r < 0.2, torch.zeros((), dtype=dt), x)
        x = torch.where((r >= 0.2) & (r < 0.4), -torch.zeros((), dtype=dt), x)          # -0.0
        x = torch.where((r >= 0.4) & (r < 0.45), torch.full((), float("nan"), dtype=dt), x)
        xb = x.view(bits)
        nan_pos = torch.isnan(x)
        # NaNs with signs and payloads (low mantissa bits set; the sign bit on every other one)
        payload = torch.randint(1, 64, shape, generator=g).to(bits)
        sign = torch.where(torch.rand(shape, generator=g) < 0.5, torch.tensor(torch.iinfo(bits).min, dtype=bits), torch.zeros((), dtype=bits))
        x = torch.where(nan_pos, (xb | payload | sign).view(dt), x)
        for dim in range(len(shape)):
            v, i = gnnops.sort(x.cuda(), dim=dim, descending=descending, stable=True)
            ev, ei = torch.sort(x, dim=dim, descending=descending, stable=True)
            assert torch.equal(i.cpu(), ei), (dt, shape, dim)
            # gathered on the BIT view: torch's CPU gather / sort of 16-bit floats go through float and canonicalise NaNs
            assert torch.equal(v.cpu().view(bits), x.view(bits).gather(dim, ei)), (dt, shape, dim)
            if dt in (torch.float32, torch.float64):
                assert torch.equal(v.cpu().view(bits), ev.view(bits)), (dt, shape, dim)


@pytest.mark.parametrize("dname", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("m,n,D,nnz", [(1000, 800, 64, 20000), (300, 300, 256, 6000), (50, 70, 7, 900), (2000, 2000, 1, 30000),
                                       (64, 64, 320, 500), (40, 50, 257, 700), (30, 40, 1030, 500), (9, 9, 2049, 30)])
def test_spmm_bit_exact(gnnops, oracle, m, n, D, nnz, dname):
    g = torch.Generator().manual_seed(5)
    idx = torch.stack([torch.randint(0, m, (nnz,), generator=g), torch.randint(0, n, (nnz,), generator=g)])
    idx[0][idx[0] == 3] = 4  # row 3 empty
    val = (torch.rand(nnz, generator=g) * 2 - 1).to(TORCH_DT[dname])
    B = (torch.rand(n, D, generator=g) * 2 - 1).to(TORCH_DT[dname])
    got = gnnops.spmm(idx.cuda(), val.cuda(), m, n, B.cuda())
    exp = oracle.spmm(idx.numpy(), to_np(val), m, n, to_np(B), dtype=dname)
    assert_bits_equal(to_np(got), exp, f"spmm {dname}")
    if D > 1 and dname == "f32":
        got1 = gnnops.spmm(idx.cuda(), None, m, n, B.cuda())
        assert_bits_equal(to_np(got1), oracle.spmm(idx.numpy(), None, m, n, to_np(B), dtype=dname), "spmm value=None")
        A = torch.sparse_coo_tensor(idx, val, (m, n)).cuda()
        assert_bits_equal(to_np(gnnops.sparse_mm(A, B.cuda())), exp, "sparse_mm")
        # CSR entry point (config 3 layout): sorted rows, identity perm
        order = torch.sort(idx[0], stable=True).indices
        rowptr = torch.zeros(m + 1, dtype=torch.int64)
        rowptr[1:] = torch.bincount(idx[0], minlength=m).cumsum(0)
        got2 = gnnops.spmm_csr(rowptr.cuda(), idx[1][order].cuda(), val[order].cuda(), B.cuda())
        assert_bits_equal(to_np(got2), exp, "spmm_csr")


@pytest.mark.parametrize("dname", ["f32", "bf16"])
@pytest.mark.parametrize("m,n,nnz,dup", [(120, 100, 5000, 4), (12000 * 8, 12000 * 8, 40000, 8), (1, 1, 100, 1), (5, 7, 0, 1),
                                         (3_000_000, 1, 50000, 2),
                                         (65536, 65536, 30000, 3),    # row + column bits = 32: the last shape with 32-bit keys
                                         (65537, 65536, 30000, 3)])   # 33 bits: 64-bit keys
def test_coalesce_and_transpose(gnnops, oracle, m, n, nnz, dup, dname):
    g = torch.Generator().manual_seed(6)
    idx = torch.stack([torch.randint(0, m, (nnz,), generator=g), torch.randint(0, n, (nnz,), generator=g)])
    idx = torch.cat([idx] * dup, dim=1)
    idx = idx[:, torch.randperm(idx.shape[1], generator=g)]  # benchmark_sparse_coalesce.py:129-159 shuffles the index
    val = torch.rand(idx.shape[1], generator=g).to(TORCH_DT[dname])
    ci, cv = gnnops.coalesce(idx.cuda(), val.cuda(), m, n)
    ei, ev = oracle.coalesce(idx.numpy(), to_np(val), m, n, dtype=dname)
    assert_bits_equal(ci.cpu().numpy(), ei, "coalesce index")
    assert_bits_equal(to_np(cv), ev, "coalesce value")
    ti, tv = gnnops.transpose(idx.cuda(), val.cuda(), m, n)
    xi, xv = oracle.transpose_sparse(idx.numpy(), to_np(val), m, n, dtype=dname)
    assert_bits_equal(ti.cpu().numpy(), xi, "transpose index")
    assert_bits_equal(to_np(tv), xv, "transpose value")
    ci2, none = gnnops.coalesce(idx.cuda(), None, m, n)
    assert none is None and np.array_equal(ci2.cpu().numpy(), ei)
    if dname == "f32" and nnz and m * n < 1 << 40:
        A = torch.sparse_coo_tensor(idx, val, (m, n)).cuda()
        Ac = gnnops.coalesce_sparse_tensor(A)
        assert Ac.is_coalesced() and np.array_equal(Ac.indices().cpu().numpy(), ei)


@pytest.mark.parametrize("dname", ["f16", "f32"])
@pytest.mark.parametrize("R,C", [(2000, 2000), (7071, 333), (65, 129), (1, 500), (64, 64), (128, 128), (129, 131), (1001, 257),
                                 (257, 1001), (3001, 2999), (128, 4097)])
def test_dense_transpose(gnnops, oracle, R, C, dname):
    g = torch.Generator().manual_seed(8)
    x = torch.rand(R, C, generator=g).to(TORCH_DT[dname])
    got = gnnops.transpose_contiguous(x.cuda())
    assert_bits_equal(to_np(got), oracle.transpose_dense(to_np(x)), "transpose")


@pytest.mark.parametrize("R,C", [(1000, 300), (128, 64), (127, 500), (4100, 129), (257, 63)])
def test_converting_transposes(gnnops, R, C):
    """gnnops_transpose2d_cvt (int64 -> int32 and back) and gnnops_transpose2d_cvt_max (the same with the largest id left on the
    device): interior and edge tiles of the 128-row narrowing kernel, the 64 x 64 fallback below 128 rows, negative ids through
    the widening, the maximum wherever it sits."""
    from gnnops import _lib
    from gnnops.ops import _stream

    L = _lib.load()
    g = torch.Generator().manual_seed(R * 7 + C)
    x = torch.randint(0, 2 ** 31 - 1, (R, C), generator=g, dtype=torch.int64)
    for where in ((0, 0), (R - 1, C - 1), (R // 2, C // 3)):
        x[where] = 2 ** 31 - 1 - where[0]
        d = x.cuda()
        n32 = torch.empty((C, R), dtype=torch.int32, device="cuda")
        top = torch.empty(1, dtype=torch.int64, device="cuda")
        assert L.gnnops_transpose2d_cvt_max(d.data_ptr(), n32.data_ptr(), R, C, top.data_ptr(), _stream()) == 0
        assert int(top.item()) == int(x.max())
        assert torch.equal(n32.cpu(), x.t().contiguous().to(torch.int32))
        n32b = torch.empty_like(n32)
        assert L.gnnops_transpose2d_cvt(d.data_ptr(), n32b.data_ptr(), R, C, 0, _stream()) == 0
        assert torch.equal(n32b, n32)
        x[where] = 5
    y = torch.randint(-2 ** 31, 2 ** 31 - 1, (R, C), generator=g, dtype=torch.int64).to(torch.int32)
    w64 = torch.empty((C, R), dtype=torch.int64, device="cuda")
    assert L.gnnops_transpose2d_cvt(y.cuda().data_ptr(), w64.data_ptr(), R, C, 1, _stream()) == 0
    assert torch.equal(w64.cpu(), y.t().contiguous().to(torch.int64))


@pytest.mark.parametrize("mode", ["0", "1", "2", "3"])
def test_dense_transpose_4_byte_forms(gnnops, mode, monkeypatch):
    """csrc/sparse.hip transpose32_kernel: one element per lane, 8-B loads (128-column tiles), 8-B stores (128-row tiles) or both;
    the library picks by size, GNNOPS_T32 forces one. Odd row lengths (4-byte aligned 8-B accesses), edge tiles on both sides,
    a batch of matrices through the 3-D entry."""
    monkeypatch.setenv("GNNOPS_T32", mode)
    g = torch.Generator().manual_seed(9)
    for R, C in ((1001, 777), (128, 128), (4097, 129), (300, 5001)):
        x = torch.randint(-2 ** 31, 2 ** 31 - 1, (R, C), generator=g, dtype=torch.int64).to(torch.int32)
        assert torch.equal(gnnops.transpose_contiguous(x.cuda()).cpu(), x.t().contiguous()), (R, C)
    from gnnops import sparse
    x3 = torch.rand(5, 257, 131, generator=g)
    assert torch.equal(sparse._transpose_batched(x3.cuda()).cpu(), x3.transpose(1, 2).contiguous())


@pytest.mark.parametrize("dname", ["f32", "bf16"])
@pytest.mark.parametrize("m,k,n,nnzA,nnzB", [(60, 50, 70, 400, 500), (1414, 1414, 1414, 10000, 10000), (5, 4, 3, 0, 7),
                                             (8, 300, 9, 50, 2000)])
def test_spspmm(gnnops, oracle, m, k, n, nnzA, nnzB, dname):
    """sparsity .995 at L=1414 is the reference's first sweep point (benchmark_sparse_spspmm.py:28-30)."""
    g = torch.Generator().manual_seed(31)
    ia = torch.stack([torch.randint(0, m, (nnzA,), generator=g), torch.randint(0, k, (nnzA,), generator=g)])
    ib = torch.stack([torch.randint(0, k, (nnzB,), generator=g), torch.randint(0, n, (nnzB,), generator=g)])
    va = (torch.rand(nnzA, generator=g) * 2 - 1).to(TORCH_DT[dname])
    vb = (torch.rand(nnzB, generator=g) * 2 - 1).to(TORCH_DT[dname])
    import torch_sparse

    gi, gv = torch_sparse.spspmm(ia.cuda(), va.cuda(), ib.cuda(), vb.cuda(), m, k, n)
    ei, ev = oracle.spspmm(ia.numpy(), to_np(va), ib.numpy(), to_np(vb), m, k, n, dtype=dname)
    assert_bits_equal(gi.cpu().numpy(), ei, "spspmm index")
    assert_bits_equal(to_np(gv), ev, "spspmm value")
    if dname == "f32" and nnzA:
        A = torch.sparse_coo_tensor(ia, va, (m, k)).coalesce()
        Bs = torch.sparse_coo_tensor(ib, vb, (k, n)).coalesce()
        ref = torch.sparse.mm(A, Bs).coalesce()  # torch's own CPU SpGEMM: same pattern, values within fp32 tolerance
        got = gnnops.sparse_mm(A.cuda(), Bs.cuda())
        assert got.is_coalesced() and np.array_equal(got.indices().cpu().numpy(), ref.indices().numpy())
        np.testing.assert_allclose(got.values().cpu().numpy(), ref.values().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16, torch.int32, torch.int64, torch.float64])
@pytest.mark.parametrize("descending", [False, True])
def test_sort_dtypes_and_descending(gnnops, dt, descending):
    """Against torch.sort(stable=True) on the CPU (the reference's own op): values and indices bit-exact."""
    g = torch.Generator().manual_seed(77)
    shapes = [(70_001,)] if dt in (torch.int64, torch.float64) else [(70_001,), (130, 257), (9, 40, 11)]
    for shape in shapes:
        if dt.is_floating_point:
            x = (torch.randn(shape, generator=g) * 3).to(dt)
            x = torch.where(torch.rand(shape, generator=g) < 0.3, torch.zeros((), dtype=dt), x)  # ties
        else:
            x = torch.randint(-1000, 1000, shape, generator=g).to(dt)
        for dim in range(len(shape)):
            ev, ei = torch.sort(x, dim=dim, descending=descending, stable=True)
            v, i = gnnops.sort(x.cuda(), dim=dim, descending=descending, stable=True)
            assert torch.equal(i.cpu(), ei), (dt, shape, dim, descending)
            assert torch.equal(v.cpu(), ev), (dt, shape, dim, descending)
    big = torch.tensor([2**62, -2**63, 0, -1, 2**63 - 1, 5], dtype=torch.int64)
    v, i = gnnops.sort(big.cuda(), descending=descending)
    ev, ei = torch.sort(big, descending=descending, stable=True)
    assert torch.equal(v.cpu(), ev) and torch.equal(i.cpu(), ei)


@pytest.mark.parametrize("rows,E", [(300, 1000), (50, 4096), (40, 8000), (20, 16384), (6, 20000), (3, 22528), (5000, 33), (7, 1025),
                                    (5, 22529), (4, 28200), (3, 32769), (2, 40000), (2, 40001)])  # two halves + rank merge; the last: HBM passes
@pytest.mark.parametrize("descending", [False, True])
def test_sort_rows_on_chip(gnnops, rows, E, descending):
    """Rows that fit in LDS take the on-chip kernel (last dim), and dim 0 of a matrix goes through our transposes:
    both bit-exact against torch.sort(stable=True) on the CPU, tie-heavy inputs included."""
    g = torch.Generator().manual_seed(rows * 7 + E)
    x = torch.nn.functional.dropout(torch.randn(rows, E, generator=g), p=0.6)
    ev, ei = torch.sort(x, dim=1, descending=descending, stable=True)
    v, i = gnnops.sort(x.cuda(), dim=1, descending=descending, stable=True)
    assert torch.equal(i.cpu(), ei), "indices"
    assert torch.equal(v.cpu(), ev), "values"
    xt = x.t().contiguous()                       # sort along dim 0 of the transposed matrix
    v0, i0 = gnnops.sort(xt.cuda(), dim=0, descending=descending, stable=True)
    assert torch.equal(i0.cpu(), ei.t()) and torch.equal(v0.cpu(), ev.t())


def test_sort_nd_non_last_dim_on_chip(gnnops, oracle):
    """3-D, sorting along dims 0 and 1: batched transposes + on-chip rows, bit-exact vs the oracle."""
    g = torch.Generator().manual_seed(3)
    x = torch.nn.functional.dropout(torch.randn(40, 70, 33, generator=g), p=0.5)
    for dim in (0, 1, 2):
        v, i = gnnops.sort(x.cuda(), dim=dim, stable=True)
        ev, ei = oracle.sort(x.numpy(), dim)
        assert_bits_equal(v.cpu().numpy(), ev, f"values d{dim}")
        assert_bits_equal(i.cpu().numpy(), ei, f"indices d{dim}")


@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_spmm_hub_rows(gnnops, oracle, dname):
    """Output rows with more than 8192 nonzeros (csrc/hub.h) are multiplied out piecewise by whole workgroups: re-associated
    fp32 sums for those rows (tolerance), every other row bit-exact; COO (plan) and CSR entry points."""
    m, n, D, nnz = 500, 400, 64, 90000
    g = torch.Generator().manual_seed(8)
    row = torch.randint(0, m, (nnz,), generator=g)
    r = torch.rand(nnz, generator=g)
    row[r < 0.4] = 7
    row[(r >= 0.4) & (r < 0.55)] = 300
    col = torch.randint(0, n, (nnz,), generator=g)
    idx = torch.stack([row, col])
    val = (torch.rand(nnz, generator=g) * 2 - 1).to(TORCH_DT[dname])
    B = (torch.rand(n, D, generator=g) * 2 - 1).to(TORCH_DT[dname])
    exp = oracle.spmm(idx.numpy(), to_np(val), m, n, to_np(B), dtype=dname)
    hubs = np.bincount(row.numpy(), minlength=m) > 8192
    assert hubs.sum() == 2
    order = torch.sort(row, stable=True).indices
    rowptr = torch.zeros(m + 1, dtype=torch.int64)
    rowptr[1:] = torch.bincount(row, minlength=m).cumsum(0)
    for got in (gnnops.spmm(idx.cuda(), val.cuda(), m, n, B.cuda()),
                gnnops.spmm_csr(rowptr.cuda(), col[order].cuda(), val[order].cuda(), B.cuda())):
        gf = f32_of(to_np(got), dname) if dname != "f32" else to_np(got)
        ef = f32_of(exp, dname) if dname != "f32" else exp
        assert np.array_equal(to_np(got)[~hubs], exp[~hubs])
        tol = 2e-3 if dname == "f32" else 0.5
        np.testing.assert_allclose(gf[hubs], ef[hubs], rtol=1e-2 if dname != "f32" else 2e-4, atol=tol)
