"""GPU parity for the kernels BASELINE configs 3 and 4 actually time (bench.py config3_leg / config4_leg).

Config 4 (fused index_select + sum, E=100M, D=128 fp16; reference op_bm_scripts/benchmark_fused_index_select_reduce.py:12-20)
runs `select_sum_rows_kernel` — the aligned-row branch of csrc/gather.hip launch_select_sum, taken when K % VEC == 0
(VEC = 8 for 16-bit, 4 for fp32). Config 3 (spmm over CSR 2M x 2M, nnz 40M, D=256 bf16 + GNN-shaped addmm; reference
op_bm_scripts/benchmark_sparse_spmm.py:12-14, benchmark_native_addmm.py:13-16) runs `spmm_rows_kernel<bf16>` through
`spmm_csr` with an int32 row pointer, and the 256^2-tile MFMA kernel with N = one tile column.

Reduced sizes are compared with the oracle; the full sizes through size-independent properties and exactly recomputed
sampled rows (the oracle cannot finish them in seconds):
  config 4   fused == double sum of the gathered rows (chunked on the device) within 1e-5; a permutation index sums the
             whole table; doubling the index doubles the sum; the unfused pair agrees
  config 3   sampled output rows recomputed sequentially in fp32 on the host, rounded once: BIT-exact; column checksum
             against a float64 evaluation of the same products; addmm sampled rows within the stated MFMA bound
"""
import numpy as np
import pytest
import torch

from helpers import TORCH_DT, assert_bits_equal, f32_of, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gnnops():
    import gnnops as g

    g.load_library()
    return g


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o

    return o


def _big_gpu():
    if torch.cuda.get_device_properties(0).total_memory < 100 * (1 << 30):
        pytest.skip("needs > 100 GB of HBM")


# ------------------------------------------------------------------------------------------------
# config 4: select_sum_rows_kernel (aligned rows)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dname", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("N,K,E", [(5000, 128, 20000), (700, 64, 3000), (300, 256, 1000), (257, 320, 999), (4000, 8, 50000),
                                   (64, 1024, 500), (33, 2048, 70), (1, 128, 9), (500, 128, 1)])
def test_index_select_sum_aligned_rows(gnnops, oracle, N, K, E, dname):
    """K % VEC == 0 -> select_sum_rows_kernel: one, a fraction of, or several 16-B chunks per lane group; repeated,
    unselected and single rows. fp32 partial sums against the oracle's double: 1e-5 relative."""
    g = torch.Generator().manual_seed(N + K)
    table = torch.rand(N, K, generator=g).to(TORCH_DT[dname])
    idx = torch.randint(0, N, (E,), generator=g)
    got = gnnops.index_select_sum(table.cuda(), 0, idx.cuda()).item()
    exp = oracle.index_select_sum(to_np(table), 0, idx.numpy(), dtype=dname)
    assert abs(got - exp) <= 1e-5 * abs(exp), (got, exp)


@pytest.mark.parametrize("dname", ["f16", "f32"])
def test_index_select_sum_aligned_rows_batched(gnnops, oracle, dname):
    """A batch in front of the indexed dim (B > 1), aligned rows behind it."""
    g = torch.Generator().manual_seed(8)
    t = torch.rand(3, 400, 128, generator=g).to(TORCH_DT[dname])
    idx = torch.randint(0, 400, (1500,), generator=g)
    got = gnnops.index_select_sum(t.cuda(), 1, idx.cuda()).item()
    exp = oracle.index_select_sum(to_np(t), 1, idx.numpy(), dtype=dname)
    assert abs(got - exp) <= 1e-5 * abs(exp), (got, exp)
    # signed values: cancellation must not hide a dropped or doubled row
    s = (torch.rand(2, 300, 64, generator=g) * 2 - 1).to(TORCH_DT[dname])
    idx = torch.randint(0, 300, (901,), generator=g)
    got = gnnops.index_select_sum(s.cuda(), 1, idx.cuda()).item()
    exp = oracle.index_select_sum(to_np(s), 1, idx.numpy(), dtype=dname)
    mag = oracle.index_select_sum(np.abs(f32_of(to_np(s), dname)), 1, idx.numpy(), dtype="f32")
    assert abs(got - exp) <= 1e-5 * mag, (got, exp)


def test_index_select_sum_exact_small_integers(gnnops):
    """Integers < 2^11 are exact in fp16 and their sums exact in fp32 below 2^24: the fused sum must be EXACT — a dropped,
    doubled or mis-addressed 16-B chunk cannot hide in a tolerance."""
    g = torch.Generator().manual_seed(3)
    N, K, E = 3000, 128, 4000
    table = torch.randint(0, 4, (N, K), generator=g).to(torch.float16)
    idx = torch.randint(0, N, (E,), generator=g)
    got = gnnops.index_select_sum(table.cuda(), 0, idx.cuda()).item()
    assert got == float(table[idx].double().sum()), got


def test_config4_full_size_fused_index_select_sum(gnnops):
    """BASELINE configs[3] at full size: E = N = 100M rows of D = 128 fp16 (25.6 GB table)."""
    _big_gpu()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(42)
    E, D = 100_000_000, 128
    table = torch.empty(E, D, device=dev, dtype=torch.float16).uniform_(0, 1, generator=g)
    index = torch.randint(0, E, (E,), generator=g, device=dev)
    fused = gnnops.index_select_sum(table, 0, index).item()

    def gathered_double_sum(idx):
        tot = 0.0
        for s in range(0, idx.numel(), 10_000_000):     # 2.56 GB gathered per chunk
            tot += float(table[idx[s:s + 10_000_000]].sum(dtype=torch.float64))
        return tot

    exp = gathered_double_sum(index)
    assert abs(fused - exp) <= 1e-5 * exp, (fused, exp)
    # the unfused pair the config compares with: our index_select (materialised) + an fp32-accumulated sum
    part = index[:20_000_000]
    unf = float(gnnops.index_select(table, 0, part).sum(dtype=torch.float32))
    fz = gnnops.index_select_sum(table, 0, part).item()
    assert abs(fz - unf) <= 1e-5 * unf, (fz, unf)
    # a permutation selects every row once: the sum of the whole table
    perm = torch.randperm(E, device=dev)
    whole = float(table.sum(dtype=torch.float64))
    got = gnnops.index_select_sum(table, 0, perm).item()
    assert abs(got - whole) <= 1e-5 * whole, (got, whole)
    del perm
    # selecting everything twice doubles it (fp32 partials: same tolerance, not bitwise)
    twice = gnnops.index_select_sum(table, 0, torch.cat([part, part])).item()
    assert abs(twice - 2 * fz) <= 2e-5 * fz
    # one hot row selected E/10 times: exactly representable check of the addressing under heavy reuse
    table[12345] = 1.0
    hot = torch.full((10_000_000,), 12345, device=dev, dtype=torch.int64)
    assert gnnops.index_select_sum(table, 0, hot).item() == pytest.approx(10_000_000 * 128, rel=1e-6)


# ------------------------------------------------------------------------------------------------
# config 3: spmm_csr with an int32 row pointer (bf16 / f16 / f32), and the GNN-shaped addmm
# ------------------------------------------------------------------------------------------------
def _csr_from_coo(row, m):
    order = torch.sort(row, stable=True).indices
    rowptr = torch.zeros(m + 1, dtype=torch.int64)
    rowptr[1:] = torch.bincount(row, minlength=m).cumsum(0)
    return order, rowptr


@pytest.mark.parametrize("dname", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("m,n,D,nnz", [(3000, 3000, 256, 60000), (500, 800, 128, 9000), (200, 100, 512, 3000),
                                       (1000, 1000, 64, 20000), (50, 60, 264, 700), (10, 10, 256, 0)])
def test_spmm_csr_int32_rowptr_bit_exact(gnnops, oracle, m, n, D, nnz, dname):
    """The entry bench.py's config-3 leg calls: CSR arrays, int32 rowptr, 16-bit values and features; bit-exact against
    the oracle (fp32 products and sums in nonzero order, one rounding)."""
    g = torch.Generator().manual_seed(m + D)
    row = torch.randint(0, m, (nnz,), generator=g)
    col = torch.randint(0, n, (nnz,), generator=g)
    if nnz:
        row[row == 7] = 8   # an empty row
    val = (torch.rand(nnz, generator=g) * 2 - 1).to(TORCH_DT[dname])
    B = (torch.rand(n, D, generator=g) * 2 - 1).to(TORCH_DT[dname])
    order, rowptr = _csr_from_coo(row, m)
    exp = oracle.spmm(torch.stack([row, col]).numpy(), to_np(val), m, n, to_np(B), dtype=dname)
    for rp in (rowptr.to(torch.int32), rowptr):
        got = gnnops.spmm_csr(rp.cuda(), col[order].cuda(), val[order].cuda(), B.cuda())
        assert_bits_equal(to_np(got), exp, f"spmm_csr {dname} rowptr {rp.dtype}")
    got = gnnops.spmm_csr(rowptr.to(torch.int32).cuda(), col[order].cuda(), None, B.cuda())
    exp1 = oracle.spmm(torch.stack([row, col]).numpy(), None, m, n, to_np(B), dtype=dname)
    assert_bits_equal(to_np(got), exp1, f"spmm_csr {dname} value=None")


def _bf16_round(x32):
    """fp32 -> bf16 -> fp32, round to nearest even (finite inputs)."""
    u = x32.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def test_config3_full_size_spmm_csr_bf16(gnnops):
    """BASELINE configs[2] at full size: CSR 2M x 2M, nnz 40M, D = 256 bf16, int32 rowptr (B = 1 GB, gathered 20.5 GB)."""
    _big_gpu()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(42)
    M, nnz, D = 2_000_000, 40_000_000, 256
    row = torch.randint(0, M, (nnz,), generator=g, device=dev).sort().values
    col = torch.randint(0, M, (nnz,), generator=g, device=dev)
    rowptr = torch.zeros(M + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.bincount(row, minlength=M).cumsum(0).to(torch.int32)
    val = (torch.rand(nnz, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    Bm = (torch.rand(M, D, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    out = gnnops.spmm_csr(rowptr, col, val, Bm)
    assert out.shape == (M, D) and out.dtype == torch.bfloat16
    # sampled rows, recomputed the oracle's way on the host: fp32 product, fp32 sequential sum, one rounding
    rp = rowptr.cpu().numpy()
    sample = [0, 1, 2, 77_777, M // 2, M - 2, M - 1] + torch.randint(0, M, (40,), generator=torch.Generator().manual_seed(1)).tolist()
    empties = np.nonzero(rp[1:] == rp[:-1])[0][:3].tolist()
    for r in sample + empties:
        lo, hi = int(rp[r]), int(rp[r + 1])
        acc = np.zeros(D, np.float32)
        if hi > lo:
            c = col[lo:hi]
            v = val[lo:hi].float().cpu().numpy()
            rows = Bm[c].float().cpu().numpy()
            for k in range(hi - lo):
                acc = acc + (v[k] * rows[k]).astype(np.float32)
        assert np.array_equal(out[r].float().cpu().numpy(), _bf16_round(acc)), f"row {r} ({hi - lo} nonzeros)"
    # column checksum over ALL rows against float64 products of the same operands; the outputs are rounded to bf16
    # (2^-9 relative each, independent), the sums have ~4e7 terms of mixed sign: compare against the sum of magnitudes
    chk = torch.zeros(D, dtype=torch.float64, device=dev)
    mag = torch.zeros(D, dtype=torch.float64, device=dev)
    for s in range(0, nnz, 4_000_000):
        p = Bm[col[s:s + 4_000_000]].double() * val[s:s + 4_000_000].double().unsqueeze(1)
        chk += p.sum(0)
        mag += p.abs().sum(0)
        del p
    got = out.double().sum(0)
    assert float(((got - chk).abs() / mag).max()) < 1e-4
    # value=None over the same structure == the same product with ones
    ones = torch.ones(nnz, dtype=torch.bfloat16, device=dev)
    assert torch.equal(gnnops.spmm_csr(rowptr, col, None, Bm), gnnops.spmm_csr(rowptr, col, ones, Bm))


def test_config3_addmm_gnn_shape_bf16(gnnops):
    """The GNN-shaped addmm of config 3: input[2M,256] + mat1[2M,256] @ mat2[256,256] bf16 — N is ONE tile column of the
    256 x 256 kernel, M = 7813 tile rows (the last one partial). Sampled rows against float64 with the stated bound of
    tests/test_gemm_fused_gpu.py: |err| <= 2^-8 |ref| + 4 K 2^-24 (|A| @ |B|)."""
    _big_gpu()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(7)
    M, K, N = 2_000_000, 256, 256
    A = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    W = (torch.rand(K, N, generator=g, device=dev) * 2 - 1 + torch.arange(N, device=dev).float().view(1, N) / N).to(torch.bfloat16)
    C = (torch.rand(M, N, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    rows = torch.cat([torch.arange(0, 300, device=dev), torch.arange(M - 300, M, device=dev),
                      torch.randint(0, M, (2000,), generator=g, device=dev),
                      torch.arange(999_936 - 128, 999_936 + 128, device=dev)])
    for inp in (C, None):
        got = gnnops.addmm(inp, A, W) if inp is not None else gnnops.matmul(A, W)
        assert got.shape == (M, N)
        ref = A[rows].double() @ W.double() + (inp[rows].double() if inp is not None else 0)
        bound = 2.0 ** -8 * ref.abs() + 4 * K * 2.0 ** -24 * (A[rows].double().abs() @ W.double().abs()) + 1e-30
        err = (got[rows].double() - ref).abs()
        assert bool((err <= bound).all()), f"max err/bound {(err / bound).max().item()}"
    # A = row selector: every output row must be exactly the selected row of W (integers exact in bf16)
    Wi = (torch.arange(K * N, device=dev).view(K, N) % 251).to(torch.bfloat16)
    sel = torch.randint(0, K, (M,), generator=g, device=dev)
    A.zero_()
    A[torch.arange(M, device=dev), sel] = 1
    assert torch.equal(gnnops.matmul(A, Wi), Wi[sel])
