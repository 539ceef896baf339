"""GPU parity: a17 addmm / matmul (MFMA GEMM) and a16 fused index_add + index_select + sum.

addmm tolerance (stated, north_star "stated fp tolerance"): the device accumulates K products in fp32 inside the
MFMA and rounds once to the 16-bit output, so |got - exact| <= 2^-8 (bf16) / 2^-11 (fp16) relative rounding of
the result plus fp32 accumulation error ~ K * 2^-24 * sum|a*b|. Checked as
    |got - ref64| <= eps_out * |ref64| + 4 * K * 2^-24 * (|A| @ |B|).
fused: fp32 partial sums vs a double reference, relative 1e-5 of the sum of magnitudes.
"""
import numpy as np
import pytest
import torch

from helpers import TORCH_DT, f32_of, to_np

pytestmark = pytest.mark.gpu
EPS_OUT = {"f16": 2.0 ** -11, "bf16": 2.0 ** -8}


@pytest.fixture(scope="module")
def gnnops():
    import gnnops as g

    g.load_library()
    return g


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o

    return o


@pytest.mark.parametrize("dname", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 512), (130, 70, 100), (1, 1, 1), (17, 9, 40), (333, 257, 191),
                                   (64, 64, 8), (512, 256, 1024)])
def test_addmm_matmul(gnnops, oracle, M, N, K, dname):
    g = torch.Generator().manual_seed(3)
    # asymmetric operands (row/col swaps in the C write would show)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(TORCH_DT[dname])
    B = (torch.rand(K, N, generator=g) * 2 - 1 + torch.arange(N).float().view(1, N) / max(N, 1)).to(TORCH_DT[dname])
    C = (torch.rand(M, N, generator=g) * 2 - 1).to(TORCH_DT[dname])
    bound_acc = 4 * K * 2.0 ** -24 * (A.double().abs() @ B.double().abs()).numpy()
    for inp in (C, None):
        got = gnnops.addmm(inp.cuda(), A.cuda(), B.cuda()) if inp is not None else gnnops.matmul(A.cuda(), B.cuda())
        ref = oracle.addmm(None if inp is None else to_np(inp), to_np(A), to_np(B), dtype=dname)
        err = np.abs(f32_of(to_np(got), dname).astype(np.float64) - ref)
        bound = EPS_OUT[dname] * np.abs(ref) + bound_acc + 1e-30
        assert (err <= bound).all(), f"max err/bound {np.max(err / bound)} at {np.unravel_index(np.argmax(err / bound), err.shape)}"


def test_addmm_identity_layout(gnnops):
    """A = I with an asymmetric B: catches fragment-layout and transposed-read mistakes exactly."""
    for n in (192, 256, 384):  # 192: register-staged kernel; 256 / 384: tile-aligned -> LDS-DMA kernel
        B = (torch.arange(n * n).view(n, n) % 251).to(torch.bfloat16)  # integers < 256 are exact in bf16
        I = torch.eye(n, dtype=torch.bfloat16)
        assert torch.equal(gnnops.matmul(I.cuda(), B.cuda()).cpu(), B), n
        assert torch.equal(gnnops.matmul(B.cuda(), I.cuda()).cpu(), B), n
        assert torch.equal(gnnops.addmm(B.cuda(), I.cuda(), B.cuda()).cpu(), (B.float() * 2).to(torch.bfloat16)), n


@pytest.mark.parametrize("dname", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(4096, 4096, 64), (4096, 4096, 192), (8192, 2048, 128), (2048, 8192, 320),
                                   (4099, 4101, 300), (4090, 4092, 263), (1581, 1581, 1581), (700, 900, 257), (513, 520, 512),
                                   (3000, 3100, 264), (2816, 2816, 128)])
def test_addmm_big_tiles(gnnops, M, N, K, dname):
    """The LDS-DMA kernels: >= 128 tiles of 256 x 256 take the eight-wave ping-pong kernel (K = 64: fewer K-steps than pipeline
    stages), smaller grids the 128 x 128 one; odd M / N / K exercise the filler rows, the zero-padded K tail and the
    8-B / per-element epilogue stores. Same bound as test_addmm_matmul against a float64 product of the same 16-bit
    operands."""
    g = torch.Generator().manual_seed(5)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(TORCH_DT[dname])
    B = (torch.rand(K, N, generator=g) * 2 - 1 + torch.arange(N).float().view(1, N) / N).to(TORCH_DT[dname])
    C = (torch.rand(M, N, generator=g) * 2 - 1).to(TORCH_DT[dname])
    bound_acc = 4 * K * 2.0 ** -24 * (A.double().abs() @ B.double().abs())
    prod = A.double() @ B.double()
    for inp in (C, None):
        got = gnnops.addmm(inp.cuda(), A.cuda(), B.cuda()) if inp is not None else gnnops.matmul(A.cuda(), B.cuda())
        ref = prod + (inp.double() if inp is not None else 0)
        err = (got.cpu().double() - ref).abs()
        bound = EPS_OUT[dname] * ref.abs() + bound_acc + 1e-30
        assert bool((err <= bound).all()), f"max err/bound {(err / bound).max().item()}"


@pytest.mark.parametrize("dname", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(4352, 4352, 192), (4352, 4096, 264), (4864, 4864, 128), (4249, 4249, 263),
                                   (4100, 4600, 320), (4352, 4352, 2048), (512, 37000, 320), (37000, 512, 328)])
def test_addmm_split_k_tail(gnnops, M, N, K, dname):
    """More than one round of 256 x 256 tiles on the 256 CUs with a last round of at most half a round: the persistent kernel
    (gemm_sk256_kernel) cuts the last round's tiles along K into 8 / 4 / 2 pieces (272 tiles: 16 left, 289: 33, 361: 105;
    K = 264 -> 10 K-steps: pieces of one or two steps), partners exchange fp32 partial tiles through the workspace. Odd sizes put
    edge tiles into the tail and run the pad copies of odd-length rows; 2 x 145 and 145 x 2 tiles walk short and
    single-row bands of the tile order. Same bound as test_addmm_big_tiles; also equal
    to the plain-grid kernel's result up to one rounding of the output type (the pieces are summed in a different order)."""
    import os
    g = torch.Generator().manual_seed(6)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(TORCH_DT[dname])
    B = (torch.rand(K, N, generator=g) * 2 - 1 + torch.arange(N).float().view(1, N) / N).to(TORCH_DT[dname])
    C = (torch.rand(M, N, generator=g) * 2 - 1).to(TORCH_DT[dname])
    L = gnnops._lib.load()
    plan_ws = L.gnnops_addmm_workspace_bytes(M, N, K)
    os.environ["GNNOPS_GEMM_SK"] = "0"
    try:
        assert plan_ws > L.gnnops_addmm_workspace_bytes(M, N, K) + (200 << 20) // 4, "the split-K path was not planned"
        plain = gnnops.addmm(C.cuda(), A.cuda(), B.cuda()).cpu()
    finally:
        del os.environ["GNNOPS_GEMM_SK"]
    bound_acc = 4 * K * 2.0 ** -24 * (A.double().abs() @ B.double().abs())
    prod = A.double() @ B.double()
    for inp in (C, None):
        got = gnnops.addmm(inp.cuda(), A.cuda(), B.cuda()) if inp is not None else gnnops.matmul(A.cuda(), B.cuda())
        ref = prod + (inp.double() if inp is not None else 0)
        err = (got.cpu().double() - ref).abs()
        bound = EPS_OUT[dname] * ref.abs() + bound_acc + 1e-30
        assert bool((err <= bound).all()), f"max err/bound {(err / bound).max().item()}"
        if inp is not None:
            ulp = 2 * EPS_OUT[dname] * plain.double().abs() + 1e-30
            assert bool(((got.cpu().double() - plain.double()).abs() <= ulp + 2 * bound_acc).all())


def test_addmm_split_k_tail_identity(gnnops):
    """I @ B and B @ I with a split last round: every piece but one adds exact zeros, the result is B bit for bit."""
    n = 4352
    B = (torch.arange(n * n).view(n, n) % 251).to(torch.bfloat16)
    I = torch.eye(n, dtype=torch.bfloat16)
    assert torch.equal(gnnops.matmul(I.cuda(), B.cuda()).cpu(), B)
    assert torch.equal(gnnops.matmul(B.cuda(), I.cuda()).cpu(), B)
    for _ in range(3):   # the flags are cleared per launch: back-to-back calls sharing a workspace address
        assert torch.equal(gnnops.addmm(B.cuda(), I.cuda(), B.cuda()).cpu(), (B.float() * 2).to(torch.bfloat16))


def test_addmm_split_k_tail_reuses_its_workspace(gnnops):
    """Back-to-back calls get the same workspace addresses from the allocator, with different operands each time: a flag
    or a partial tile that survived from the previous call would put a partner's OLD partial into the sum — an error of
    whole units against the plain-grid kernel's result, where a correct sum differs by one rounding of the output."""
    import os
    M = N = 4352
    K = 1024
    g = torch.Generator(device="cuda").manual_seed(12)
    b = (torch.rand(K, N, generator=g, device="cuda") - 0.5).half()
    c = (torch.rand(M, N, generator=g, device="cuda") - 0.5).half()
    for _ in range(8):
        a = (torch.rand(M, K, generator=g, device="cuda") - 0.5).half()
        got = gnnops.addmm(c, a, b)
        os.environ["GNNOPS_GEMM_SK"] = "0"
        try:
            plain = gnnops.addmm(c, a, b)
        finally:
            del os.environ["GNNOPS_GEMM_SK"]
        assert (got.float() - plain.float()).abs().max().item() <= 0.02


@pytest.mark.parametrize("M,N,K", [(513, 517, 333), (600, 700, 257), (1023, 515, 1001)])
def test_addmm_pad_copies_of_odd_rows(gnnops, M, N, K):
    """pad_rows_kernel<2>: rows of odd length are read as whole dwords and shifted; the last row's last piece must not
    read past the matrix and elements past a row's end must come out zero (they would otherwise carry the next row's
    head into the K tail). Integer-valued operands: the product is exact, any leak shows."""
    g = torch.Generator().manual_seed(8)
    A = torch.randint(-2, 3, (M, K), generator=g).to(torch.float16)
    B = torch.randint(-2, 3, (K, N), generator=g).to(torch.float16)
    ref = (A.double() @ B.double())
    assert ref.abs().max() < 2048
    # operands at the very end of their allocations: a read past the last row would fault or pick up the guard values
    a_buf = torch.full((M * K + 64,), 7.0, dtype=torch.float16, device="cuda")
    b_buf = torch.full((K * N + 64,), 7.0, dtype=torch.float16, device="cuda")
    a = a_buf[:M * K].view(M, K); a.copy_(A)
    b = b_buf[:K * N].view(K, N); b.copy_(B)
    got = gnnops.matmul(a, b).cpu().double()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("dname", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(1581, 1581, 1581), (700, 900, 257), (513, 520, 512), (1030, 1027, 320), (4249, 4249, 263),
                                   (3000, 3100, 264), (2050, 4100, 1000), (515, 519, 4101)])
def test_addmm_operands_read_in_place(gnnops, M, N, K, dname):
    """The LDS-DMA kernels read rows of any alignment where they lie and take only the LAST K-tile from zero-filled side
    copies (K % 64 != 0, or N % 8 != 0 where a B row's last 16-B piece would run past the matrix). The products and their
    order are those of the whole-matrix padded copies (GNNOPS_GEMM_PAD=full, the earlier path): bit-identical results.
    Operands sit at the very end of their allocations with a NaN guard behind them: a read past the end that reached an
    output would show."""
    import os
    g = torch.Generator().manual_seed(9)
    dt = TORCH_DT[dname]
    def at_end(rows, cols):
        buf = torch.full((rows * cols + 40,), float("nan"), dtype=dt, device="cuda")
        v = buf[:rows * cols].view(rows, cols)
        v.copy_((torch.rand(rows, cols, generator=g) * 2 - 1).to(dt))
        return v
    A, B, C = at_end(M, K), at_end(K, N), at_end(M, N)
    got = gnnops.addmm(C, A, B)
    os.environ["GNNOPS_GEMM_PAD"] = "full"
    try:
        ref = gnnops.addmm(C, A, B)
    finally:
        del os.environ["GNNOPS_GEMM_PAD"]
    assert not torch.isnan(got).any()
    assert torch.equal(got, ref)
    rows = torch.randint(0, M, (32,), generator=g).cuda()
    ref64 = C[rows].double() + A[rows].double() @ B.double()
    err = (got[rows].double() - ref64).abs()
    bound = EPS_OUT[dname] * ref64.abs() + 4 * K * 2.0 ** -24 * (A[rows].double().abs() @ B.double().abs()) + 1e-30
    assert bool((err <= bound).all())


@pytest.mark.parametrize("M,N,K", [(6000, 5200, 5100), (1536, 1536, 520), (4352, 4352, 1024)])
def test_addmm_operands_at_odd_offsets(gnnops, M, N, K):
    """Operands that start 2 bytes into their allocation (a slice of a flat buffer): the LDS-DMA kernels read them where they lie,
    with A copied whole (first shape: > 30 M elements), with both in place, and through the split-K kernel. Against the same
    product from 256-B aligned copies of the operands: bit-identical."""
    g = torch.Generator(device="cuda").manual_seed(15)
    def odd(rows, cols):
        buf = torch.empty(rows * cols + 9, dtype=torch.float16, device="cuda")
        v = buf[1:1 + rows * cols].view(rows, cols)
        v.copy_((torch.rand(rows, cols, generator=g, device="cuda") - 0.5).half())
        assert v.data_ptr() % 4 == 2
        return v
    A, B, C = odd(M, K), odd(K, N), odd(M, N)
    got = gnnops.addmm(C, A, B)
    ref = gnnops.addmm(C.clone(), A.clone(), B.clone())
    assert torch.equal(got, ref)


def test_addmm_big_tiles_identity(gnnops):
    n = 4096
    B = (torch.arange(n * n).view(n, n) % 251).to(torch.bfloat16)
    I = torch.eye(n, dtype=torch.bfloat16)
    assert torch.equal(gnnops.matmul(I.cuda(), B.cuda()).cpu(), B)
    assert torch.equal(gnnops.matmul(B.cuda(), I.cuda()).cpu(), B)


def test_addmm_reference_shape_fp16(gnnops):
    """The reference's first sweep length (benchmark_native_addmm.py:23-38: L = 1581, fp16): odd, unaligned rows."""
    L = 1581
    g = torch.Generator().manual_seed(4)
    A = torch.rand(L, L, generator=g).half()
    B = torch.rand(L, L, generator=g).half()
    C = torch.rand(L, L, generator=g).half()
    got = gnnops.addmm(C.cuda(), A.cuda(), B.cuda()).float().cpu()
    ref = (C.double() + A.double() @ B.double())
    rel = ((got.double() - ref).abs() / ref.abs()).max().item()
    assert rel <= 2.0 ** -10, rel


@pytest.mark.parametrize("dname", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("L", [300, 1000])
def test_fused_index_add_select_sum(gnnops, oracle, L, dname):
    g = torch.Generator().manual_seed(21)
    x = torch.rand(L, L, generator=g).to(TORCH_DT[dname])
    for dim in (0, 1):
        idx = torch.randint(0, L, (L,), generator=g)
        got = gnnops.index_add_select_sum(x.cuda(), dim, idx.cuda(), x.clone().cuda()).cpu().numpy().astype(np.float64)
        ref = oracle.index_add_select_sum(to_np(x), dim, idx.numpy(), to_np(x), dtype=dname)
        assert got.shape == ref.shape == (L,)
        assert np.max(np.abs(got - ref) / np.abs(ref).max()) <= 1e-5


def test_aten_sort_and_addmm_overrides(gnnops):
    g = torch.Generator().manual_seed(23)
    x = torch.nn.functional.dropout(torch.rand(200, 300, generator=g), p=0.9)
    A = torch.rand(96, 64, generator=g).half()
    B = torch.rand(64, 80, generator=g).half()
    C = torch.rand(96, 80, generator=g).half()
    ev, ei = torch.sort(x, dim=1, stable=True)
    gnnops.install()
    try:
        v, i = torch.sort(x.cuda(), dim=1, stable=True)
        v2, i2 = torch.sort(x.cuda(), 0)
        out = torch.addmm(C.cuda(), A.cuda(), B.cuda())
        mm = torch.matmul(A.cuda(), B.cuda())
    finally:
        gnnops.uninstall()
    assert torch.equal(v.cpu(), ev) and torch.equal(i.cpu(), ei)
    assert torch.equal(v2.cpu(), torch.sort(x, dim=0, stable=True).values)
    ref = C.double() + A.double() @ B.double()
    assert ((out.double().cpu() - ref).abs() / ref.abs()).max() <= 2.0 ** -10
    assert ((mm.double().cpu() - A.double() @ B.double()).abs().max()) <= 2.0 ** -10 * 64


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (256, 384, 512), (130, 70, 100), (1, 1, 1), (17, 9, 40), (333, 257, 191)])
def test_addmm_fp32(gnnops, M, N, K):
    """fp32 operands run on v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 sums in k order.
    Tolerance: |err| <= 2 * K * 2^-24 * (|A| @ |B| + |C|) against a float64 reference."""
    g = torch.Generator().manual_seed(33)
    A = torch.rand(M, K, generator=g) * 2 - 1
    B = torch.rand(K, N, generator=g) * 2 - 1 + torch.arange(N).float().view(1, N) / max(N, 1)
    C = torch.rand(M, N, generator=g) * 2 - 1
    got = gnnops.addmm(C.cuda(), A.cuda(), B.cuda()).cpu().double()
    ref = C.double() + A.double() @ B.double()
    bound = 2 * max(K, 1) * 2.0 ** -24 * (A.double().abs() @ B.double().abs() + C.double().abs()) + 1e-30
    assert ((got - ref).abs() <= bound).all(), ((got - ref).abs() / bound).max()
    n = 160
    Bi = (torch.arange(n * n).view(n, n) % 1009).float()
    assert torch.equal(gnnops.matmul(torch.eye(n).cuda(), Bi.cuda()).cpu(), Bi)
    assert torch.equal(gnnops.matmul(Bi.cuda(), torch.eye(n).cuda()).cpu(), Bi)


@pytest.mark.parametrize("variant", ["", "1", "4"])
@pytest.mark.parametrize("M,N,K", [(3072, 3072, 64), (4096, 2048, 512), (2905, 3332, 272), (4099, 2820, 48), (8192, 4096, 32),
                                   (3000, 3100, 1040)])
def test_addmm_fp32_big_tiles(gnnops, monkeypatch, M, N, K, variant):
    """>= 128 tiles of 256 x 256, K a multiple of 16, N a multiple of 4: the LDS-DMA fp32 kernel (gemm_f32_dma256_kernel) —
    XCD-contiguous strip order (a last strip narrower than 8 tiles), filler rows and columns, K-steps 2 .. 65, the k
    permutation inside a K-step. Same bound as test_addmm_fp32; products with an identity and a row selector exact; and
    bit-identical results from the 128 x 128 kernel would be too strong (different order of the sum over k)."""
    if variant:   # GNNOPS_GEMM_F32_BIG: "" = default, 1 = eight waves in lockstep, 4 = four waves of 128 x 128 (one per SIMD)
        monkeypatch.setenv("GNNOPS_GEMM_F32_BIG", variant)
    else:
        monkeypatch.delenv("GNNOPS_GEMM_F32_BIG", raising=False)
    g = torch.Generator().manual_seed(34)
    A = torch.rand(M, K, generator=g) * 2 - 1
    B = torch.rand(K, N, generator=g) * 2 - 1 + torch.arange(N).float().view(1, N) / N
    C = torch.rand(M, N, generator=g) * 2 - 1
    bound = 2 * K * 2.0 ** -24 * (A.double().abs() @ B.double().abs() + C.double().abs()) + 1e-30
    for inp in (C, None):
        got = (gnnops.addmm(inp.cuda(), A.cuda(), B.cuda()) if inp is not None else gnnops.matmul(A.cuda(), B.cuda())).cpu().double()
        ref = A.double() @ B.double() + (inp.double() if inp is not None else 0)
        assert ((got - ref).abs() <= bound).all(), ((got - ref).abs() / bound).max()
    # every k and every (row, column) position lands where it should: a selector matrix picks rows of an integer matrix
    Kb = 1040 if K == 1040 else 256
    Bi = (torch.arange(Kb * N).view(Kb, N) % 1009).float()
    sel = torch.randint(0, Kb, (M,), generator=g)
    S = torch.zeros(M, Kb)
    S[torch.arange(M), sel] = 1
    assert torch.equal(gnnops.matmul(S.cuda(), Bi.cuda()).cpu(), Bi[sel])


def test_aten_sparse_and_scatter_reduce_overrides(gnnops, monkeypatch):
    """torch.sparse.mm, Tensor.coalesce() and Tensor.scatter_(reduce=...) reach our kernels when the build accepts the
    SparseCUDA registrations (otherwise the by-name entry points remain): verified by counting calls."""
    from gnnops import aten, sparse as sp, ops

    calls = {"mm": 0, "coalesce": 0, "mul": 0}
    real_mm, real_co, real_mul = sp.sparse_mm, sp.coalesce_sparse_tensor, ops.scatter_reduce_mul_
    monkeypatch.setattr(sp, "sparse_mm", lambda a, b: (calls.__setitem__("mm", calls["mm"] + 1), real_mm(a, b))[1])
    monkeypatch.setattr(sp, "coalesce_sparse_tensor", lambda a: (calls.__setitem__("coalesce", calls["coalesce"] + 1), real_co(a))[1])
    monkeypatch.setattr(ops, "scatter_reduce_mul_", lambda *a: (calls.__setitem__("mul", calls["mul"] + 1), real_mul(*a))[1])
    g = torch.Generator().manual_seed(29)
    dense = torch.nn.functional.dropout(torch.rand(200, 150, generator=g), p=0.9)
    B = torch.rand(150, 32, generator=g)
    idx = torch.stack([torch.randint(0, 200, (900,), generator=g), torch.randint(0, 150, (900,), generator=g)])
    val = torch.rand(900, generator=g)
    src = torch.rand(64, 48, generator=g)
    full = torch.randint(0, 48, (64, 48), generator=g)
    gnnops.install()
    try:
        A = dense.cuda().to_sparse()
        out = torch.sparse.mm(A, B.cuda())
        Bs = torch.nn.functional.dropout(torch.rand(150, 90, generator=g), p=0.9).cuda().to_sparse()
        ss = torch.sparse.mm(A, Bs)
        unc = torch.sparse_coo_tensor(idx.cuda(), val.cuda(), (200, 150))
        co = unc.coalesce()
        temp = torch.zeros_like(src).cuda()
        temp.scatter_(-1, full.cuda(), src.cuda(), reduce="multiply")
    finally:
        gnnops.uninstall()
    np.testing.assert_allclose(out.cpu().numpy(), (dense @ B).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ss.to_dense().cpu().numpy(), (dense @ Bs.to_dense().cpu()).numpy(), rtol=1e-5, atol=1e-5)
    ref = torch.sparse_coo_tensor(idx, val, (200, 150)).coalesce()
    assert torch.equal(co.indices().cpu(), ref.indices())
    np.testing.assert_allclose(co.values().cpu().numpy(), ref.values().numpy(), rtol=1e-6)
    assert torch.count_nonzero(temp).item() == 0 and calls["mul"] == 1
    print("routed:", sorted(aten.routed_ops), calls)
    if "addmm@SparseCUDA" in aten.routed_ops:
        assert calls["mm"] >= 1
    if "_sparse_sparse_matmul@SparseCUDA" in aten.routed_ops:
        assert calls["mm"] >= 2
    if "_coalesce@SparseCUDA" in aten.routed_ops:
        assert calls["coalesce"] >= 1


@pytest.mark.parametrize("M,N,K", [(700, 900, 257), (3000, 3100, 264), (1581, 1581, 1581), (4099, 4101, 300), (4096, 4100, 256),
                                   (2052, 4096, 1000)])
def test_addmm_rows_and_columns_do_not_leak(gnnops, M, N, K):
    """Odd shapes reach the LDS-DMA kernels through zero-padded K tails, filler rows and clamped column chunks, and 0 x Inf
    or 0 x NaN would poison valid outputs if any of that leaked: a NaN row of A may touch only that output row, an Inf
    column of B only that output column — including the LAST row / column and the first column (the neighbour in memory of
    the previous row's last one) — and everything else must match the float64 product."""
    g = torch.Generator().manual_seed(9)
    A = (torch.rand(M, K, generator=g) * 2 - 1).half()
    B = (torch.rand(K, N, generator=g) * 2 - 1).half()
    A[3, :] = float("nan")
    A[M - 1, :] = float("nan")
    B[:, 0] = float("inf")       # first column: sits right behind column N-1 of the previous row in memory
    B[:, N - 1] = float("inf")
    got = gnnops.matmul(A.cuda(), B.cuda()).cpu().double()
    rows = torch.ones(M, dtype=torch.bool)
    rows[[3, M - 1]] = False
    cols = torch.ones(N, dtype=torch.bool)
    cols[[0, N - 1]] = False
    assert bool(torch.isnan(got[~rows]).all())
    assert bool((~torch.isfinite(got[:, ~cols])).all())
    clean = got[rows][:, cols]
    assert bool(torch.isfinite(clean).all())
    Ac, Bc = A[rows].double(), B[:, cols].double()
    ref = Ac @ Bc
    bound = 2.0 ** -10 * ref.abs() + 4 * K * 2.0 ** -24 * (Ac.abs() @ Bc.abs()) + 1e-30
    assert bool(((clean - ref).abs() <= bound).all())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(9134, 44, 11), (75, 2, 1), (600, 2048, 13), (6177, 2048, 178), (29, 11, 22), (131, 77, 9),
                                   (4096, 4096, 64), (3000, 3100, 264), (700, 900, 257), (8192, 2048, 128)])
def test_addmm_bias_row_and_odd_rows(gnnops, M, N, K, dtype):
    """What a Linear layer needs (gnnops/conv.py): the bias is ONE row added to every output row (gnnops_addmm_ld, pitch 0 —
    never expanded to [M, N]) and operand rows of any length go to the register-staged kernel as they are (K = 11 node
    features, N = 44: element / 4-B / 8-B pieces instead of padded copies). Every kernel of the family takes the pitch:
    shapes cover the register-staged, the 128 x 128 and the 256 x 256 LDS-DMA kernels and both fp32 kernels."""
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dtype)
    B = (torch.rand(K, N, generator=g) * 2 - 1 + torch.arange(N).float().view(1, N) / N).to(dtype)
    bias = (torch.rand(N, generator=g) * 2 - 1).to(dtype)
    eps = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8, torch.float32: 2.0 ** -24}[dtype]
    ref = bias.double().view(1, N) + A.double() @ B.double()
    bound = eps * ref.abs() + 4 * K * 2.0 ** -24 * (A.double().abs() @ B.double().abs()) + eps * bias.double().abs().view(1, N) + 1e-30
    for b in (bias, bias.view(1, N)):
        got = gnnops.addmm(b.cuda(), A.cuda(), B.cuda()).cpu().double()
        assert got.shape == (M, N)
        assert bool(((got - ref).abs() <= bound).all()), f"max err/bound {((got - ref).abs() / bound).max().item()}"


def test_addmm_more_rows_than_one_grid(gnnops):
    """M = 8.4M rows (a node-feature matrix of BASELINE config 2's order): more 128-row tiles than gridDim.y holds — run as
    slabs. Checked on sampled rows against float64."""
    M, K, N = 65280 * 128 + 777, 16, 24
    g = torch.Generator(device="cuda").manual_seed(4)
    A = (torch.rand(M, K, generator=g, device="cuda") - 0.5).half()
    B = (torch.rand(K, N, generator=g, device="cuda") - 0.5).half()
    bias = torch.rand(N, generator=g, device="cuda").half()
    got = gnnops.addmm(bias, A, B)
    rows = torch.tensor([0, 1, 127, 128, 65280 * 128 - 1, 65280 * 128, 65280 * 128 + 1, M - 1, 4_000_003], device="cuda")
    ref = bias.double().view(1, N) + A[rows].double() @ B.double()
    assert torch.allclose(got[rows].double(), ref, rtol=2e-3, atol=2e-3)
    full = gnnops.addmm(bias.expand(M, N).contiguous(), A, B)
    assert torch.equal(full, got)
