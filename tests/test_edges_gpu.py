"""GPU suite: edge cases — empty inputs, single elements, error behaviour at the drop-in boundary, the C ABI's
own argument checks (called directly through ctypes), non-contiguous inputs, the int32 limits."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import assert_bits_equal

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


def test_empty_inputs(gnnops):
    dev = "cuda"
    src = torch.empty(0, 16, device=dev)
    idx = torch.empty(0, dtype=torch.int64, device=dev)
    assert gnnops.scatter_add(src, idx, 0).shape == (0, 16)                       # dim_size from empty index -> 0
    out = gnnops.scatter_add(src, idx, 0, dim_size=5)
    assert out.shape == (5, 16) and torch.count_nonzero(out).item() == 0
    mn, arg = gnnops.scatter_min(src, idx, 0, dim_size=3)
    assert torch.count_nonzero(mn).item() == 0 and (arg == 0).all()               # arg = src.size(dim) = 0
    assert (gnnops.scatter_mul(src, idx, 0, dim_size=2) == 1).all()
    table = torch.rand(7, 16, device=dev)
    assert gnnops.index_select(table, 0, idx).shape == (0, 16)
    assert gnnops.index_select(torch.empty(0, 16, device=dev), 1, torch.tensor([3, 3], device=dev)).shape == (0, 2)
    assert gnnops.gather(table, 0, torch.empty(0, 16, dtype=torch.int64, device=dev)).shape == (0, 16)
    v, i = gnnops.sort(torch.empty(0, device=dev))
    assert v.numel() == 0 and i.dtype == torch.int64
    ci, cv = gnnops.coalesce(torch.empty(2, 0, dtype=torch.int64, device=dev), torch.empty(0, device=dev), 4, 4)
    assert ci.shape == (2, 0) and cv.shape == (0,)
    assert gnnops.spmm(torch.empty(2, 0, dtype=torch.int64, device=dev), torch.empty(0, device=dev), 3, 4,
                       torch.rand(4, 8, device=dev)).abs().sum().item() == 0
    assert gnnops.matmul(torch.empty(0, 8, device=dev, dtype=torch.bfloat16), torch.rand(8, 4, device=dev).bfloat16()).shape == (0, 4)
    z = gnnops.matmul(torch.empty(5, 0, device=dev, dtype=torch.bfloat16), torch.empty(0, 4, device=dev, dtype=torch.bfloat16))
    assert z.shape == (5, 4) and torch.count_nonzero(z).item() == 0               # K = 0: a sum over nothing
    assert gnnops.index_select_sum(table, 0, idx).item() == 0.0


def test_single_elements_and_scalars(gnnops, oracle):
    one = torch.tensor([[2.5]], device="cuda")
    assert gnnops.scatter_add(one, torch.tensor([0], device="cuda"), 0).item() == 2.5
    mx, arg = gnnops.scatter_max(one, torch.tensor([3], device="cuda"), 0)
    assert mx.flatten().tolist() == [0, 0, 0, 2.5] and arg.flatten().tolist() == [1, 1, 1, 0]
    v, i = gnnops.sort(torch.tensor([3.0], device="cuda"))
    assert v.item() == 3.0 and i.item() == 0


def test_python_boundary_errors(gnnops):
    src = torch.rand(8, 4, device="cuda")
    idx = torch.randint(0, 3, (8,), device="cuda")
    with pytest.raises(RuntimeError, match="int64"):
        gnnops.scatter_add(src, idx.int(), 0)
    with pytest.raises(IndexError):
        gnnops.scatter_add(src, idx, 2)
    with pytest.raises(ValueError):
        gnnops.scatter(src, idx, 0, reduce="median")
    with pytest.raises(NotImplementedError):
        gnnops.scatter_add(src.double(), idx, 0)
    with pytest.raises(IndexError, match="vector"):
        gnnops.index_select(src, 0, idx.view(2, 4))
    with pytest.raises(RuntimeError, match="same dtype"):
        gnnops.index_add_(src.clone(), 0, idx, src.half())
    with pytest.raises(IndexError, match="Number of indices"):
        gnnops.index_add_(src.clone(), 0, idx[:5], src)
    with pytest.raises(RuntimeError, match="cannot be multiplied"):
        gnnops.matmul(src.half(), src.half())
    with pytest.raises(NotImplementedError):
        gnnops.sort(src.to(torch.int64), dim=0)   # 64-bit keys: 1-D only
    plan = gnnops.Plan(idx, 3)
    with pytest.raises(ValueError, match="N=3"):
        gnnops.get_plan(plan, 4)
    with pytest.raises(ValueError, match="positions"):
        gnnops.scatter_add(src[:5], plan, 0)


def test_c_abi_argument_checks(gnnops):
    """The library itself refuses bad arguments with a code and a message; it never throws or exits."""
    L = gnnops.load_library()
    assert L.gnnops_index_max(None, 0, None, None) == 1 and b"d_max" in L.gnnops_last_error()
    assert L.gnnops_plan_build(None, -1, 3, None, None, None, 0, None) == 1
    buf = torch.zeros(64, dtype=torch.int32, device="cuda")
    idx = torch.zeros(4, dtype=torch.int64, device="cuda")
    assert L.gnnops_plan_build(idx.data_ptr(), 4, 2, buf.data_ptr(), buf.data_ptr(), None, 0, None) == 2  # workspace
    assert b"workspace" in L.gnnops_last_error()
    assert L.gnnops_plan_build(idx.data_ptr(), 1 << 31, 2, buf.data_ptr(), buf.data_ptr(), None, 0, None) == 4  # E >= 2^31
    assert L.gnnops_segment_reduce(None, buf.data_ptr(), None, buf.data_ptr(), None, 1, 0, 4, 2, 9, 0, 0, None) == 1  # dtype 9
    assert L.gnnops_segment_reduce(None, buf.data_ptr(), None, buf.data_ptr(), None, 1, 0, 4, 2, 0, 1, 1, None) == 1  # mean + init
    assert L.gnnops_index_select(buf.data_ptr(), idx.data_ptr(), buf.data_ptr(), 1, 4, 4, 4, 3, None) == 4      # 3-byte elems
    assert L.gnnops_addmm(None, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 2, 2, 2, 7, None, 0, None) == 4  # dtype 7
    torch.cuda.synchronize()


def test_non_contiguous_and_views(gnnops, oracle):
    g = torch.Generator().manual_seed(2)
    big = torch.rand(40, 64, generator=g)
    src = big[:, ::2]                        # strided view -> made contiguous by the host side
    idx = torch.randint(0, 9, (40,), generator=g)
    got = gnnops.scatter_add(src.cuda(), idx.cuda(), 0, dim_size=9)
    assert_bits_equal(got.cpu().numpy(), oracle.scatter(src.contiguous().numpy(), idx.numpy(), 0, dim_size=9), "strided src")
    t = big.t()                              # transposed view as index_select input
    sel = gnnops.index_select(t.cuda(), 0, idx.cuda() % 64)
    assert_bits_equal(sel.cpu().numpy(), oracle.index_select(t.contiguous().numpy(), 0, (idx % 64).numpy()), "transposed input")
    # offset (unaligned) base pointers take the element kernels
    base = torch.rand(1 + 33 * 6, generator=g).cuda()
    off = base[1:].view(33, 6)
    i2 = torch.randint(0, 33, (50,), generator=g)
    assert_bits_equal(gnnops.index_select(off, 0, i2.cuda()).cpu().numpy(), oracle.index_select(off.cpu().numpy(), 0, i2.numpy()), "offset base")


def test_skewed_and_maximal_degree(gnnops, oracle):
    """One destination takes every row (the 'long segment' path), and destinations far beyond the sources."""
    g = torch.Generator().manual_seed(4)
    src = torch.rand(70000, 16, generator=g)
    idx = torch.full((70000,), 3, dtype=torch.int64)
    got = gnnops.scatter_add(src.cuda(), idx.cuda(), 0, dim_size=5)
    exp = oracle.scatter(src.numpy(), idx.numpy(), 0, dim_size=5)
    # 70 000 contributions to one destination: a hub, reduced piecewise (csrc/hub.h) — deterministic, re-associated
    np.testing.assert_allclose(got.cpu().numpy(), exp, rtol=2e-5, atol=0)
    assert torch.equal(got, gnnops.scatter_add(src.cuda(), idx.cuda(), 0, dim_size=5)), "deterministic"
    few_idx = torch.full((8000,), 3, dtype=torch.int64)               # below the hub threshold: sequential, bit-exact
    assert_bits_equal(gnnops.scatter_add(src[:8000].cuda(), few_idx.cuda(), 0, dim_size=5).cpu().numpy(),
                      oracle.scatter(src[:8000].numpy(), few_idx.numpy(), 0, dim_size=5), "8000 rows -> one destination")
    mx, arg = gnnops.scatter_max(src.cuda(), idx.cuda(), 0, dim_size=5)
    emx, earg = oracle.scatter(src.numpy(), idx.numpy(), 0, dim_size=5, reduce="max")
    assert_bits_equal(mx.cpu().numpy(), emx, "max") and assert_bits_equal(arg.cpu().numpy(), earg, "argmax")
    few = torch.rand(10, 16, generator=g)
    far = torch.tensor([0, 2_999_999, 17, 17, 2_000_000, 5, 5, 5, 1_000_000, 0])
    got = gnnops.scatter_add(few.cuda(), far.cuda(), 0)              # dim_size discovered: 3,000,000 rows, 6 non-empty
    assert got.shape == (3_000_000, 16)
    assert_bits_equal(got.cpu().numpy(), oracle.scatter(few.numpy(), far.numpy(), 0), "sparse destinations")


def test_byte_movers_take_any_real_dtype(gnnops):
    """index_select / gather / transpose are opaque byte movers: int64, int32, uint8, bool, float64 all work
    (so torch.index_select on an edge_index keeps working after gnnops.install())."""
    g = torch.Generator().manual_seed(6)
    for dt in (torch.int64, torch.int32, torch.uint8, torch.bool, torch.float64, torch.int16):
        x = (torch.rand(300, 37, generator=g) * 200).to(dt)
        idx = torch.randint(0, 300, (500,), generator=g)
        assert torch.equal(gnnops.index_select(x.cuda(), 0, idx.cuda()).cpu(), x[idx]), dt
        cidx = torch.randint(0, 37, (20,), generator=g)
        assert torch.equal(gnnops.index_select(x.cuda(), 1, cidx.cuda()).cpu(), x[:, cidx]), dt
        gi = torch.randint(0, 300, (150, 37), generator=g)
        assert torch.equal(gnnops.gather(x.cuda(), 0, gi.cuda()).cpu(), torch.gather(x, 0, gi)), dt
        assert torch.equal(gnnops.transpose_contiguous(x.cuda()).cpu(), x.t().contiguous()), dt
    gnnops.install()
    try:
        ei = torch.randint(0, 50, (2, 400), generator=g).cuda()
        perm = torch.randperm(400, generator=g).cuda()
        assert torch.equal(torch.index_select(ei, 1, perm), ei[:, perm])
    finally:
        gnnops.uninstall()


def test_less_travelled_shapes(gnnops, oracle):
    """Paths the main suites touch lightly: push-form index_select with batch and > 1 KiB rows, coalesce with vector
    values, spmm with a 1-D operand, composite ops along a middle / last dim."""
    g = torch.Generator().manual_seed(41)
    x = torch.rand(3, 200, 320, generator=g)                      # B = 3, rows of 1280 B: two column chunks
    idx = torch.randint(0, 200, (700,), generator=g)
    plan = gnnops.Plan(idx.cuda(), 200)
    got = gnnops.index_select(x.cuda(), 1, idx.cuda(), plan=plan)
    assert_bits_equal(got.cpu().numpy(), oracle.index_select(x.numpy(), 1, idx.numpy()), "push form, batch, wide rows")
    coo = torch.stack([torch.randint(0, 30, (400,), generator=g), torch.randint(0, 20, (400,), generator=g)])
    val = torch.rand(400, 3, generator=g)
    ci, cv = gnnops.coalesce(coo.cuda(), val.cuda(), 30, 20)
    ei, ev = oracle.coalesce(coo.numpy(), val.numpy(), 30, 20)
    assert_bits_equal(ci.cpu().numpy(), ei, "coalesce index") and assert_bits_equal(cv.cpu().numpy(), ev, "coalesce [nnz, 3] values")
    vec = torch.rand(20, generator=g)
    v1 = torch.rand(400, generator=g)
    got = gnnops.spmm(coo.cuda(), v1.cuda(), 30, 20, vec.cuda())
    assert got.shape == (30,)
    assert_bits_equal(got.cpu().numpy(), oracle.spmm(coo.numpy(), v1.numpy(), 30, 20, vec.numpy().reshape(20, 1)).reshape(30), "spmm vector")
    import torch_scatter

    src = torch.randn(4, 300, 6, generator=g)
    idx2 = torch.randint(0, 17, (300,), generator=g)
    sm = torch_scatter.scatter_softmax(src.cuda(), idx2.cuda(), dim=1, dim_size=17).cpu()
    lse = torch_scatter.scatter_logsumexp(src.cuda(), idx2.cuda(), dim=1, dim_size=17).cpu()
    for b in range(4):
        np.testing.assert_allclose(sm[b].numpy(), oracle.composite(src[b].numpy(), idx2.numpy(), 17, "softmax"), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(lse[b].numpy(), oracle.composite(src[b].numpy(), idx2.numpy(), 17, "logsumexp"), rtol=2e-6, atol=2e-6)
    last = torch.randn(50, 400, generator=g)                      # dim = -1: K = 1
    sd = torch_scatter.scatter_std(last.cuda(), torch.randint(0, 9, (400,), generator=g).cuda(), dim=-1, dim_size=9)
    assert sd.shape == (50, 9) and torch.isfinite(sd).all()


def test_int32_indices_where_aten_takes_them(gnnops):
    """torch.index_select / Tensor.index_add_ accept int32 indices; so do ours (widened once)."""
    g = torch.Generator().manual_seed(1)
    table = torch.rand(50, 16, generator=g).cuda()
    idx = torch.randint(0, 50, (80,), generator=g, dtype=torch.int32).cuda()
    assert torch.equal(gnnops.index_select(table, 0, idx), table[idx.long()])
    src = torch.rand(80, 16, generator=g).cuda()
    out = table.clone()
    gnnops.index_add_(out, 0, idx, src)
    ref = table.clone()
    ref.index_add_(0, idx.long(), src)
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
