"""GPU parity for the SURVEY.md §8(f) rank-1 widening: segment_csr / segment_coo / gather_csr / gather_coo,
scatter_softmax / log_softmax / logsumexp / std, and the autograd wrappers.

Bars: segment reductions bit-exact (same kernel and order as the scatter rows); composite ops within 2e-6
relative (fp32 exp/log of the device vs numpy differ by a few ulp; sums are sequential on both sides) and
1e-5 against torch's own per-group softmax; gradients equal to torch-CPU autograd of the equivalent
scatter_reduce / index formulation (bit-exact where the backward is a pure copy, 1e-6 otherwise)."""
import numpy as np
import pytest
import torch

from helpers import TORCH_DT, assert_bits_equal, to_np

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


def _sorted_problem(seed, E=4000, N=300, K=48):
    g = torch.Generator().manual_seed(seed)
    src = torch.rand(E, K, generator=g) * 4 - 2
    idx = torch.randint(0, N, (E,), generator=g).sort().values
    idx[idx == 7] = 8  # segment 7 empty
    indptr = torch.zeros(N + 1, dtype=torch.int64)
    indptr[1:] = torch.bincount(idx, minlength=N).cumsum(0)
    return src, idx, indptr


@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_segment_csr_coo(gnnops, oracle, dname):
    import torch_scatter

    src, idx, indptr = _sorted_problem(3)
    src = src.to(TORCH_DT[dname])
    N = indptr.numel() - 1
    for r in ("sum", "mean", "min", "max"):
        exp = oracle.segment_csr(to_np(src), indptr.numpy(), reduce=r, dtype=dname)
        got_csr = torch_scatter.segment_csr(src.cuda(), indptr.cuda(), reduce=r)
        got_coo = torch_scatter.segment_coo(src.cuda(), idx.cuda(), dim_size=N, reduce=r)
        if r in ("min", "max"):
            assert_bits_equal(got_csr[1].cpu().numpy(), exp[1], f"csr arg{r}")
            assert_bits_equal(got_coo[1].cpu().numpy(), exp[1], f"coo arg{r}")
            got_csr, got_coo, exp = got_csr[0], got_coo[0], exp[0]
        assert_bits_equal(to_np(got_csr), exp, f"segment_csr {r}")
        assert_bits_equal(to_np(got_coo), exp, f"segment_coo {r}")
    pooled = torch.rand(N, 48)
    assert torch.equal(torch_scatter.gather_csr(pooled.cuda(), indptr.cuda()).cpu(), pooled[idx])
    assert torch.equal(torch_scatter.gather_coo(pooled.cuda(), idx.cuda()).cpu(), pooled[idx])
    # int32 indptr, dim_size discovered from the index
    assert torch.equal(torch_scatter.segment_csr(src.cuda(), indptr.int().cuda()).cpu(),
                       torch_scatter.segment_coo(src.cuda(), idx.cuda()).cpu())


@pytest.mark.parametrize("E,N,K", [(3000, 200, 64), (500, 40, 7), (64, 64, 1)])
def test_composite_ops(gnnops, oracle, E, N, K):
    import torch_scatter

    g = torch.Generator().manual_seed(5)
    src = torch.randn(E, K, generator=g) * 3
    idx = torch.randint(0, N, (E,), generator=g)
    idx[idx == 2] = 3
    for mode, fn in (("softmax", torch_scatter.scatter_softmax), ("log_softmax", torch_scatter.scatter_log_softmax)):
        got = fn(src.cuda(), idx.cuda(), dim=0, dim_size=N).cpu().numpy()
        exp = oracle.composite(src.numpy(), idx.numpy(), N, mode)
        np.testing.assert_allclose(got, exp, rtol=2e-6, atol=2e-6, err_msg=mode)
    sm = torch_scatter.scatter_softmax(src.cuda(), idx.cuda(), dim=0, dim_size=N).cpu()
    for n in (0, 5, N - 1):  # torch's own softmax per group
        rows = torch.nonzero(idx == n).flatten()
        if rows.numel():
            np.testing.assert_allclose(sm[rows].numpy(), torch.softmax(src[rows], 0).numpy(), rtol=1e-5, atol=1e-7)
    got = torch_scatter.scatter_logsumexp(src.cuda(), idx.cuda(), dim=0, dim_size=N).cpu().numpy()
    np.testing.assert_allclose(got, oracle.composite(src.numpy(), idx.numpy(), N, "logsumexp"), rtol=2e-6, atol=2e-6)
    for unbiased in (True, False):
        got = torch_scatter.scatter_std(src.cuda(), idx.cuda(), dim=0, dim_size=N, unbiased=unbiased).cpu().numpy()
        np.testing.assert_allclose(got, oracle.composite(src.numpy(), idx.numpy(), N, "std", unbiased=unbiased), rtol=2e-6, atol=1e-6)
    rows = torch.nonzero(idx == 5).flatten()
    if rows.numel() > 1:
        got = torch_scatter.scatter_std(src.cuda(), idx.cuda(), dim=0, dim_size=N).cpu()
        np.testing.assert_allclose(got[5].numpy(), src[rows].std(0).numpy(), rtol=1e-4)


def _ref_scatter(src, idx, N, reduce):
    """torch-CPU formulation with autograd (scatter_reduce / index_add), the independent check of SURVEY.md §8c."""
    full = idx.view(-1, 1).expand_as(src)
    if reduce == "sum":
        return torch.zeros(N, src.size(1), dtype=src.dtype).index_add(0, idx, src)
    name = {"mean": "mean", "min": "amin", "max": "amax"}[reduce]
    return torch.zeros(N, src.size(1), dtype=src.dtype).scatter_reduce(0, full, src, name, include_self=False)


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_scatter_backward(gnnops, reduce):
    import torch_scatter

    g = torch.Generator().manual_seed(11)
    E, N, K = 2000, 150, 32
    src = torch.rand(E, K, generator=g).double().float()  # distinct values: no min/max ties, so the subgradient is unique
    idx = torch.randint(0, N, (E,), generator=g)
    w = torch.rand(N, K, generator=g)
    ref_src = src.clone().requires_grad_(True)
    (_ref_scatter(ref_src, idx, N, reduce) * w).sum().backward()
    dsrc = src.clone().cuda().requires_grad_(True)
    out = torch_scatter.scatter(dsrc, idx.cuda(), 0, dim_size=N, reduce=reduce)
    out = out[0] if isinstance(out, tuple) else out
    (out * w.cuda()).sum().backward()
    if reduce in ("sum", "min", "max"):
        assert torch.equal(dsrc.grad.cpu(), ref_src.grad), reduce  # pure routing of w: exact
    else:
        np.testing.assert_allclose(dsrc.grad.cpu().numpy(), ref_src.grad.numpy(), rtol=1e-6, atol=1e-7)
    # full-shape index takes the gather backward
    full = torch.randint(0, N, (E, K), generator=g)
    ref2 = src.clone().requires_grad_(True)
    (torch.zeros(N, K).scatter_add(0, full, ref2) * w).sum().backward()
    d2 = src.clone().cuda().requires_grad_(True)
    (torch_scatter.scatter_add(d2, full.cuda(), 0, dim_size=N) * w.cuda()).sum().backward()
    assert torch.equal(d2.grad.cpu(), ref2.grad)


def test_index_select_and_gather_backward(gnnops):
    from gnnops import autograd as ga

    g = torch.Generator().manual_seed(13)
    table = torch.rand(120, 16, generator=g)
    idx = torch.randint(0, 120, (500,), generator=g)
    w = torch.rand(500, 16, generator=g)
    ref = table.clone().requires_grad_(True)
    (torch.index_select(ref, 0, idx) * w).sum().backward()
    dev = table.clone().cuda().requires_grad_(True)
    (ga.index_select(dev, 0, idx.cuda()) * w.cuda()).sum().backward()
    assert torch.equal(dev.grad.cpu(), ref.grad)  # sequential index_add_ order on both sides
    gidx = torch.randint(0, 120, (300, 16), generator=g)
    w2 = torch.rand(300, 16, generator=g)
    ref = table.clone().requires_grad_(True)
    (torch.gather(ref, 0, gidx) * w2).sum().backward()
    dev = table.clone().cuda().requires_grad_(True)
    (ga.gather(dev, 0, gidx.cuda()) * w2.cuda()).sum().backward()
    np.testing.assert_allclose(dev.grad.cpu().numpy(), ref.grad.numpy(), rtol=1e-6, atol=1e-6)


def test_layers_and_plan_persistence(gnnops, oracle, tmp_path):
    """Fused propagate (one spmm launch) == index_select + scatter_add; GIN / SAGE forward vs a torch-CPU formulation;
    a plan survives save / load."""
    g = torch.Generator().manual_seed(17)
    Nn, E, D, Do = 500, 4000, 64, 32
    x = torch.rand(Nn, D, generator=g)
    ei = torch.stack([torch.randint(0, Nn, (E,), generator=g), torch.randint(0, Nn, (E,), generator=g)])
    fused = gnnops.layers.propagate_sum(x.cuda(), ei.cuda())
    unfused = gnnops.scatter_add(gnnops.index_select(x.cuda(), 0, ei[0].cuda()), ei[1].cuda(), 0, dim_size=Nn)
    assert torch.equal(fused, unfused)  # same sequential order per destination on both paths
    ref_sum = torch.zeros(Nn, D).index_add_(0, ei[1], x[ei[0]])
    assert torch.equal(fused.cpu(), ref_sum)
    W = (torch.rand(D, Do, generator=g) - 0.5).half()
    xh = x.half()
    gin = gnnops.layers.gin_conv(xh.cuda(), ei.cuda(), W.cuda()).float().cpu()
    agg = torch.zeros(Nn, D).index_add_(0, ei[1], xh.float()[ei[0]]).half().float()   # our aggregation rounds once to fp16
    ref = ((agg + xh.float()).half().float() @ W.float())
    assert ((gin - ref).abs() <= 2.0 ** -9 * ref.abs() + 2e-2).all()
    sage = gnnops.layers.sage_conv(xh.cuda(), ei.cuda(), W.cuda(), W.cuda()).float().cpu()
    deg = torch.bincount(ei[1], minlength=Nn).clamp(min=1).unsqueeze(1)
    ref = xh.float() @ W.float() + (agg / deg) @ W.float()
    assert ((sage - ref).abs() <= 2.0 ** -8 * ref.abs() + 5e-2).all()
    plan = gnnops.Plan(ei[1].cuda(), Nn)
    plan.save(tmp_path / "plan.pt")
    again = gnnops.Plan.load(tmp_path / "plan.pt")
    assert torch.equal(again.rowptr, plan.rowptr) and torch.equal(again.perm[:E], plan.perm[:E])
    src = torch.rand(E, 16, generator=g)
    assert torch.equal(gnnops.scatter_add(src.cuda(), again, 0), gnnops.scatter_add(src.cuda(), plan, 0))


@pytest.mark.parametrize("mode", ["softmax", "log_softmax", "logsumexp"])
def test_composite_backward(gnnops, mode):
    """Gradients of the composite ops against torch-CPU autograd of the per-group torch formulation."""
    import torch_scatter

    g = torch.Generator().manual_seed(23)
    E, N, K = 400, 30, 8
    src = torch.randn(E, K, generator=g)
    idx = torch.randint(0, N, (E,), generator=g)
    idx[idx == 3] = 4
    w = torch.rand(E if mode != "logsumexp" else N, K, generator=g)
    ref_src = src.clone().requires_grad_(True)
    total = torch.zeros(())
    for n in range(N):
        rows = torch.nonzero(idx == n).flatten()
        if rows.numel() == 0:
            continue
        x = ref_src[rows]
        if mode == "softmax":
            total = total + (torch.softmax(x, 0) * w[rows]).sum()
        elif mode == "log_softmax":
            total = total + (torch.log_softmax(x, 0) * w[rows]).sum()
        else:
            total = total + (torch.logsumexp(x, 0) * w[n]).sum()
    total.backward()
    dsrc = src.clone().cuda().requires_grad_(True)
    fn = {"softmax": torch_scatter.scatter_softmax, "log_softmax": torch_scatter.scatter_log_softmax,
          "logsumexp": torch_scatter.scatter_logsumexp}[mode]
    out = fn(dsrc, idx.cuda(), dim=0, dim_size=N)
    (out * w.cuda()).sum().backward()
    np.testing.assert_allclose(dsrc.grad.cpu().numpy(), ref_src.grad.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("unbiased", [True, False])
def test_scatter_std_backward(gnnops, unbiased):
    """d scatter_std / d src against torch-CPU autograd of the same per-group formula
    (sqrt(sum (x - mean)^2 / (max(cnt - ddof, 1) + 1e-6)), composite.hip); groups with one element or none get 0."""
    import torch_scatter

    g = torch.Generator().manual_seed(29)
    E, N, K = 500, 40, 8
    src = torch.randn(E, K, generator=g)
    idx = torch.randint(0, N, (E,), generator=g)
    idx[idx == 3] = 4          # group 3 empty
    idx[idx == 9] = 10
    idx[0] = 9                 # group 9 has exactly one element (std 0)
    w = torch.rand(N, K, generator=g)
    ref_src = src.clone().requires_grad_(True)
    total = torch.zeros(())
    for n in range(N):
        rows = torch.nonzero(idx == n).flatten()
        if rows.numel() < 2:
            continue
        x = ref_src[rows]
        c = max(rows.numel() - (1 if unbiased else 0), 1) + 1e-6
        total = total + (((x - x.mean(0)) ** 2).sum(0) / c).sqrt().mul(w[n]).sum()
    total.backward()
    dsrc = src.clone().cuda().requires_grad_(True)
    out = torch_scatter.scatter_std(dsrc, idx.cuda(), dim=0, dim_size=N, unbiased=unbiased)
    (out * w.cuda()).sum().backward()
    np.testing.assert_allclose(dsrc.grad.cpu().numpy(), ref_src.grad.numpy(), rtol=5e-5, atol=5e-6)
    assert (dsrc.grad[0] == 0).all()


@pytest.mark.parametrize("mode", ["softmax", "log_softmax", "logsumexp", "std"])
def test_composite_hub_groups(gnnops, oracle, mode):
    """Groups with more than 8192 members (csrc/hub.h) are processed piecewise by whole workgroups: same formulas, sums
    re-associated (tolerance); every other group as before."""
    import torch_scatter

    g = torch.Generator().manual_seed(41)
    E, N, K = 120_000, 300, 16
    src = torch.randn(E, K, generator=g)
    idx = torch.randint(0, N, (E,), generator=g)
    r = torch.rand(E, generator=g)
    idx[r < 0.35] = 5            # ~42 000 members
    idx[(r >= 0.35) & (r < 0.45)] = 200   # ~12 000
    fn = {"softmax": torch_scatter.scatter_softmax, "log_softmax": torch_scatter.scatter_log_softmax,
          "logsumexp": torch_scatter.scatter_logsumexp, "std": torch_scatter.scatter_std}[mode]
    got = fn(src.cuda(), idx.cuda(), dim=0, dim_size=N).cpu().numpy()
    # float64 reference of the same formulas (the fp32 sequential oracle itself carries ~1e-5 over 40 000 terms)
    x = src.double().numpy()
    ii = idx.numpy()
    exp = np.zeros_like(got, dtype=np.float64)
    for n in np.unique(ii):
        rows = np.nonzero(ii == n)[0]
        v = x[rows]
        if mode == "std":
            c = max(len(rows) - 1, 1) + 1e-6
            exp[n] = np.sqrt(((v - v.mean(0)) ** 2).sum(0) / c)
            continue
        m = v.max(0)
        s = np.exp(v - m).sum(0)
        if mode == "softmax":
            exp[rows] = np.exp(v - m) / s
        elif mode == "log_softmax":
            exp[rows] = (v - m) - np.log(s + 1e-12)
        else:
            exp[n] = m + np.log(s + 1e-12)
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-6)


# ------------------------------------------------------------------------------------------------
# autograd of the segment / gather / sparse ops and the layers (no silent loss of gradients)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_segment_csr_coo_backward(gnnops, reduce):
    """d segment_csr / d src and d segment_coo / d src against torch-CPU autograd of the scatter formulation; the CSR
    pointer here starts past 0 and stops short of E, so some positions belong to no segment (gradient 0)."""
    import torch_scatter

    src, idx, indptr = _sorted_problem(13, E=3000, N=200, K=24)
    src = src.double().float()
    N = indptr.numel() - 1
    w = torch.rand(N, src.size(1), generator=torch.Generator().manual_seed(1))
    ref = src.clone().requires_grad_(True)
    (_ref_scatter(ref, idx, N, reduce) * w).sum().backward()
    for form in ("csr", "coo"):
        d = src.clone().cuda().requires_grad_(True)
        out = (torch_scatter.segment_csr(d, indptr.cuda(), reduce=reduce) if form == "csr"
               else torch_scatter.segment_coo(d, idx.cuda(), dim_size=N, reduce=reduce))
        out = out[0] if isinstance(out, tuple) else out
        assert out.grad_fn is not None
        (out * w.cuda()).sum().backward()
        if reduce == "mean":
            np.testing.assert_allclose(d.grad.cpu().numpy(), ref.grad.numpy(), rtol=1e-6, atol=1e-7)
        else:
            assert torch.equal(d.grad.cpu(), ref.grad), (form, reduce)
    # a pointer that covers only positions [40, 2900): the rest of src gets a zero gradient
    part = indptr.clamp(min=40, max=2900)
    d = src.clone().cuda().requires_grad_(True)
    out = torch_scatter.segment_csr(d, part.cuda(), reduce="sum")
    (out * w.cuda()).sum().backward()
    seg = torch.searchsorted(part[1:], torch.arange(3000), right=True)
    pos = torch.arange(3000)
    exp = torch.where(((pos >= int(part[0])) & (pos < int(part[-1]))).view(-1, 1), w[seg.clamp(max=N - 1)], torch.zeros(()))
    assert torch.equal(d.grad.cpu(), exp)


def test_gather_csr_coo_forward_backward(gnnops):
    import torch_scatter

    src, idx, indptr = _sorted_problem(5, E=2500, N=180, K=16)
    N = indptr.numel() - 1
    x = torch.rand(N, 16, generator=torch.Generator().manual_seed(2))
    wt = torch.rand(2500, 16, generator=torch.Generator().manual_seed(3))
    ref = x.clone().requires_grad_(True)
    (ref[idx] * wt).sum().backward()
    for form in ("csr", "coo"):
        d = x.clone().cuda().requires_grad_(True)
        out = torch_scatter.gather_csr(d, indptr.cuda()) if form == "csr" else torch_scatter.gather_coo(d, idx.cuda())
        assert torch.equal(out.detach().cpu(), x[idx]), form
        (out * wt.cuda()).sum().backward()
        np.testing.assert_allclose(d.grad.cpu().numpy(), ref.grad.numpy(), rtol=1e-6, atol=1e-6)
    assert torch.equal(torch_scatter.gather_csr(x.cuda(), indptr.cuda()).cpu(), x[idx])   # no-grad form


@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_sddmm_vs_oracle_formula(gnnops, dname):
    g = torch.Generator().manual_seed(4)
    for D in (64, 24, 7, 300):
        a = (torch.rand(90, D, generator=g) * 2 - 1).to(TORCH_DT[dname])
        b = (torch.rand(70, D, generator=g) * 2 - 1).to(TORCH_DT[dname])
        ra, rb = torch.randint(0, 90, (800,), generator=g), torch.randint(0, 70, (800,), generator=g)
        got = gnnops.sddmm(ra.cuda(), rb.cuda(), a.cuda(), b.cuda()).float().cpu()
        exp = (a[ra].double() * b[rb].double()).sum(1)
        tol = 2.0 ** -8 if dname == "bf16" else 1e-5
        assert float(((got.double() - exp).abs() / (a[ra].double().abs() * b[rb].double().abs()).sum(1)).max()) <= tol


def test_spmm_backward_matrix_and_value(gnnops):
    """torch_sparse.spmm under autograd: d matrix = A^T g (the transposed row-split launch), d value = per-nonzero dot
    (gnnops_sddmm) — against torch-CPU autograd of the dense formulation; spmm_t likewise."""
    import torch_sparse

    g = torch.Generator().manual_seed(6)
    m, n, D, nnz = 120, 90, 40, 1500
    idx = torch.stack([torch.randint(0, m, (nnz,), generator=g), torch.randint(0, n, (nnz,), generator=g)])
    val = torch.rand(nnz, generator=g) * 2 - 1
    B = torch.rand(n, D, generator=g) * 2 - 1
    w = torch.rand(m, D, generator=g)
    rv, rB = val.clone().requires_grad_(True), B.clone().requires_grad_(True)
    (torch.zeros(m, D).index_add(0, idx[0], rB[idx[1]] * rv.unsqueeze(1)) * w).sum().backward()
    dv, dB = val.clone().cuda().requires_grad_(True), B.clone().cuda().requires_grad_(True)
    out = torch_sparse.spmm(idx.cuda(), dv, m, n, dB)
    assert out.grad_fn is not None
    (out * w.cuda()).sum().backward()
    np.testing.assert_allclose(dB.grad.cpu().numpy(), rB.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dv.grad.cpu().numpy(), rv.grad.numpy(), rtol=1e-5, atol=1e-6)
    # transposed form (message passing over edge_index = (source, destination)), value = None
    x = torch.rand(m, D, generator=g)
    wt = torch.rand(n, D, generator=g)
    rx = x.clone().requires_grad_(True)
    (torch.zeros(n, D).index_add(0, idx[1], rx[idx[0]]) * wt).sum().backward()
    dx = x.clone().cuda().requires_grad_(True)
    (gnnops.spmm_t(idx.cuda(), None, m, n, dx) * wt.cuda()).sum().backward()
    np.testing.assert_allclose(dx.grad.cpu().numpy(), rx.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_gin_layer_trains(gnnops):
    """layers.gin_conv inside a training graph: gradients of x, weight and bias against the torch-CPU formulation
    (bf16 weights would hide errors: fp32 operands through the fp32 MFMA GEMM, 1e-4)."""
    from gnnops import layers

    g = torch.Generator().manual_seed(7)
    Nn, E, Din, Dout = 300, 2000, 64, 48
    x = torch.rand(Nn, Din, generator=g) - 0.5
    ei = torch.stack([torch.randint(0, Nn, (E,), generator=g), torch.randint(0, Nn, (E,), generator=g)])
    W = torch.rand(Din, Dout, generator=g) - 0.5
    b = torch.rand(Dout, generator=g)
    rx, rW, rb = (t.clone().requires_grad_(True) for t in (x, W, b))
    h = torch.zeros(Nn, Din).index_add(0, ei[1], rx[ei[0]]) + rx
    ((h @ rW + rb) ** 2).sum().backward()
    dx, dW, db = (t.clone().cuda().requires_grad_(True) for t in (x, W, b))
    out = layers.gin_conv(dx, ei.cuda(), dW, db)
    (out ** 2).sum().backward()
    for got, ref in ((dx, rx), (dW, rW), (db, rb)):
        scale = float(ref.grad.abs().max())
        assert float((got.grad.cpu() - ref.grad).abs().max()) <= 1e-4 * scale


def test_raw_ops_refuse_operands_that_require_grad(gnnops):
    """No silent loss of gradients: the raw (ctypes) entry points raise when handed a tensor that requires grad, and so do
    the forms without a backward (out=, a Plan as index, spmm_csr); under no_grad everything runs."""
    from gnnops import ops

    src = torch.rand(50, 8).cuda().requires_grad_(True)
    idx = torch.randint(0, 10, (50,)).cuda()
    with pytest.raises(NotImplementedError):
        ops.scatter(src, idx, 0, dim_size=10)
    with pytest.raises(NotImplementedError):
        ops.index_select(src, 0, idx)
    with pytest.raises(NotImplementedError):
        gnnops.scatter(src, idx, 0, out=torch.zeros(10, 8).cuda())
    with pytest.raises(NotImplementedError):
        gnnops.scatter(src, gnnops.Plan(idx, 10), 0)
    with pytest.raises(NotImplementedError):
        gnnops.scatter_mul(src, idx, 0, dim_size=10)
    rowptr = torch.tensor([0, 50], dtype=torch.int32).cuda()
    with pytest.raises(NotImplementedError):
        gnnops.spmm_csr(rowptr, idx, None, src)
    with torch.no_grad():
        assert ops.scatter(src, idx, 0, dim_size=10).shape == (10, 8)
    assert gnnops.scatter(src, idx, 0, dim_size=10).grad_fn is not None
    assert gnnops.segment_csr(src, torch.tensor([0, 20, 50]).cuda()).grad_fn is not None
