"""GPU parity of the one-shot row scatter (bucket.hip: the last radix pass finished on chip inside the reduction) —
what scatter_* / index_add_ run when the plan cache is off. Bit-exact against the sequential oracle AND against the
plan path (plan build + segment reduce), including buckets larger than the on-chip capacity (chunked in source order),
empty buckets, a ragged last bucket, rows wider than one lane group and `out=` / index_add_ accumulation."""
import numpy as np
import pytest
import torch

from helpers import TORCH_DT, to_np

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


@pytest.fixture()
def no_cache(gnnops):
    gnnops.set_plan_cache(False)
    yield
    gnnops.set_plan_cache(True)


def _index(kind, E, N, g):
    if kind == "uniform":
        return torch.randint(0, N, (E,), generator=g)
    if kind == "skew":  # most edges hit one destination, the rest are spread: buckets far beyond the on-chip capacity
        idx = torch.randint(0, N, (E,), generator=g)
        idx[torch.rand(E, generator=g) < 0.7] = min(5, N - 1)
        return idx
    if kind == "front":  # only the first few destinations are touched: every other bucket is empty
        return torch.randint(0, min(N, 100), (E,), generator=g)
    raise ValueError(kind)


CASES = [  # E, N, K
    (5000, 1000, 128), (100000, 300, 16), (20000, 5000, 4), (3000, 257, 260), (70000, 70000, 32), (1, 1000, 8),
]


@pytest.mark.parametrize("kind", ["uniform", "skew", "front"])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max", "mul"])
@pytest.mark.parametrize("E,N,K", CASES)
def test_oneshot_fp32(gnnops, oracle, no_cache, reduce, kind, E, N, K):
    g = torch.Generator().manual_seed(E + N + K)
    src = torch.rand(E, K, generator=g) * 2 - 1
    if reduce == "mul":
        src = 1 + src / 8
    idx = _index(kind, E, N, g)
    got = gnnops.scatter(src.cuda(), idx.cuda(), 0, dim_size=N, reduce=reduce)
    exp = oracle.scatter(src.numpy(), idx.numpy(), dim=0, dim_size=N, reduce=reduce)
    gnnops.set_plan_cache(True)
    via_plan = gnnops.scatter(src.cuda(), idx.cuda(), 0, dim_size=N, reduce=reduce)
    if reduce in ("min", "max"):
        assert np.array_equal(got[0].cpu().numpy(), exp[0]) and np.array_equal(got[1].cpu().numpy(), exp[1])
        assert torch.equal(got[0], via_plan[0]) and torch.equal(got[1], via_plan[1])
    elif np.bincount(idx.numpy(), minlength=N).max() > 8192:
        # a hub (more than 8192 contributions to one destination) is reduced piecewise (hub.h): deterministic, but its
        # fp32 sum / mean / product is re-associated — every other destination stays bit-exact
        hubs = np.bincount(idx.numpy(), minlength=N) > 8192
        assert np.array_equal(got.cpu().numpy()[~hubs], exp[~hubs])
        assert torch.equal(got[torch.from_numpy(~hubs)], via_plan[torch.from_numpy(~hubs)])
        np.testing.assert_allclose(got.cpu().numpy()[hubs], exp[hubs], rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(via_plan.cpu().numpy()[hubs], exp[hubs], rtol=2e-4, atol=2e-3)
    else:
        assert np.array_equal(got.cpu().numpy(), exp)
        assert torch.equal(got, via_plan)


@pytest.mark.parametrize("dname", ["f16", "bf16"])
@pytest.mark.parametrize("reduce", ["min", "max", "sum"])
def test_oneshot_16bit(gnnops, oracle, no_cache, reduce, dname):
    """min / max take the one-shot form in any dtype; 16-bit sums stay on the plan path — both must match the oracle."""
    E, N, K = 60000, 700, 64
    g = torch.Generator().manual_seed(9)
    src = (torch.rand(E, K, generator=g) * 2 - 1).to(TORCH_DT[dname])
    idx = _index("skew", E, N, g)
    got = gnnops.scatter(src.cuda(), idx.cuda(), 0, dim_size=N, reduce=reduce)
    exp = oracle.scatter(to_np(src), idx.numpy(), dim=0, dim_size=N, reduce=reduce, dtype=dname)
    if reduce == "sum":
        assert np.array_equal(to_np(got), exp)
    else:
        assert np.array_equal(to_np(got[0]), exp[0]) and np.array_equal(got[1].cpu().numpy(), exp[1])


@pytest.mark.parametrize("kind", ["uniform", "skew"])
def test_oneshot_accumulates_into_out(gnnops, no_cache, kind):
    E, N, K = 50000, 900, 32
    g = torch.Generator().manual_seed(4)
    src = torch.rand(E, K, generator=g)
    idx = _index(kind, E, N, g)
    base = torch.rand(N, K, generator=g)
    exp = base.numpy().copy()
    for e in range(E):  # sequential, like the oracle: base row first, then contributions in order
        exp[idx[e]] += src[e].numpy()
    out = base.clone().cuda()
    assert gnnops.index_add_(out, 0, idx.cuda(), src.cuda()) is out
    hubs = np.bincount(idx.numpy(), minlength=N) > 8192       # re-associated (hub.h); everything else bit-exact
    assert np.array_equal(out.cpu().numpy()[~hubs], exp[~hubs])
    np.testing.assert_allclose(out.cpu().numpy()[hubs], exp[hubs], rtol=2e-4, atol=2e-3)
    out2 = base.clone().cuda()
    gnnops.scatter(src.cuda(), idx.cuda(), 0, out=out2, reduce="sum")
    assert torch.equal(out, out2)
    mx = base.clone().cuda()
    got, arg = gnnops.scatter(src.cuda(), idx.cuda(), 0, out=mx, reduce="max")
    gnnops.set_plan_cache(True)
    ref, rarg = gnnops.scatter(src.cuda(), idx.cuda(), 0, out=base.clone().cuda(), reduce="max")
    assert torch.equal(got, ref) and torch.equal(arg, rarg)


def test_oneshot_config2_slice(gnnops, no_cache):
    """A few million edges at config 2's row shape: one-shot equals plan path bit for bit."""
    E, N, K = 4_000_000, 800_000, 128
    g = torch.Generator(device="cuda").manual_seed(1)
    src = torch.rand(E, K, generator=g, device="cuda")
    idx = torch.randint(0, N, (E,), generator=g, device="cuda")
    a = gnnops.scatter_add(src, idx, 0, dim_size=N)
    mn, am = gnnops.scatter_min(src, idx, 0, dim_size=N)
    gnnops.set_plan_cache(True)
    assert torch.equal(a, gnnops.scatter_add(src, idx, 0, dim_size=N))
    mn2, am2 = gnnops.scatter_min(src, idx, 0, dim_size=N)
    assert torch.equal(mn, mn2) and torch.equal(am, am2)


def _bucket_select(gnnops, table, idx):
    """gnnops_bucket_partition + gnnops_bucket_select through the C ABI (the size heuristic of ops.index_select aside)."""
    from gnnops import _lib
    from gnnops.ops import _stream

    L = _lib.load()
    N, K = table.shape
    E = idx.numel()
    out = torch.empty(E, K, dtype=table.dtype, device=table.device)
    ws_bytes = L.gnnops_bucket_workspace_bytes(E, N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=table.device)
    _lib.check(L.gnnops_bucket_partition(idx.data_ptr(), E, N, ws.data_ptr(), ws_bytes, _stream()), "bucket_partition")
    hb = L.gnnops_hub_workspace_bytes(E, 0, 0)
    hw = torch.empty(max(hb, 1), dtype=torch.uint8, device=table.device)
    _lib.check(L.gnnops_bucket_select_hubs(table.data_ptr(), ws.data_ptr(), out.data_ptr(), N, K, E, table.element_size(),
                                           hw.data_ptr() if hb else None, hb, _stream()), "bucket_select")
    return out


@pytest.mark.parametrize("kind", ["uniform", "skew", "front"])
@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.int64])
@pytest.mark.parametrize("E,N,K", [(5000, 1000, 128), (100000, 300, 16), (3000, 257, 264), (70000, 70000, 32), (1, 1000, 8)])
def test_bucket_select(gnnops, kind, dt, E, N, K):
    g = torch.Generator().manual_seed(E + N)
    table = (torch.rand(N, K, generator=g) * 1000).to(dt).cuda()
    idx = _index(kind, E, N, g).cuda()
    assert torch.equal(_bucket_select(gnnops, table, idx), table[idx])


def test_index_select_oneshot_push(gnnops, no_cache):
    """A table beyond the push threshold (>= 1 GiB, E >= 3 N): ops.index_select takes the bucketed push form when the
    plan cache is off, the planned push form when it is on; both equal torch's gather."""
    N, K, E = 2_200_000, 128, 7_000_000
    g = torch.Generator(device="cuda").manual_seed(2)
    table = torch.rand(N, K, generator=g, device="cuda")
    idx = torch.randint(0, N, (E,), generator=g, device="cuda")
    a = gnnops.index_select(table, 0, idx)
    assert torch.equal(a, table[idx])
    gnnops.set_plan_cache(True)
    assert torch.equal(gnnops.index_select(table, 0, idx), a)


@pytest.mark.parametrize("cache", [False, True])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max", "mul"])
def test_hubs_are_reduced_piecewise(gnnops, oracle, reduce, cache):
    """Destinations with more than 8192 contributions (hub.h), in both forms (one-shot: cache off; plan: cache on): several
    hubs, two of them in ONE bucket of 256 destinations, one spanning many on-chip chunks, plus ordinary destinations.
    min / max and their positions stay bit-exact; sums / means / products of the hubs are re-associated (tolerance), the
    ordinary destinations stay bit-exact."""
    E, N, K = 400_000, 3000, 32
    g = torch.Generator().manual_seed(123)
    src = torch.rand(E, K, generator=g) * 2 - 1
    if reduce == "mul":
        src = 1 + src / 4096
    idx = torch.randint(0, N, (E,), generator=g)
    r = torch.rand(E, generator=g)
    idx[r < 0.25] = 5          # ~100 000 contributions
    idx[(r >= 0.25) & (r < 0.30)] = 7      # ~20 000, same bucket as 5
    idx[(r >= 0.30) & (r < 0.33)] = 2900   # ~12 000, another bucket
    gnnops.set_plan_cache(cache)
    try:
        got = gnnops.scatter(src.cuda(), idx.cuda(), 0, dim_size=N, reduce=reduce)
        base = (torch.rand(N, K, generator=g) * 2 - 1) if reduce != "mean" else None
        got_out = None
        if base is not None:
            got_out = gnnops.scatter(src.cuda(), idx.cuda(), 0, out=base.clone().cuda(), reduce=reduce)
    finally:
        gnnops.set_plan_cache(True)
    exp = oracle.scatter(src.numpy(), idx.numpy(), dim=0, dim_size=N, reduce=reduce)
    hubs = np.bincount(idx.numpy(), minlength=N) > 8192
    assert hubs.sum() == 3
    if reduce in ("min", "max"):
        assert np.array_equal(got[0].cpu().numpy(), exp[0]) and np.array_equal(got[1].cpu().numpy(), exp[1])
        eo, ea = oracle.scatter(src.numpy(), idx.numpy(), dim=0, out=base.numpy().copy(), reduce=reduce)
        assert np.array_equal(got_out[0].cpu().numpy(), eo) and np.array_equal(got_out[1].cpu().numpy(), ea)
        return
    g_np = got.cpu().numpy()
    assert np.array_equal(g_np[~hubs], exp[~hubs])
    np.testing.assert_allclose(g_np[hubs], exp[hubs], rtol=3e-4, atol=3e-3)
    if got_out is not None:
        eo = oracle.scatter(src.numpy(), idx.numpy(), dim=0, out=base.numpy().copy(), reduce=reduce)
        go = got_out.cpu().numpy()
        assert np.array_equal(go[~hubs], eo[~hubs])
        np.testing.assert_allclose(go[hubs], eo[hubs], rtol=3e-4, atol=3e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_index_select_hot_rows(gnnops, dt):
    """Push-form index_select with hot table rows (selected by more than 8192 outputs — hub.h): two in one bucket of 256
    rows, one elsewhere, plus ordinary rows; bucketed (one-shot) and planned forms both equal torch's gather."""
    N, K, E = 3000, 64, 300_000
    g = torch.Generator().manual_seed(31)
    table = (torch.rand(N, K, generator=g) * 100).to(dt).cuda()
    idx = torch.randint(0, N, (E,), generator=g)
    r = torch.rand(E, generator=g)
    idx[r < 0.3] = 5
    idx[(r >= 0.3) & (r < 0.35)] = 9
    idx[(r >= 0.35) & (r < 0.4)] = 2000
    idx = idx.cuda()
    exp = table[idx]
    assert torch.equal(_bucket_select(gnnops, table, idx), exp)
    assert torch.equal(gnnops.index_select(table, 0, idx, plan=gnnops.Plan(idx, N)), exp)
