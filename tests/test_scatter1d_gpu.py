"""GPU parity of the long 1-D min / max form (csrc/scatter1d.hip: the reference's ">= 95 % of memory" 1-D shapes,
/root/reference data/scatter_min.csv:2, benchmark_scatter_min.py:15-18): values carried through a partial radix sort, buckets
of 32768 destinations finished in LDS. Against the sequential oracle, BIT-exact values and positions — NaNs, signed zeros,
infinities, ties, empty groups, a ragged last bucket, a single heavy destination — and against the plan path at 5M elements."""
import numpy as np
import pytest
import torch

from helpers import TORCH_DT, assert_bits_equal, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture()
def routed(monkeypatch):
    import gnnops
    from gnnops import _lib, ops

    lib = gnnops.load_library()
    calls = []
    real = lib.gnnops_scatter1d_minmax

    real_sum = lib.gnnops_scatter1d_sum

    class Spy:
        def __getattr__(self, name):
            if name == "gnnops_scatter1d_minmax":
                return lambda *a: (calls.append(1), real(*a))[1]
            if name == "gnnops_scatter1d_sum":
                return lambda *a: (calls.append(2), real_sum(*a))[1]
            return getattr(lib, name)

    monkeypatch.setattr(_lib, "load", lambda: Spy())
    monkeypatch.setattr(ops, "_SCATTER1D_MIN_N", 32769)
    return gnnops, calls


def _special(src, g):
    """sprinkle the values whose handling is a convention: NaN, +-inf, +-0, and heavy ties"""
    n = src.numel()
    pick = lambda k: torch.randint(0, n, (k,), generator=g)   # noqa: E731
    src[pick(n // 50)] = float("nan")
    src[pick(n // 50)] = float("inf")
    src[pick(n // 50)] = float("-inf")
    src[pick(n // 40)] = 0.0
    src[pick(n // 40)] = -0.0
    return src


@pytest.mark.parametrize("dname", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("N,E", [(100_000, 300_000), (32768 * 3 + 5, 1000), (40_000, 500_000)])
def test_long_1d_minmax_matches_the_oracle(routed, dname, reduce, N, E):
    gnnops, calls = routed
    from oracle import oracle

    g = torch.Generator().manual_seed(N + E)
    src = (torch.randint(-6, 7, (E,), generator=g).float() * 0.25).to(TORCH_DT[dname])      # few distinct values: ties everywhere
    src = _special(src, g)
    idx = torch.randint(0, N, (E,), generator=g)
    idx[idx == 7] = 8                                   # destination 7 receives nothing
    idx[: E // 20] = 12345                              # one heavy destination
    fn = gnnops.scatter_min if reduce == "min" else gnnops.scatter_max
    out, arg = fn(src.cuda(), idx.cuda(), 0, dim_size=N)
    assert calls, "the carried-value form did not run"
    ev, ea = oracle.scatter(to_np(src), idx.numpy(), dim=0, dim_size=N, reduce=reduce, dtype=dname)
    assert np.array_equal(arg.cpu().numpy(), ea), f"{(arg.cpu().numpy() != ea).sum()} positions differ"
    got, exp = to_np(out), ev
    # the sign of a zero result is the sign of the element `arg` points at (what torch's strict compare keeps): compare bits
    assert np.array_equal(got.view(f"u{got.dtype.itemsize}"), exp.view(f"u{exp.dtype.itemsize}")), "values differ (bits)"
    assert (arg[7] == E) and float(out[7]) == 0.0


@pytest.mark.parametrize("dname", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("reduce", ["sum", "mean", "mul"])
@pytest.mark.parametrize("N,E", [(100_000, 300_000), (32768 * 3 + 5, 1000), (40_000, 500_000)])
def test_long_1d_sums_keep_the_sequential_order(routed, dname, reduce, N, E):
    """sum / mean / product of a long 1-D tensor: three passes to buckets of 256 destinations, a stable on-chip sort, then every
    destination's values added one after the other in source position order — BIT-identical to the oracle's sequential loop
    (fp32 accumulator, 16-bit types rounded once), a heavy destination (25 000 terms) and empty groups included."""
    gnnops, calls = routed
    from oracle import oracle

    g = torch.Generator().manual_seed(N + E + 1)
    if reduce == "mul":
        src = (1.0 + (torch.rand(E, generator=g) - 0.5) * 0.01).to(TORCH_DT[dname])
    else:
        src = (torch.rand(E, generator=g) * 2 - 1).to(TORCH_DT[dname])
    idx = torch.randint(0, N, (E,), generator=g)
    idx[idx == 7] = 8
    idx[: E // 20] = 12345
    out = gnnops.scatter(src.cuda(), idx.cuda(), 0, dim_size=N, reduce=reduce)
    assert 2 in calls, "the carried-value sum form did not run"
    exp = oracle.scatter(to_np(src), idx.numpy(), dim=0, dim_size=N, reduce=reduce, dtype=dname)
    assert_bits_equal(to_np(out), exp, f"{reduce} {dname}")
    assert float(out[7]) == (1.0 if reduce == "mul" else 0.0)


def test_long_1d_minmax_equals_the_plan_path_at_5M(monkeypatch):
    import gnnops
    from gnnops import ops

    gnnops.load_library()
    N = E = 5_000_000                                    # above the default threshold: routed by itself
    g = torch.Generator(device="cuda").manual_seed(3)
    src = torch.rand(E, generator=g, device="cuda")
    idx = torch.randint(0, N, (E,), generator=g, device="cuda")
    gnnops.set_plan_cache(False)
    try:
        out, arg = gnnops.scatter_min(src, idx, 0, dim_size=N)
        mx, amx = gnnops.scatter_max(src, idx, 0, dim_size=N)
        monkeypatch.setattr(ops, "_SCATTER1D_MIN_N", 1 << 62)      # the plan path (full sort + gather)
        ref, rarg = gnnops.scatter_min(src, idx, 0, dim_size=N)
        rmx, ramx = gnnops.scatter_max(src, idx, 0, dim_size=N)
        plan_sum, plan_mean = gnnops.scatter_add(src, idx, 0, dim_size=N), gnnops.scatter_mean(src, idx, 0, dim_size=N)
        monkeypatch.setattr(ops, "_SCATTER1D_MIN_N", 1 << 22)
        assert torch.equal(gnnops.scatter_add(src, idx, 0, dim_size=N), plan_sum)       # same order of adds: bit for bit
        assert torch.equal(gnnops.scatter_mean(src, idx, 0, dim_size=N), plan_mean)
    finally:
        gnnops.set_plan_cache(True)
    assert torch.equal(out, ref) and torch.equal(arg, rarg) and torch.equal(mx, rmx) and torch.equal(amx, ramx)
    nonempty = arg < E
    assert bool((idx[arg[nonempty]] == nonempty.nonzero().flatten()).all()) and bool((src[arg[nonempty]] == out[nonempty]).all())
