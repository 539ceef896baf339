"""GPU suite: seeded random sweep over shapes / dims / dtypes / reduces / index layouts for the scatter family, the
gathers and the plan, each case checked against the oracle with the bars of test_parity_gpu.py. Catches shape-dependent
slips (tile edges, strip widths, chunking, alignment classes) the hand-picked cases may miss."""
import numpy as np
import pytest
import torch

from helpers import TORCH_DT, assert_bits_equal, f32_of, to_np

pytestmark = pytest.mark.gpu
RTOL = {"f32": 1e-5, "f16": 1e-3, "bf16": 8e-3}


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


def _rand_shape(rng):
    nd = int(rng.integers(1, 4))
    pools = ([1, 2, 3, 5, 8, 16, 17, 31, 33, 64, 100, 127, 128, 129, 200, 256, 300, 511, 1000],)
    return tuple(int(rng.choice(pools[0])) for _ in range(nd))


@pytest.mark.parametrize("seed", range(12))
def test_random_scatter_and_gather(gnnops, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    g = torch.Generator().manual_seed(2000 + seed)
    for _ in range(14):
        shape = _rand_shape(rng)
        while np.prod(shape) > 400_000:
            shape = _rand_shape(rng)
        dim = int(rng.integers(0, len(shape)))
        dname = str(rng.choice(["f32", "f16", "bf16"]))
        reduce = str(rng.choice(["sum", "mean", "mul", "min", "max"]))
        layout = str(rng.choice(["R", "F"]))
        E = shape[dim]
        N = int(rng.integers(1, max(2, 2 * E)))
        src = (torch.rand(shape, generator=g) + 0.5).to(TORCH_DT[dname])
        idx = torch.randint(0, N, (E,) if layout == "R" else shape, generator=g)
        what = f"seed={seed} shape={shape} dim={dim} {dname} {reduce} {layout} N={N}"
        res = gnnops.scatter(src.cuda(), idx.cuda(), dim, dim_size=N, reduce=reduce)
        exp = oracle.scatter(to_np(src), idx.numpy(), dim, dim_size=N, reduce=reduce, dtype=dname)
        if reduce in ("min", "max"):
            assert_bits_equal(res[1].cpu().numpy(), exp[1], what + " arg")
            assert_bits_equal(to_np(res[0]), exp[0], what)
        elif layout == "R":
            assert_bits_equal(to_np(res), exp, what)
        else:
            a, b = f32_of(to_np(res), dname).astype(np.float64), f32_of(exp, dname).astype(np.float64)
            tol = RTOL[dname] * max(1.0, E if reduce == "mul" else np.sqrt(E))
            assert (np.abs(a - b) <= tol * np.maximum(np.abs(b), 1.0)).all(), what
        # gathers on the same shape
        sel_idx = torch.randint(0, E, (int(rng.integers(1, 2 * E + 1)),), generator=g)
        assert_bits_equal(to_np(gnnops.index_select(src.cuda(), dim, sel_idx.cuda())), oracle.index_select(to_np(src), dim, sel_idx.numpy()), what + " index_select")
        gshape = list(shape)
        gshape[dim] = int(rng.integers(1, E + 3))
        gidx = torch.randint(0, E, gshape, generator=g)
        assert_bits_equal(to_np(gnnops.gather(src.cuda(), dim, gidx.cuda())), oracle.gather(to_np(src), dim, gidx.numpy()), what + " gather")


@pytest.mark.parametrize("seed", range(4))
def test_random_plans_and_sorts(gnnops, oracle, seed):
    rng = np.random.default_rng(3000 + seed)
    g = torch.Generator().manual_seed(4000 + seed)
    for _ in range(10):
        E = int(rng.integers(0, 60_000))
        N = int(rng.choice([1, 2, 255, 256, 257, 65_535, 65_536, 65_537, 1 << 20, (1 << 24) + 5]))
        idx = torch.randint(0, N, (E,), generator=g)
        plan = gnnops.Plan(idx.cuda(), N)
        rowptr, perm = oracle.plan(idx.numpy(), N)
        assert_bits_equal(plan.rowptr.cpu().numpy(), rowptr, f"rowptr E={E} N={N}")
        assert_bits_equal(plan.perm.cpu().numpy()[:E], perm, f"perm E={E} N={N}")
        shape = _rand_shape(rng)
        x = torch.nn.functional.dropout(torch.rand(shape, generator=g) - 0.5, p=float(rng.choice([0.0, 0.5, 0.95])))
        dim = int(rng.integers(0, len(shape)))
        v, i = gnnops.sort(x.cuda(), dim=dim, stable=True)
        ev, ei = oracle.sort(x.numpy(), dim)
        assert_bits_equal(v.cpu().numpy(), ev, f"sort values {shape} d{dim}")
        assert_bits_equal(i.cpu().numpy(), ei, f"sort indices {shape} d{dim}")
