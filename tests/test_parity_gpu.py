"""GPU suite: parity of the HIP path (called through the C ABI, csrc/libgnnops.so) against the oracle.

Bars (SURVEY.md §8c / BASELINE.json north_star):
  - index_select / gather / plan (rowptr, perm) / arg_out: bit-exact.
  - layout R reductions (plan path): the kernel adds in the same sequential order as the oracle and
    rounds 16-bit outputs once, so fp32, fp16 and bf16 results are required to be BIT-EXACT too.
  - layout F sums/means/products use float atomics whose arrival order is not fixed:
    fp32 |err| <= 1e-5 * sqrt(max_degree) * max|partial|, fp16 rtol 1e-3, bf16 rtol 8e-3 (the tolerances
    SURVEY.md §8c states); F min/max and their arg are bit-exact.
"""
import numpy as np
import pytest
import torch

from helpers import TORCH_DT, assert_bits_equal, f32_of, from_np, load_golden, to_np

pytestmark = pytest.mark.gpu

REDUCES = ["sum", "mean", "mul", "min", "max"]
RTOL = {"f32": 1e-5, "f16": 1e-3, "bf16": 8e-3}


@pytest.fixture(scope="module")
def gnnops():
    import gnnops as g

    g.load_library()
    g.set_plan_cache(False)  # every call builds its plan: exercises plan_build everywhere
    yield g
    g.set_plan_cache(True)


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o

    return o


def _close(got, exp, dname, degree, what):
    g32, e32 = f32_of(got, dname).astype(np.float64), f32_of(exp, dname).astype(np.float64)
    tol = RTOL[dname] * max(1.0, np.sqrt(degree))
    err = np.abs(g32 - e32)
    bound = tol * np.maximum(np.abs(e32), 1.0)
    assert (err <= bound).all(), f"{what}: max err {err.max()} (bound {bound[err.argmax()]})"


# ------------------------------------------------------------------------------------------------
# plan (stable inverted index)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("E,N,kind", [
    (0, 5, "uniform"), (1, 1, "uniform"), (7, 3, "uniform"), (8192, 100, "uniform"), (8193, 257, "uniform"),
    (100_000, 70_000, "uniform"), (300_000, 256, "uniform"), (200_000, 1 << 17, "uniform"),
    (50_000, 1, "uniform"), (60_000, 1000, "allsame"), (40_000, 5_000_000, "uniform"), (123_457, 9, "sorted"),
    (70_000, 20_000_000, "high"),
])
def test_plan_matches_oracle(gnnops, oracle, E, N, kind):
    g = torch.Generator().manual_seed(42)
    if kind == "allsame":
        idx = torch.full((E,), N - 1, dtype=torch.int64)
    elif kind == "high":  # exercises the 4th radix pass and long empty-destination runs
        idx = torch.randint(N - 1000, N, (E,), generator=g)
    else:
        idx = torch.randint(0, N, (E,), generator=g)
        if kind == "sorted":
            idx = idx.sort().values
    plan = gnnops.Plan(idx.cuda(), N)
    rowptr, perm = oracle.plan(idx.numpy(), N)
    assert_bits_equal(plan.rowptr.cpu().numpy(), rowptr, "rowptr")
    assert_bits_equal(plan.perm.cpu().numpy()[:E], perm, "perm")


def test_index_max(gnnops):
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, 123456, (1_000_003,), generator=g)
    assert gnnops.index_max(idx.cuda()) == int(idx.max())
    assert gnnops.index_max(torch.empty(0, dtype=torch.int64, device="cuda")) == -1


# ------------------------------------------------------------------------------------------------
# golden fixtures (inputs and expected outputs committed under tests/golden)
# ------------------------------------------------------------------------------------------------
def test_scatter_golden(gnnops):
    g = load_golden("scatter_golden.npz")
    keys = sorted(k[:-4] for k in g.files if k.startswith("scatter_") and k.endswith("_src"))
    for key in keys:
        dname = key.rsplit("_", 1)[1]
        layout = key.split("_")[-2]
        dim = int(key.split("_d")[1][0])
        src = from_np(g[key + "_src"], dname).cuda()
        idx = torch.from_numpy(g[key + "_idx"]).cuda()
        N = int(g[key + "_N"])
        E = src.shape[dim]
        for r in REDUCES:
            res = gnnops.scatter(src, idx, dim, dim_size=N, reduce=r)
            exp = g[f"{key}_{r}"]
            if r in ("min", "max"):
                out, arg = res
                assert_bits_equal(arg.cpu().numpy(), g[f"{key}_arg{r}"], f"{key} arg{r}")
                assert_bits_equal(to_np(out), exp, f"{key} {r}")
            elif layout == "R":
                assert_bits_equal(to_np(res), exp, f"{key} {r}")
            else:
                _close(to_np(res), exp, dname, E, f"{key} {r}")
    out, arg = gnnops.scatter_min(torch.from_numpy(g["known_min_src"]).cuda(), torch.from_numpy(g["known_min_idx"]).cuda(), 1)
    assert_bits_equal(out.cpu().numpy(), g["known_min_out"], "known min")
    assert_bits_equal(arg.cpu().numpy(), g["known_min_arg"], "known arg")
    res = gnnops.scatter_add(torch.from_numpy(g["allsame_src"]).cuda(), torch.from_numpy(g["allsame_idx"]).cuda(), 0)
    assert_bits_equal(res.cpu().numpy(), g["allsame_sum"], "all-same (dim_size from index.max()+1)")
    src = torch.from_numpy(g["ref223_src"]).cuda()
    idx = torch.from_numpy(g["ref223_idx"].astype(np.int64)).cuda()
    for dim in (0, 1):
        res = gnnops.scatter_add(src, idx, dim=dim)
        _close(to_np(res), g[f"ref223_d{dim}_sum"], "f16", 223, f"ref223 d{dim}")


def test_native_golden(gnnops):
    g = load_golden("native_golden.npz")
    keys = sorted(k[:-3] for k in g.files if k.endswith("_in"))
    for key in keys:
        dname = key.rsplit("_", 1)[1]
        dim = int(key.split("_d")[1][0])
        inp = from_np(g[key + "_in"], dname).cuda()
        idx = torch.from_numpy(g[key + "_idx"]).cuda()
        assert_bits_equal(to_np(gnnops.index_select(inp, dim, idx)), g[key + "_index_select"], key + " index_select")
        gidx = torch.from_numpy(g[key + "_gidx"]).cuda()
        assert_bits_equal(to_np(gnnops.gather(inp, dim, gidx)), g[key + "_gather"], key + " gather")
        source = from_np(g[key + "_source"], dname).cuda()
        got = gnnops.index_add_(inp.clone(), dim, idx, source)
        assert_bits_equal(to_np(got), g[key + "_index_add"], key + " index_add_")


# ------------------------------------------------------------------------------------------------
# seeded inputs vs the oracle
# ------------------------------------------------------------------------------------------------
R_SHAPES = [
    # (shape, dim, N)   row form: K*elem multiple of 16 B; element form otherwise
    ((5000, 64), 0, 1000),      # BASELINE config 1 shape family (N=100k,E=500k,D=64) scaled
    ((3000, 128), 0, 700),      # config 2 row length
    ((2000, 256), 0, 300),      # config 3 row length
    ((1500, 320), 0, 200),      # > 1 KiB rows: several column chunks
    ((999, 12), 0, 50),         # 48-byte rows: 3 of 4 lanes active
    ((777, 7), 0, 40),          # unaligned rows -> element kernel
    ((64, 3000), 1, 500),       # dim 1: B=64, K=1 (reference index_add_/index_select dim 1 shape)
    ((6, 400, 16), 1, 37),      # 3-D, middle dim
    ((4096, 1), 0, 10),         # K = 1
]


@pytest.mark.parametrize("dname", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape,dim,N", R_SHAPES)
def test_scatter_row_index_bit_exact(gnnops, oracle, shape, dim, N, dname):
    g = torch.Generator().manual_seed(42)
    src = (torch.rand(shape, generator=g) * 4 - 2).to(TORCH_DT[dname])
    E = shape[dim]
    idx = torch.randint(0, N, (E,), generator=g)
    idx[idx == 1] = 0  # destination 1 stays empty
    dsrc, didx = src.cuda(), idx.cuda()
    for r in REDUCES:
        res = gnnops.scatter(dsrc, didx, dim, dim_size=N, reduce=r)
        exp = oracle.scatter(to_np(src), idx.numpy(), dim, dim_size=N, reduce=r, dtype=dname)
        if r in ("min", "max"):
            assert_bits_equal(res[1].cpu().numpy(), exp[1], f"arg{r}")
            res, exp = res[0], exp[0]
        assert_bits_equal(to_np(res), exp, f"{r} {shape} d{dim} {dname}")


@pytest.mark.parametrize("dname", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape,dim,N", [((223, 223), 0, 111), ((223, 223), 1, 27), ((40, 50, 6), 1, 9), ((1000,), 0, 13),
                                         ((3000, 300), 0, 1500),     # LDS form, strips of 12-24 columns
                                         ((50000, 6), 0, 45000),     # destinations cut into LDS chunks (re-scan form)
                                         ((700000, 2), 0, 690000),   # too many chunks: global atomics
                                         ((7, 60000), 1, 50000)])    # K == 1, global atomics
def test_scatter_full_index(gnnops, oracle, shape, dim, N, dname):
    g = torch.Generator().manual_seed(43)
    src = (torch.rand(shape, generator=g) + 0.5).to(TORCH_DT[dname])  # positive, O(1): products stay finite
    idx = torch.randint(0, N, shape, generator=g)
    E = shape[dim]
    for r in REDUCES:
        res = gnnops.scatter(src.cuda(), idx.cuda(), dim, dim_size=N + 1, reduce=r)  # last destination empty
        exp = oracle.scatter(to_np(src), idx.numpy(), dim, dim_size=N + 1, reduce=r, dtype=dname)
        if r in ("min", "max"):
            assert_bits_equal(res[1].cpu().numpy(), exp[1], f"arg{r}")
            assert_bits_equal(to_np(res[0]), exp[0], r)
        elif r == "mul":
            g32, e32 = f32_of(to_np(res), dname).astype(np.float64), f32_of(exp, dname).astype(np.float64)
            rel = np.abs(g32 - e32) / np.maximum(np.abs(e32), 1e-30)
            assert rel.max() <= RTOL[dname] * E, f"mul rel err {rel.max()}"
        else:
            _close(to_np(res), exp, dname, E, f"{r} {shape} d{dim} {dname}")


@pytest.mark.parametrize("layout", ["R", "F", "R3"])
def test_scatter_mean_accepts_out(gnnops, oracle, layout):
    """torch_scatter.scatter_mean(src, index, dim, out=out): the sums are accumulated INTO out and the total is divided by
    max(count, 1) — out = (out + sum) / max(count, 1) (torch_scatter/scatter.py scatter_mean; upstream accepts `out` for every
    reduce, benchmark_scatter_mean.py:15-18 passes none). fp32, vs the sequential oracle: layout R bit-exact, F within the
    layout-F tolerance."""
    g = torch.Generator().manual_seed(8)
    if layout == "R":
        src, idx, dim, base = torch.rand(700, 48, generator=g), torch.randint(0, 90, (700,), generator=g), 0, torch.rand(90, 48, generator=g)
    elif layout == "R3":
        src, idx, dim, base = torch.rand(4, 300, 16, generator=g), torch.randint(0, 40, (300,), generator=g), 1, torch.rand(4, 40, 16, generator=g)
    else:
        src, idx, dim, base = torch.rand(200, 64, generator=g), torch.randint(0, 64, (200, 64), generator=g), 1, torch.rand(200, 64, generator=g)
    idx[idx == 3] = 4                                            # an empty group keeps out / 1
    out = base.clone().cuda()
    ret = gnnops.scatter_mean(src.cuda(), idx.cuda(), dim, out=out)
    assert ret.data_ptr() == out.data_ptr()
    exp = oracle.scatter(src.numpy(), idx.numpy(), dim, out=base.numpy(), reduce="mean")
    if layout == "F":
        _close(out.cpu().numpy(), exp, "f32", 16, "mean out= F")
    else:
        assert_bits_equal(out.cpu().numpy(), exp, "mean out= " + layout)
    import torch_scatter

    out2 = base.clone().cuda()
    assert torch_scatter.scatter_mean(src.cuda(), idx.cuda(), dim, out2) is out2      # the shim, positional `out`
    assert torch.equal(out2, out) or layout == "F"


def test_scatter_out_and_inplace_forms(gnnops, oracle):
    g = torch.Generator().manual_seed(7)
    src = torch.rand(500, 32, generator=g)
    idx = torch.randint(0, 60, (500,), generator=g)
    base = torch.rand(60, 32, generator=g)
    out = base.clone().cuda()
    ret = gnnops.scatter_add(src.cuda(), idx.cuda(), 0, out=out)
    assert ret.data_ptr() == out.data_ptr()
    assert_bits_equal(out.cpu().numpy(), oracle.scatter(src.numpy(), idx.numpy(), 0, out=base.numpy()), "out=")
    # expanded index (PyG idiom index.view(-1,1).expand_as(src)) takes the row path -> bit exact
    res = gnnops.scatter_add(src.cuda(), idx.cuda().view(-1, 1).expand(500, 32), 0, dim_size=60)
    assert_bits_equal(res.cpu().numpy(), oracle.scatter(src.numpy(), idx.numpy(), 0, dim_size=60), "expanded index")
    # native in-place forms (benchmark_scatter_add.py:22-25, benchmark_scatter_multiply.py:42-45)
    full = torch.randint(0, 500, (500, 32), generator=g)
    temp = torch.zeros_like(src).cuda()
    assert gnnops.scatter_add_(temp, 0, full.cuda(), src.cuda()) is temp
    exp = oracle.scatter(src.numpy(), full.numpy(), 0, out=np.zeros((500, 32), np.float32))
    _close(temp.cpu().numpy(), exp, "f32", 16, "scatter_add_")
    temp = torch.zeros_like(src).cuda()
    gnnops.scatter_reduce_mul_(temp, -1, torch.randint(0, 32, (500, 32), generator=g).cuda(), src.cuda())
    assert torch.count_nonzero(temp).item() == 0  # the reference's op: 0 * x stays 0
    # min with out=: keeps out where nothing beats it, no zero fill
    mo = torch.full((60, 32), 0.25).cuda()
    got, arg = gnnops.scatter_min(src.cuda(), idx.cuda(), 0, out=mo)
    eo, ea = oracle.scatter(src.numpy(), idx.numpy(), 0, out=np.full((60, 32), 0.25, np.float32), reduce="min")
    assert_bits_equal(got.cpu().numpy(), eo, "min out=")
    assert_bits_equal(arg.cpu().numpy(), ea, "min out= arg")


def test_plan_reuse_and_cache(gnnops, oracle):
    g = torch.Generator().manual_seed(9)
    src = torch.rand(20000, 64, generator=g)
    idx = torch.randint(0, 3000, (20000,), generator=g)
    didx = idx.cuda()
    plan = gnnops.Plan(didx, 3000)
    exp = oracle.scatter(src.numpy(), idx.numpy(), 0, dim_size=3000)
    assert_bits_equal(gnnops.scatter_add(src.cuda(), plan, 0).cpu().numpy(), exp, "explicit plan")
    gnnops.set_plan_cache(True)
    try:
        a = gnnops.get_plan(didx, 3000)
        assert gnnops.get_plan(didx, 3000) is a
        didx[0] = (didx[0] + 1) % 3000  # in-place edit bumps the version counter -> rebuild
        b = gnnops.get_plan(didx, 3000)
        assert b is not a
        idx2 = didx.cpu()
        assert_bits_equal(gnnops.scatter_add(src.cuda(), didx, 0, dim_size=3000).cpu().numpy(),
                          oracle.scatter(src.numpy(), idx2.numpy(), 0, dim_size=3000), "after edit")
    finally:
        gnnops.set_plan_cache(False)
    # push-form index_select over the same plan == pull form == oracle
    table = torch.rand(3000, 64, generator=g)
    idx3 = torch.randint(0, 3000, (20000,), generator=g)
    p3 = gnnops.Plan(idx3.cuda(), 3000)
    exp = oracle.index_select(table.numpy(), 0, idx3.numpy())
    assert_bits_equal(gnnops.index_select(table.cuda(), 0, idx3.cuda()).cpu().numpy(), exp, "pull")
    assert_bits_equal(gnnops.index_select(table.cuda(), 0, idx3.cuda(), plan=p3).cpu().numpy(), exp, "push")


@pytest.mark.parametrize("dname", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape,dim", [((3000, 128), 0), ((2738, 2738), 0), ((500, 1000), 1), ((777, 7), 0),
                                       ((5, 300, 24), 1), ((1500, 320), 0)])
def test_index_select_and_gather_bit_exact(gnnops, oracle, shape, dim, dname):
    g = torch.Generator().manual_seed(11)
    inp = torch.rand(shape, generator=g).to(TORCH_DT[dname])
    Nn = shape[dim]
    for E in (Nn, max(1, Nn // 8)):  # reduce factors 1 and 8 (benchmark_native_index_select.py:62)
        idx = torch.randint(0, Nn, (E,), generator=g)
        got = gnnops.index_select(inp.cuda(), dim, idx.cuda())
        assert_bits_equal(to_np(got), oracle.index_select(to_np(inp), dim, idx.numpy()), f"index_select E={E}")
    if inp.numel() <= 2_000_000:
        gidx = torch.randint(0, Nn, shape, generator=g)
        got = gnnops.gather(inp.cuda(), dim, gidx.cuda())
        assert_bits_equal(to_np(got), oracle.gather(to_np(inp), dim, gidx.numpy()), "gather")


@pytest.mark.parametrize("dname", ["f32", "f16"])
def test_index_add_reference_shape(gnnops, oracle, dname):
    """dim=1, index length = input.shape[1] (benchmark_native_index_add_.py:62,79-86)."""
    g = torch.Generator().manual_seed(13)
    L = 1581
    inp = torch.rand(L, L, generator=g).to(TORCH_DT[dname])
    source = torch.rand(L, L, generator=g).to(TORCH_DT[dname])
    idx = torch.randint(0, L, (L,), generator=g)
    got = gnnops.index_add_(inp.clone().cuda(), 1, idx.cuda(), source.cuda())
    exp = oracle.index_add_(to_np(inp), 1, idx.numpy(), to_np(source), dtype=dname)
    assert_bits_equal(to_np(got), exp, "index_add_ dim 1")


@pytest.mark.parametrize("dname", ["f32", "f16", "bf16"])
def test_fused_index_select_sum(gnnops, oracle, dname):
    g = torch.Generator().manual_seed(17)
    inp = torch.rand(2738, 2738, generator=g).to(TORCH_DT[dname])
    idx = torch.randint(0, 2738, (2738,), generator=g)
    for dim in (0, 1):
        got = gnnops.index_select_sum(inp.cuda(), dim, idx.cuda()).item()
        exp = oracle.index_select_sum(to_np(inp), dim, idx.numpy(), dtype=dname)
        assert abs(got - exp) <= 1e-5 * abs(exp), (got, exp)  # fp32 tree sum vs double
    small = torch.rand(7, 5, generator=g).to(TORCH_DT[dname])
    got = gnnops.index_select_sum(small.cuda(), 1, torch.tensor([4, 0, 0]).cuda()).item()
    assert abs(got - oracle.index_select_sum(to_np(small), 1, np.array([4, 0, 0]), dtype=dname)) < 1e-5


def test_aten_overrides_route_to_hip(gnnops, oracle):
    """The ATen seam: unchanged script text (torch.index_select, Tensor.index_add_, ...) reaches our kernels."""
    g = torch.Generator().manual_seed(19)
    inp = torch.rand(300, 64, generator=g)
    idx = torch.randint(0, 300, (450,), generator=g)
    source = torch.rand(450, 64, generator=g)
    gnnops.install()
    try:
        assert gnnops.installed()
        sel = torch.index_select(inp.cuda(), 0, idx.cuda())
        acc = inp.clone().cuda()
        acc.index_add_(0, idx.cuda(), source.cuda())
        gat = torch.gather(inp.cuda(), 0, idx[:300].view(-1, 1).expand(300, 64).contiguous().cuda())
        tmp = torch.zeros(300, 64, device="cuda")
        tmp.scatter_add_(0, idx.cuda().view(-1, 1).expand(450, 64), source.cuda())
    finally:
        gnnops.uninstall()
    assert not gnnops.installed()
    assert_bits_equal(sel.cpu().numpy(), oracle.index_select(inp.numpy(), 0, idx.numpy()), "aten index_select")
    assert_bits_equal(acc.cpu().numpy(), oracle.index_add_(inp.numpy(), 0, idx.numpy(), source.numpy()), "aten index_add_")
    assert_bits_equal(gat.cpu().numpy(), oracle.index_select(inp.numpy(), 0, idx[:300].numpy()), "aten gather")
    assert_bits_equal(tmp.cpu().numpy(), oracle.scatter(source.numpy(), idx.numpy(), 0, dim_size=300), "aten scatter_add_")
    # and the stock kernels are back afterwards
    assert torch.equal(torch.index_select(inp.cuda(), 0, idx.cuda()).cpu(), sel.cpu())


def test_config1_scatter_add(gnnops, oracle):
    """BASELINE config 1 at full size: N=100k, E=500k, D=64 fp32 (the reference's CPU-runnable case)."""
    g = torch.Generator().manual_seed(42)
    N, E, D = 100_000, 500_000, 64
    src = torch.rand(E, D, generator=g)
    idx = torch.randint(0, N, (E,), generator=g)
    got = gnnops.scatter_add(src.cuda(), idx.cuda(), dim=0)
    exp = oracle.scatter(src.numpy(), idx.numpy(), 0)
    assert_bits_equal(got.cpu().numpy(), exp, "config 1")


@pytest.mark.parametrize("dname", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max", "mul"])
@pytest.mark.parametrize("B,E,N", [(37, 1001, 300), (5, 4096, 4096), (300, 600, 256), (3, 70000, 20000)])  # last: one 1024-thread workgroup per CU
def test_batched_k1_rows_in_lds(gnnops, oracle, B, E, N, reduce, dname):
    """src [B, E] reduced along dim 1 with ONE row index for every b (layout R, K == 1): the LDS-staged kernel of
    segment.hip (whole rows parked on chip, no atomics) — bit-exact against the sequential oracle, ragged last tile of
    rows, row lengths that are not 16-B multiples, empty destinations, and accumulation into `out`."""
    g = torch.Generator().manual_seed(B + E)
    src = (torch.rand(B, E, generator=g) * 2 - 1).to(TORCH_DT[dname])
    if reduce == "mul":
        src = (1 + src.float() / 8).to(TORCH_DT[dname])
    idx = torch.randint(0, N, (E,), generator=g)
    idx[idx % 11 == 3] = 0
    got = gnnops.scatter(src.cuda(), idx.cuda(), 1, dim_size=N, reduce=reduce)
    exp = oracle.scatter(to_np(src), idx.numpy(), dim=1, dim_size=N, reduce=reduce, dtype=dname)
    if reduce in ("min", "max"):
        assert_bits_equal(to_np(got[0]), exp[0], "values")
        assert np.array_equal(got[1].cpu().numpy(), exp[1])
    else:
        assert_bits_equal(to_np(got), exp, reduce)
    if reduce == "sum":
        base = (torch.rand(B, N, generator=g)).to(TORCH_DT[dname])
        out = base.clone().cuda()
        gnnops.index_add_(out, 1, idx.cuda(), src.cuda())
        assert_bits_equal(to_np(out), oracle.index_add_(to_np(base), 1, idx.numpy(), to_np(src), dtype=dname), "index_add_")


@pytest.mark.parametrize("dt,K", [(torch.float16, 1001), (torch.float16, 1002), (torch.float16, 1004), (torch.float32, 513),
                                  (torch.float32, 514), (torch.int64, 600), (torch.uint8, 777)])
def test_index_select_long_misaligned_rows(gnnops, dt, K):
    """Rows whose byte length is not a multiple of 16 (the reference's (L, L) fp16 shapes): the wave-per-row kernel, in
    every copy unit (1, 2, 4, 8 bytes), with repeated and unselected rows; bit-exact against torch's own gather."""
    g = torch.Generator().manual_seed(K)
    N, E = 300, 700
    table = (torch.rand(N, K, generator=g) * 200).to(dt).cuda()
    idx = torch.randint(0, N, (E,), generator=g).cuda()
    assert torch.equal(gnnops.index_select(table, 0, idx), table[idx])
    t3 = (torch.rand(3, N, K, generator=g) * 200).to(dt).cuda()      # a batch in front
    assert torch.equal(gnnops.index_select(t3, 1, idx), t3[:, idx])
    if dt.is_floating_point:
        got = gnnops.index_select_sum(table, 0, idx).item()
        exp = table[idx].double().sum().item()
        assert abs(got - exp) <= 1e-5 * abs(exp) + 1e-3, (got, exp)


@pytest.mark.parametrize("dname", ["f32", "f16"])
@pytest.mark.parametrize("reduce", ["min", "max"])
def test_layout_f_minmax_ties_zeros_nans(gnnops, oracle, reduce, dname):
    """Full-shape index (layout F) min / max through the one-pass packed LDS form: massive ties, +0.0 / -0.0 (equal: the
    earlier position wins and ITS bits are returned), NaNs (never win), empty groups (0, arg = E), accumulation into out."""
    g = torch.Generator().manual_seed(77)
    L, N = 300, 40
    src = torch.randint(-3, 4, (L, L), generator=g).float()      # few distinct values: ties everywhere
    src[src == 0] = torch.where(torch.rand(int((src == 0).sum()), generator=g) < 0.5, 0.0, -0.0)
    src[torch.rand(L, L, generator=g) < 0.02] = float("nan")
    src = src.to(TORCH_DT[dname])
    idx = torch.randint(0, N, (L, L), generator=g)
    idx[idx == 7] = 8                                             # group 7 is empty
    for dim in (0, 1):
        got, arg = gnnops.scatter(src.cuda(), idx.cuda(), dim, dim_size=N, reduce=reduce)
        exp, earg = oracle.scatter(to_np(src), idx.numpy(), dim=dim, dim_size=N, reduce=reduce, dtype=dname)
        assert_bits_equal(to_np(got), exp, f"{reduce} dim {dim}")
        assert np.array_equal(arg.cpu().numpy(), earg)
    shape = (N, L)
    base = torch.randint(-2, 3, shape, generator=g).float().to(TORCH_DT[dname])
    got, arg = gnnops.scatter(src.cuda(), idx.cuda(), 0, out=base.clone().cuda(), reduce=reduce)
    exp, earg = oracle.scatter(to_np(src), idx.numpy(), dim=0, out=to_np(base).copy(), reduce=reduce, dtype=dname)
    assert_bits_equal(to_np(got), exp, "with out")
    assert np.array_equal(arg.cpu().numpy(), earg)


@pytest.mark.parametrize("dname", ["f32", "f16"])
@pytest.mark.parametrize("reduce", ["min", "max"])
def test_minmax_identity_only_groups_agree_across_layouts(gnnops, oracle, reduce, dname):
    """A group whose only contributions are the reduce's identity (+inf for min, -inf for max) never improves on it: the
    sequential loop (strict compare) leaves the group "empty" — value 0, arg = E — in EVERY form: the row kernels (plan and
    one-shot), the packed LDS element kernel and the global-atomic fallback. The reference pins nothing here
    (torch_scatter absent: parity unpinned); what is pinned is that all forms agree with the oracle's loop and each other."""
    ident = float("inf") if reduce == "min" else float("-inf")
    g = torch.Generator().manual_seed(5)
    E, K, N = 600, 16, 50
    src = (torch.rand(E, K, generator=g) * 4 - 2).to(TORCH_DT[dname])
    row = torch.randint(0, N, (E,), generator=g)
    row[row == 3] = 4                          # group 3: nothing at all
    src[row == 5] = ident                      # group 5: only identities
    src[(row == 6).nonzero()[:1]] = ident      # group 6: an identity among ordinary values
    src[row == 9] = -ident                     # group 9: only the opposite infinity (an ordinary winner)
    exp, earg = oracle.scatter(to_np(src), row.numpy(), dim=0, dim_size=N, reduce=reduce, dtype=dname)
    assert (f32_of(exp, dname)[5] == 0).all() and (earg[5] == E).all()
    full = row.view(-1, 1).expand(E, K).contiguous()              # layout F: same destinations, per-element index
    results = {
        "row index (plan / one-shot)": gnnops.scatter(src.cuda(), row.cuda(), 0, dim_size=N, reduce=reduce),
        "full index (LDS cells)": gnnops.scatter(src.cuda(), full.cuda(), 0, dim_size=N, reduce=reduce),
    }
    for what, (got, arg) in results.items():
        assert_bits_equal(to_np(got), exp, what)
        assert np.array_equal(arg.cpu().numpy(), earg), what
    # the global-atomic fallback (destinations that do not fit LDS strips): two rows along dim 1, N large
    E1, N1 = 20000, 3_000_000
    s1 = (torch.rand(2, E1, generator=g) * 4 - 2).to(TORCH_DT[dname])
    i1 = torch.randint(0, N1, (2, E1), generator=g)
    s1[:, :100] = ident
    i1[:, :100] = torch.arange(100) + 17                            # 100 groups per row fed the identity (maybe plus more)
    got, arg = gnnops.scatter(s1.cuda(), i1.cuda(), 1, dim_size=N1, reduce=reduce)
    exp, earg = oracle.scatter(to_np(s1), i1.numpy(), dim=1, dim_size=N1, reduce=reduce, dtype=dname)
    assert_bits_equal(to_np(got), exp, "atomic fallback")
    assert np.array_equal(arg.cpu().numpy(), earg)


def test_ops_inside_inference_mode(gnnops, oracle):
    """Tensors created under torch.inference_mode() have no version counter (`_version` raises): the plan / CSR caches
    must step aside instead of crashing (the normal GNN inference setting)."""
    gnnops.set_plan_cache(True)
    try:
        with torch.inference_mode():
            g = torch.Generator().manual_seed(9)
            src = torch.rand(500, 32, generator=g)
            idx = torch.randint(0, 60, (500,), generator=g)
            d_src, d_idx = src.cuda(), idx.cuda()
            for _ in range(2):
                got = gnnops.scatter_add(d_src, d_idx, dim=0, dim_size=60)
            exp = oracle.scatter(src.numpy(), idx.numpy(), dim=0, dim_size=60, reduce="sum")
            assert_bits_equal(to_np(got), exp, "scatter_add in inference_mode")
            coo = torch.stack([idx, torch.randint(0, 500, (500,), generator=g)])
            val = torch.rand(500, generator=g)
            d_coo, d_val = coo.cuda(), val.cuda()
            for _ in range(2):
                out = gnnops.spmm(d_coo, d_val, 60, 500, d_src)
            assert_bits_equal(to_np(out), oracle.spmm(coo.numpy(), val.numpy(), 60, 500, src.numpy()), "spmm in inference_mode")
            sel = gnnops.index_select(d_src, 0, d_idx)
            assert torch.equal(sel.cpu(), src[idx])
    finally:
        gnnops.set_plan_cache(False)


def test_csr_cache_hits_for_the_same_coo_tensor(gnnops):
    """The CSR arrays of a COO operand are kept on its plan, keyed on the [2, nnz] tensor object (not on the transient row
    view): a second spmm / spmm_t with the same tensors must not permute again."""
    from gnnops import sparse

    gnnops.set_plan_cache(True)
    try:
        g = torch.Generator().manual_seed(2)
        coo = torch.stack([torch.randint(0, 70, (900,), generator=g), torch.randint(0, 80, (900,), generator=g)]).cuda()
        val = torch.rand(900, generator=g).cuda()
        B = torch.rand(80, 16, generator=g).cuda()
        calls = []
        real = sparse._permute
        sparse._permute = lambda *a: (calls.append(1), real(*a))[1]
        try:
            a = gnnops.spmm(coo, val, 70, 80, B)
            n_first = len(calls)
            b = gnnops.spmm(coo, val, 70, 80, B)
            assert n_first == 2 and len(calls) == 2, calls     # columns + values, once
            assert torch.equal(a, b)
            val.mul_(2)                                         # an in-place write bumps the version: re-materialised
            c = gnnops.spmm(coo, val, 70, 80, B)
            assert len(calls) == 4
            assert torch.equal(c, a * 2)
        finally:
            sparse._permute = real
    finally:
        gnnops.set_plan_cache(False)


@pytest.mark.parametrize("dname,reduce", [("f32", "max"), ("f32", "min"), ("f32", "mean"), ("f32", "sum"), ("f16", "sum"), ("f32", "mul")])
def test_layout_f_dim0_more_destinations_than_an_lds_strip(gnnops, oracle, reduce, dname, monkeypatch):
    """Full-shape index along dim 0 with N too large for an LDS strip of destinations (the reference's (38000, 38000) shapes,
    data/scatter_max.csv:32-33): routed through tile transposes + the last-dim kernel. Values and arg bit-exact for min / max,
    sums / means / products within the layout-F tolerance; also a 3-D src reduced along dim 0."""
    g = torch.Generator().manual_seed(12)
    E, K = 300, 24
    N = 45_000 if reduce in ("sum", "mul") else 21_000          # > 159.5 KiB / (4 or 8 B per destination)
    src = (torch.rand(E, K, generator=g) * 2 - 1).to(TORCH_DT[dname])
    idx = torch.randint(0, N, (E, K), generator=g)
    idx[:40] = idx[40:80]                                        # ties for min / max, repeated destinations for sums
    from gnnops import ops

    took = []
    real = ops._scatter_transposed
    monkeypatch.setattr(ops, "_scatter_transposed", lambda *a: (lambda r: (took.append(r is not None), r)[1])(real(*a)))
    got = gnnops.scatter(src.cuda(), idx.cuda(), 0, dim_size=N, reduce=reduce)
    assert took == [True], "the narrowed route (int32 index / arg inside the transposes) did not take the call"
    exp = oracle.scatter(to_np(src), idx.numpy(), dim=0, dim_size=N, reduce=reduce, dtype=dname)
    if reduce in ("min", "max"):
        assert_bits_equal(to_np(got[0]), exp[0], reduce)
        assert np.array_equal(got[1].cpu().numpy(), exp[1])
    else:
        _close(to_np(got), exp, dname, 4, reduce)
    src3 = src.view(E, 4, 6)
    got3 = gnnops.scatter(src3.cuda(), idx.view(E, 4, 6).cuda(), 0, dim_size=N, reduce=reduce)
    g3 = got3[0] if isinstance(got3, tuple) else got3
    e3 = exp[0] if isinstance(exp, tuple) else exp
    assert g3.shape == (N, 4, 6)
    _close(to_np(g3).reshape(N, K), e3, dname, 4, reduce + " 3-D")


@pytest.mark.parametrize("reduce", ["max", "min", "sum", "mean"])
def test_layout_f_dim0_implicit_size_comes_out_of_the_index_transpose(gnnops, oracle, reduce, monkeypatch):
    """dim_size=None on the transposed route: index.max() is found INSIDE the index's transpose (gnnops_transpose2d_cvt_max)
    after a look at the first 2^20 ids says the destinations will not fit an LDS strip — no separate pass over the index.
    Same results as with the size given; the size itself is exact even when the largest id sits in the last row (beyond the
    ids the guess looked at), and a guess that says "fits" falls back to the ordinary pass. Threshold lowered for the test."""
    from gnnops import ops

    monkeypatch.setattr(ops, "_FUSED_MAX_MIN_NUMEL", 1)
    g = torch.Generator().manual_seed(14)
    E, K, N = 321, 40, 50_000
    src = (torch.rand(E, K, generator=g) * 2 - 1)
    idx = torch.randint(0, N - 7, (E, K), generator=g)
    idx[E - 1, K - 1] = N - 1                                     # the maximum is the very last id
    idx[:30] = idx[30:60]
    full_pass = []
    real_max = ops._index_max_now
    monkeypatch.setattr(ops, "_index_max_now", lambda i: (full_pass.append(i.numel()), real_max(i))[1])
    got = gnnops.scatter(src.cuda(), idx.cuda(), 0, reduce=reduce)
    assert full_pass == [E * K]              # one look at (at most) 2^20 ids — here all of them — and no second pass
    exp = oracle.scatter(to_np(src), idx.numpy(), dim=0, dim_size=N, reduce=reduce, dtype="f32")
    ref = gnnops.scatter(src.cuda(), idx.cuda(), 0, dim_size=N, reduce=reduce)
    if reduce in ("min", "max"):
        assert got[0].shape == (N, K)
        assert_bits_equal(to_np(got[0]), exp[0], reduce)
        assert np.array_equal(got[1].cpu().numpy(), exp[1])
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    else:
        assert got.shape == (N, K)
        _close(to_np(got), exp, "f32", 4, reduce)
    # ids that start small: the guess (from the first 64 ids here) says the destinations fit, the ordinary pass over the whole
    # index finds the true size, results unchanged
    monkeypatch.setattr(ops, "_FUSED_MAX_SAMPLE", 64)
    idx2 = idx.clone()
    idx2[:2] %= 1000
    full_pass.clear()
    got2 = gnnops.scatter(src.cuda(), idx2.cuda(), 0, reduce=reduce)
    assert full_pass == [64, E * K]
    exp2 = oracle.scatter(to_np(src), idx2.numpy(), dim=0, dim_size=N, reduce=reduce, dtype="f32")
    g2 = got2[0] if isinstance(got2, tuple) else got2
    e2 = exp2[0] if isinstance(exp2, tuple) else exp2
    assert g2.shape == (N, K)
    _close(to_np(g2), e2, "f32", 4, reduce + " (guess says fits)")


def test_implicit_dim_size_is_remembered_per_index_tensor(gnnops, oracle, monkeypatch):
    """torch_scatter's dim_size=None means int(index.max()) + 1: a full read of the index and a host round trip per call. With
    the plan cache on it is remembered per index tensor object + version (an in-place write recomputes it); with the cache off
    — the cold numbers — it is computed on every call."""
    from gnnops import ops

    calls = []
    real = ops._index_max_now
    monkeypatch.setattr(ops, "_index_max_now", lambda i: (calls.append(1), real(i))[1])
    g = torch.Generator().manual_seed(31)
    src = torch.rand(400, 32, generator=g)
    idx = torch.randint(0, 90, (400, 32), generator=g)
    d_src, d_idx = src.cuda(), idx.cuda()
    gnnops.set_plan_cache(True)
    try:
        for _ in range(3):
            out = gnnops.scatter_add(d_src, d_idx, 0)
        assert len(calls) == 1 and out.shape == (int(idx.max()) + 1, 32)
        d_idx[0, 0] = 200                                   # in-place write: new version, new maximum
        out = gnnops.scatter_add(d_src, d_idx, 0)
        assert len(calls) == 2 and out.shape == (201, 32)
        gnnops.set_plan_cache(False)
        for _ in range(2):
            gnnops.scatter_add(d_src, d_idx, 0)
        assert len(calls) == 4
    finally:
        gnnops.set_plan_cache(False)


@pytest.mark.parametrize("N,dname", [(700, "f16"), (700, "f32"), (70_000, "f32"), (65536, "bf16"), (65537, "f16")])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max", "mul"])
def test_layout_f_narrowed_index_copies(gnnops, oracle, reduce, N, dname):
    """a full-shape index passed again (same tensor, unchanged) is streamed from a 2-byte (N <= 65536)
    or 4-byte copy. First call int64, second call narrows, third reuses; an in-place write to the index invalidates the
    copy. Same results as the int64 index every time (min / max / arg bit-exact, sums within the layout-F tolerance)."""
    from gnnops import ops

    g = torch.Generator().manual_seed(N)
    L, K = 160, 48
    src = (torch.rand(L, K, generator=g) * 2 - 1).to(TORCH_DT[dname])
    idx = torch.randint(0, N, (L, K), generator=g)
    idx[0, 0] = N - 1
    d_src, d_idx = src.cuda(), idx.cuda()
    gnnops.set_plan_cache(True)
    try:
        for dim in (0, 1):
            n_out = N
            exp = oracle.scatter(to_np(src), idx.numpy(), dim=dim, dim_size=n_out, reduce=reduce, dtype=dname)
            for call in range(3):
                got = gnnops.scatter(d_src, d_idx, dim, dim_size=n_out, reduce=reduce)
                if reduce in ("min", "max"):
                    assert_bits_equal(to_np(got[0]), exp[0], f"{reduce} call {call}")
                    assert np.array_equal(got[1].cpu().numpy(), exp[1])
                else:
                    _close(to_np(got), exp, dname, 8, f"{reduce} call {call}")
            hit = ops._narrow_cache.get(id(d_idx))
            cell = 4 if reduce in ("sum", "mul") or (reduce in ("min", "max") and dname != "f32") else 8
            if dim == 0 and N * cell > ops._LDS_STRIP_BYTES:
                continue          # routed through transposes (fresh tensors every call): nothing to key a copy on
            assert hit is not None and hit[3] not in (None,), "the second call must have narrowed (or recorded why not)"
            if hit[3] != "unsupported":
                assert hit[3][1] == (2 if N <= 65535 else 4)      # 0xFFFF marks ids outside [0, N): N = 65536 takes four bytes
        d_idx[1, 1] = 0                                   # in-place write: version bump -> the copy is stale and dropped
        idx[1, 1] = 0
        exp = oracle.scatter(to_np(src), idx.numpy(), dim=0, dim_size=N, reduce=reduce, dtype=dname)
        for call in range(3):
            got = gnnops.scatter(d_src, d_idx, 0, dim_size=N, reduce=reduce)
            g0 = got[0] if isinstance(got, tuple) else got
            e0 = exp[0] if isinstance(exp, tuple) else exp
            _close(to_np(g0), e0, dname, 8, f"after write, call {call}")
    finally:
        gnnops.set_plan_cache(False)
