"""CPU suite, part 1: the oracle is pinned before it is trusted.

- its 16-bit conversions against numpy / torch,
- every function against the golden vectors generated with the PyTorch CPU ops the reference calls
  (tests/golden/make_golden.py), and against live torch-CPU runs on fresh seeded inputs,
- the one hand-computable torch_scatter known answer recorded in SURVEY.md §8c.
"""
import numpy as np
import pytest
import torch

from helpers import TORCH_DT, assert_bits_equal, f32_of, from_np, load_golden, to_np
from oracle import oracle

REDUCES = ["sum", "mean", "mul", "min", "max"]


def test_f16_to_f32_exhaustive():
    L = oracle.lib()
    bits = np.arange(65536, dtype=np.uint16)
    exp = bits.view(np.float16).astype(np.float32)
    got = np.array([L.ora_f16_to_f32(int(b)) for b in bits], dtype=np.float32)
    ok = (got == exp) | (np.isnan(got) & np.isnan(exp))
    assert ok.all()


def test_f32_to_f16_and_bf16_rounding():
    L = oracle.lib()
    rng = np.random.default_rng(42)
    vals = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * 100,
        rng.standard_normal(5000).astype(np.float32) * 1e-6,
        np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e9, -1e9, 2.0**-24, 2.0**-25, 2.0**-25 * 1.0001,
                  6.1e-5, 6.0e-8, np.inf, -np.inf], dtype=np.float32),
    ])
    exp16 = vals.astype(np.float16).view(np.uint16)
    got16 = np.array([L.ora_f32_to_f16(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(got16, exp16)
    expb = to_np(torch.from_numpy(vals).to(torch.bfloat16))
    gotb = np.array([L.ora_f32_to_bf16(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(gotb, expb)
    back = np.array([L.ora_bf16_to_f32(int(b)) for b in gotb], dtype=np.float32)
    assert np.array_equal(back, f32_of(gotb, "bf16"))


def _scatter_keys(g):
    return sorted(k[:-4] for k in g.files if k.startswith("scatter_") and k.endswith("_src"))


def test_oracle_scatter_matches_golden():
    g = load_golden("scatter_golden.npz")
    keys = _scatter_keys(g)
    assert len(keys) >= 40
    for key in keys:
        dname = key.rsplit("_", 1)[1]
        dim = int(key.split("_d")[1][0])
        src, idx, N = g[key + "_src"], g[key + "_idx"], int(g[key + "_N"])
        for r in REDUCES:
            res = oracle.scatter(src, idx, dim=dim, dim_size=N, reduce=r, dtype=dname)
            if r in ("min", "max"):
                out, arg = res
                assert_bits_equal(arg, g[f"{key}_arg{r}"], f"{key} arg{r}")
            else:
                out = res
            if r == "mean" and dname != "f32":
                # torch divides in fp32 then rounds; so do we — still require exact bits
                pass
            assert_bits_equal(out, g[f"{key}_{r}"], f"{key} {r}")


def test_oracle_known_answers():
    g = load_golden("scatter_golden.npz")
    out, arg = oracle.scatter(g["known_min_src"], g["known_min_idx"], dim=1, reduce="min")
    assert_bits_equal(out, g["known_min_out"], "known min")
    assert_bits_equal(arg, g["known_min_arg"], "known min arg")
    # dim_size defaults to index.max()+1 = 3
    res = oracle.scatter(g["allsame_src"], g["allsame_idx"], dim=0, reduce="sum")
    assert res.shape == (3, 8)
    assert_bits_equal(res, g["allsame_sum"], "all-same")
    src, idx = g["ref223_src"], g["ref223_idx"].astype(np.int64)
    for dim in (0, 1):
        res = oracle.scatter(src, idx, dim=dim, reduce="sum")
        assert_bits_equal(res, g[f"ref223_d{dim}_sum"], f"ref223 d{dim}")


def test_oracle_native_matches_golden():
    g = load_golden("native_golden.npz")
    keys = sorted(k[:-3] for k in g.files if k.endswith("_in"))
    assert len(keys) >= 20
    for key in keys:
        dname = key.rsplit("_", 1)[1]
        dim = int(key.split("_d")[1][0])
        inp, idx = g[key + "_in"], g[key + "_idx"]
        assert_bits_equal(oracle.index_select(inp, dim, idx), g[key + "_index_select"], key + " index_select")
        assert_bits_equal(oracle.gather(inp, dim, g[key + "_gidx"]), g[key + "_gather"], key + " gather")
        got = oracle.index_add_(inp, dim, idx, g[key + "_source"], dtype=dname)
        assert_bits_equal(got, g[key + "_index_add"], key + " index_add_")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_vs_live_torch_cpu(seed):
    """Fresh inputs, independent torch formulations: index_add_, scatter_reduce_."""
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(seed)
    E, N, D = 300 + seed, 40, 12
    src = torch.rand(E, D, generator=g) * 2 - 1
    idx = torch.randint(0, N, (E,), generator=g)
    exp = torch.zeros(N, D).index_add_(0, idx, src)
    assert_bits_equal(oracle.scatter(src.numpy(), idx.numpy(), 0, dim_size=N), exp.numpy(), "sum R")
    full = idx.view(-1, 1).expand(E, D).contiguous()
    for r, tr in (("min", "amin"), ("max", "amax"), ("mean", "mean"), ("mul", "prod")):
        base = torch.ones(N, D) if r == "mul" else torch.zeros(N, D)
        exp = base.scatter_reduce_(0, full, src, tr, include_self=(r == "mul"))
        got = oracle.scatter(src.numpy(), idx.numpy(), 0, dim_size=N, reduce=r)
        got = got[0] if isinstance(got, tuple) else got
        assert_bits_equal(got, exp.numpy(), r)
    # dim=1 (B=E rows, K=1): the reference's index_add_ shape (benchmark_native_index_add_.py:62)
    inp = torch.rand(D, N, generator=g)
    srcT = torch.rand(D, E, generator=g)
    exp = inp.clone().index_add_(1, idx, srcT)
    assert_bits_equal(oracle.index_add_(inp.numpy(), 1, idx.numpy(), srcT.numpy()), exp.numpy(), "index_add_ dim1")
    # plan: stable counting sort == torch stable argsort
    rowptr, perm = oracle.plan(idx.numpy(), N)
    exp_perm = torch.sort(idx, stable=True).indices.numpy()
    assert np.array_equal(perm, exp_perm)
    assert np.array_equal(rowptr, np.concatenate([[0], np.bincount(idx.numpy(), minlength=N).cumsum()]))
    # fused index_select + sum
    s = oracle.index_select_sum(inp.numpy(), 1, idx.numpy())
    assert abs(s - torch.index_select(inp, 1, idx).double().sum().item()) < 1e-9 * E * D


def test_oracle_errors():
    src = np.zeros((4, 2), np.float32)
    with pytest.raises(IndexError):
        oracle.scatter(src, np.array([0, 1, 2, 9]), 0, dim_size=3)
    with pytest.raises(IndexError):
        oracle.index_select(src, 0, np.array([4]))
    out = oracle.scatter(np.zeros((0, 2), np.float32), np.zeros((0,), np.int64), 0)
    assert out.shape == (0, 2)


def test_oracle_sparse_sort_matches_golden():
    """sort / coalesce / sparse transpose / dense transpose bit-exact vs torch CPU; spmm within 1e-5 (torch's
    sparse.mm sums in its own order)."""
    g = load_golden("sparse_sort_golden.npz")
    for name in ("1d", "2d", "3d"):
        x = g[f"sort_{name}_in"]
        for dim in range(x.ndim):
            v, i = oracle.sort(x, dim)
            assert_bits_equal(v, g[f"sort_{name}_d{dim}_values"], f"sort {name} d{dim} values")
            assert_bits_equal(i, g[f"sort_{name}_d{dim}_indices"], f"sort {name} d{dim} indices")
    m, n = (int(v) for v in g["coo_mn"])
    ci, cv = oracle.coalesce(g["coo_index"], g["coo_value"], m, n)
    assert_bits_equal(ci, g["coalesce_index"], "coalesce index")
    np.testing.assert_allclose(cv, g["coalesce_value"], rtol=1e-6)
    ti, tv = oracle.transpose_sparse(g["coo_index"], g["coo_value"], m, n)
    assert_bits_equal(ti, g["transpose_index"], "transpose index")
    np.testing.assert_allclose(tv, g["transpose_value"], rtol=1e-6)
    out = oracle.spmm(g["coo_index"], g["coo_value"], m, n, g["spmm_B"])
    np.testing.assert_allclose(out, g["spmm_out"], rtol=1e-5, atol=1e-6)
    assert_bits_equal(oracle.transpose_dense(g["dense_in"]), g["dense_T"], "dense transpose")
    # sort: -0.0 and NaN conventions — equal as keys, original bits returned (== torch.sort on the CPU, bit for bit)
    x = np.array([0.0, -0.0, np.nan, -1.0, 0.0], np.float32)
    x.view(np.uint32)[2] = 0xFFC12345   # a negative NaN with a payload
    v, i = oracle.sort(x, 0)
    assert list(i) == [3, 0, 1, 4, 2] and np.isnan(v[-1]) and np.signbit(v[2]) and not np.signbit(v[1])
    tv, ti = torch.sort(torch.from_numpy(x.copy()), stable=True)
    assert np.array_equal(v.view(np.uint32), tv.numpy().view(np.uint32)) and list(ti.numpy()) == list(i)


def test_oracle_segment_and_composite_vs_torch_cpu():
    """The §8(f) restatements against independent torch-CPU formulations (per-group softmax / logsumexp / std)."""
    g = torch.Generator().manual_seed(9)
    E, N, K = 600, 50, 5
    src = torch.randn(E, K, generator=g)
    idx = torch.randint(0, N, (E,), generator=g)
    idx[idx == 4] = 5
    sm = oracle.composite(src.numpy(), idx.numpy(), N, "softmax")
    lsm = oracle.composite(src.numpy(), idx.numpy(), N, "log_softmax")
    lse = oracle.composite(src.numpy(), idx.numpy(), N, "logsumexp")
    sd = oracle.composite(src.numpy(), idx.numpy(), N, "std")
    for n in range(N):
        rows = torch.nonzero(idx == n).flatten()
        if rows.numel() == 0:
            assert abs(lse[n, 0] - np.log(np.float32(1e-12))) < 1e-3 and (sd[n] == 0).all()
            continue
        np.testing.assert_allclose(sm[rows.numpy()], torch.softmax(src[rows], 0).numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(lsm[rows.numpy()], torch.log_softmax(src[rows], 0).numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(lse[n], torch.logsumexp(src[rows], 0).numpy(), rtol=1e-5, atol=1e-6)
        if rows.numel() > 1:
            np.testing.assert_allclose(sd[n], src[rows].std(0).numpy(), rtol=1e-4, atol=1e-6)
    order = torch.sort(idx, stable=True).indices
    indptr = np.concatenate([[0], np.bincount(idx.numpy(), minlength=N).cumsum()])
    seg = oracle.segment_csr(src[order].numpy(), indptr, reduce="sum")
    assert_bits_equal(seg, torch.zeros(N, K).index_add_(0, idx[order], src[order]).numpy(), "segment_csr")
