"""GPU suite at BASELINE config 2's FULL size (N=10M, E=50M, D=128 fp32): the oracle cannot run these in
seconds, so parity is checked through size-independent properties plus exact checks on sampled rows.

  plan         perm is a permutation; index[perm] is sorted; stable (perm ascending inside a segment);
               rowptr == exclusive scan of bincount(index)
  scatter_add  column checksum: sum over destinations == sum over sources (fp64, rel 1e-6);
               sampled destinations recomputed sequentially on the host (bit-exact)
  scatter_min  sampled destinations exact, arg points at a source row of that destination whose value
               equals the minimum and is the first such row
  index_select push == pull bit-for-bit; sampled rows equal table[index]
  mean         sum / max(count,1) on sampled rows; linearity: scatter_add(2*src) == 2*scatter_add(src) exactly
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, E, D = 10_000_000, 50_000_000, 128


@pytest.fixture(scope="module")
def data():
    import gnnops

    if torch.cuda.get_device_properties(0).total_memory < 100 * (1 << 30):
        pytest.skip("needs > 100 GB of HBM")
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(42)
    src = torch.rand(E, D, generator=g, device=dev)
    idx = torch.randint(0, N, (E,), generator=g, device=dev)
    gnnops.set_plan_cache(False)
    plan = gnnops.Plan(idx, N)
    yield gnnops, src, idx, plan
    gnnops.set_plan_cache(True)


def test_plan_properties(data):
    gnnops, src, idx, plan = data
    perm = plan.perm.long()
    assert int(perm.sum()) == E * (E - 1) // 2
    seen = torch.zeros(E, dtype=torch.bool, device=idx.device)
    seen[perm] = True
    assert bool(seen.all())
    del seen
    keys = idx[perm]
    assert bool((keys[1:] >= keys[:-1]).all()), "index[perm] not sorted"
    same = keys[1:] == keys[:-1]
    assert bool((perm[1:][same] > perm[:-1][same]).all()), "plan not stable"
    del keys, same
    counts = torch.bincount(idx, minlength=N)
    rowptr = torch.zeros(N + 1, dtype=torch.int64, device=idx.device)
    rowptr[1:] = counts.cumsum(0)
    assert torch.equal(plan.rowptr.long(), rowptr)


def _host_rows(src, idx, rows):
    """sequential fp32 reduction of the sampled destinations on the host: {n: (positions, values)}"""
    res = {}
    for n in rows:
        pos = torch.nonzero(idx == n).flatten()  # ascending
        res[n] = (pos.cpu().numpy(), src[pos].cpu().numpy())
    return res


def test_scatter_add_full_size(data):
    gnnops, src, idx, plan = data
    out = gnnops.scatter_add(src, idx, dim=0, dim_size=N)  # cold path: builds its own plan
    assert out.shape == (N, D)
    a = out.double().sum(0)
    b = src.double().sum(0)
    assert float(((a - b).abs() / b.abs()).max()) < 1e-6
    rows = [0, 1, 12345, N // 2, N - 1]
    for n, (pos, vals) in _host_rows(src, idx, rows).items():
        acc = np.zeros(D, np.float32)
        for v in vals:
            acc = acc + v
        assert np.array_equal(out[n].cpu().numpy(), acc), f"row {n}"
    out2 = gnnops.scatter_add(src * 2, plan, dim=0)
    assert torch.equal(out2, out * 2)  # scaling by 2 is exact in fp32, so linearity must hold bit-for-bit
    del out2
    mean = gnnops.scatter_mean(src, plan, dim=0)
    cnt = (plan.rowptr[1:] - plan.rowptr[:-1]).clamp(min=1).float().unsqueeze(1)
    assert torch.equal(mean, out / cnt)


def test_scatter_min_full_size(data):
    gnnops, src, idx, plan = data
    out, arg = gnnops.scatter_min(src, plan, dim=0)
    rows = [3, 777, N // 3, N - 2]
    for n, (pos, vals) in _host_rows(src, idx, rows).items():
        if len(pos) == 0:
            assert (out[n] == 0).all() and (arg[n] == E).all()
            continue
        exp = vals.min(0)
        assert np.array_equal(out[n].cpu().numpy(), exp)
        first = pos[(vals == exp).argmax(0)]
        assert np.array_equal(arg[n].cpu().numpy(), first)
    nonempty = (plan.rowptr[1:] > plan.rowptr[:-1])
    assert bool((arg[nonempty] < E).all()) and bool((arg[~nonempty] == E).all())
    # every arg points into its own destination
    sample = torch.randint(0, N, (100_000,), device=idx.device)
    sample = sample[nonempty[sample]]
    assert bool((idx[arg[sample, 0]] == sample).all())


def test_index_select_full_size(data):
    gnnops, src, idx, plan = data
    table = src[:N]  # [N, D] view of the first N source rows
    push = gnnops.index_select(table, 0, idx, plan=plan)
    rows = torch.tensor([0, 1, 999_999, E // 2, E - 1], device=idx.device)
    assert torch.equal(push[rows], table[idx[rows]])
    assert float((push.double().sum() - table.double().mul(torch.bincount(idx, minlength=N).double().unsqueeze(1)).sum()).abs()) < 1e-3
    from gnnops import ops

    saved = ops._PUSH_MIN_TABLE_BYTES
    ops._PUSH_MIN_TABLE_BYTES = 1 << 62
    try:
        pull = gnnops.index_select(table, 0, idx)
    finally:
        ops._PUSH_MIN_TABLE_BYTES = saved
    assert torch.equal(pull, push)


def test_index_add_full_size_into_a_nonzero_out(data):
    """Tensor.index_add_ (benchmark_native_index_add_.py:13-16) at config 2's size, cold (one-shot) and with a plan, starting
    from a NON-zero `self`: sampled rows recomputed sequentially on the host from self's row (bit-exact), the column checksum
    sum(result) == sum(self) + sum(source), rows nothing reaches keep self's bits, and cold == plan bit for bit."""
    gnnops, src, idx, plan = data
    g = torch.Generator(device=src.device).manual_seed(7)
    base = torch.rand(N, D, generator=g, device=src.device)
    acc = base.clone()
    ret = gnnops.index_add_(acc, 0, idx, src)                      # cold: partition + bucketed reduce from `self`
    assert ret is acc
    a, b = acc.double().sum(0), base.double().sum(0) + src.double().sum(0)
    assert float(((a - b).abs() / b.abs()).max()) < 1e-6
    rows = [0, 2, 54321, N // 2 + 1, N - 1]
    for n, (pos, vals) in _host_rows(src, idx, rows).items():
        ref = base[n].cpu().numpy().copy()
        for v in vals:
            ref = ref + v
        assert np.array_equal(acc[n].cpu().numpy(), ref), f"row {n}"
    empty = (plan.rowptr[1:] == plan.rowptr[:-1]).nonzero().flatten()[:1000]
    assert empty.numel() > 0 and torch.equal(acc[empty], base[empty])
    acc2 = base.clone()
    gnnops.index_add_(acc2, 0, plan, src)                          # the plan form (seg_rows_kernel from `self`)
    assert torch.equal(acc2, acc)


def test_scatter_max_full_size(data):
    """torch_scatter.scatter_max (benchmark_scatter_max.py:15-18) at config 2's size, cold: sampled destinations exact,
    arg = the FIRST source row of the destination holding the maximum, empty groups (0, E), every arg inside its own
    destination, and max(src) == -min(-src) with identical args (negation is exact)."""
    gnnops, src, idx, plan = data
    out, arg = gnnops.scatter_max(src, idx, dim=0, dim_size=N)     # cold path
    rows = [5, 778, N // 3 + 1, N - 3]
    for n, (pos, vals) in _host_rows(src, idx, rows).items():
        if len(pos) == 0:
            assert (out[n] == 0).all() and (arg[n] == E).all()
            continue
        exp = vals.max(0)
        assert np.array_equal(out[n].cpu().numpy(), exp)
        assert np.array_equal(arg[n].cpu().numpy(), pos[(vals == exp).argmax(0)])
    nonempty = (plan.rowptr[1:] > plan.rowptr[:-1])
    assert bool((arg[nonempty] < E).all()) and bool((arg[~nonempty] == E).all()) and bool((out[~nonempty] == 0).all())
    sample = torch.randint(0, N, (100_000,), device=idx.device)
    sample = sample[nonempty[sample]]
    assert bool((idx[arg[sample, 5]] == sample).all())
    assert bool((src[arg[sample, 5], 5] == out[sample, 5]).all())
    neg, narg = gnnops.scatter_min(-src, plan, dim=0)
    assert torch.equal(narg, arg) and torch.equal(-neg, out)
