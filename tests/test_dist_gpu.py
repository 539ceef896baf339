"""GPU parity of the per-rank pieces of gnnops.dist (HipLocal) — the owner counts, the routing of one rank's edges into the
own slab and the per-owner (id, row) edge lists, the compact per-destination split (min / max / mul, with positions), and
the two ways the received lists are folded in — against the oracle, as if this GPU
were rank 1 of 3 (destinations both below and above its range). The exchange itself is covered on CPU (test_dist_cpu.py,
gloo), by bench.py's one-rank RCCL rehearsal, and end to end by the last test here: gloo ranks sharing this box's one GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def local():
    import gnnops
    from gnnops.dist import HipLocal

    gnnops.load_library()
    return HipLocal()


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o

    return o


def _inputs(E, n_total, D, seed, holes=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.rand(E, D, generator=g) * 2 - 1
    idx = torch.randint(0, n_total, (E,), generator=g)
    if holes:
        idx[idx % 7 == 3] = 0  # many untouched destinations, one heavy one
    return src, idx


@pytest.mark.parametrize("reduce", ["sum", "min", "max", "mul"])
@pytest.mark.parametrize("E,n_total,D", [(5000, 900, 16), (300, 3000, 5), (0, 30, 4), (40000, 300, 128)])
def test_split_matches_oracle(local, oracle, reduce, E, n_total, D):
    src, idx = _inputs(E, n_total, D, 11)
    lo, hi = n_total // 3, 2 * n_total // 3
    own_dense = reduce == "sum"
    want_arg = reduce in ("min", "max")
    own, ids, rows, args = local.split(src.cuda(), idx.cuda(), n_total, lo, hi, reduce, own_dense, want_arg=want_arg)
    exp = oracle.scatter(src.numpy(), idx.numpy(), dim=0, dim_size=n_total, reduce=reduce)
    earg = None
    if isinstance(exp, tuple):
        exp, earg = exp
    touched = np.bincount(idx.numpy(), minlength=n_total) > 0
    remote = touched.copy()
    remote[lo:hi] = False
    assert np.array_equal(ids.cpu().numpy(), np.nonzero(remote)[0])
    assert np.array_equal(rows.cpu().numpy(), exp[remote])  # sequential order inside a destination: bit-exact
    if want_arg:
        assert np.array_equal(args.cpu().numpy(), earg[remote])   # local position of each extremum
    part = own()
    if own_dense:
        assert np.array_equal(part.cpu().numpy(), exp[lo:hi])
        buf = torch.full((hi - lo, D), 3.0, device="cuda")
        assert own(buf) is buf and np.array_equal(buf.cpu().numpy(), exp[lo:hi])
    else:
        own_ids, own_rows, own_args = part
        assert np.array_equal(own_ids.cpu().numpy(), np.nonzero(touched[lo:hi])[0])
        assert np.array_equal(own_rows.cpu().numpy(), exp[lo:hi][touched[lo:hi]])
        if want_arg:
            assert np.array_equal(own_args.cpu().numpy(), earg[lo:hi][touched[lo:hi]])


@pytest.mark.parametrize("E,n_total,D", [(50000, 3000, 128), (7000, 9000, 16), (0, 3000, 4), (30000, 1200, 64)])
def test_owner_counts_and_route_match_oracle(local, oracle, E, n_total, D):
    """The edge-list form of the exchange (sums): owner counts, the windowed partition, the remote (id, row) pairs grouped
    by owner in source order, and the own slab — as rank 1 of 3."""
    src, idx = _inputs(E, n_total, D, 13)
    world, rank = 3, 1
    per = n_total // world
    lo, hi = rank * per, (rank + 1) * per
    counts = local.owner_counts(idx.cuda(), per, world)
    exp_counts = np.bincount(idx.numpy() // per, minlength=world)
    assert counts.cpu().tolist() == exp_counts.tolist()
    assert local.route_ready(src.cuda(), lo, hi)
    state = local.route_begin(src.cuda(), idx.cuda(), lo, hi)
    own, send_ids, send_rows = local.route(state, n_total, lo, hi, exp_counts.tolist(), rank)
    owner = idx.numpy() // per
    remote = np.nonzero(owner != rank)[0]
    pos = remote[np.argsort(owner[remote], kind="stable")]
    assert np.array_equal(send_ids.cpu().numpy(), idx.numpy()[pos])
    assert np.array_equal(send_rows.cpu().numpy(), src.numpy()[pos])
    mine = owner == rank
    exp = oracle.scatter(src.numpy()[mine], idx.numpy()[mine] - lo, dim=0, dim_size=per, reduce="sum")
    assert np.array_equal(own().cpu().numpy(), exp)              # same order as the sequential loop: bit-exact
    buf = torch.full((per, D), 3.0, device="cuda")
    assert own(buf) is buf and np.array_equal(buf.cpu().numpy(), exp)


def test_out_of_range_ids_are_noticed_not_misrouted(local):
    """gnnops_owner_counts used to clamp every id to an owner, so sharded_scatter's "counts do not add up" guard
    could never fire and an out-of-range edge was shipped (or reduced) into the wrong rows. Now such an id belongs to no
    owner: the counts fall short of E and the step raises."""
    src, idx = _inputs(5000, 3000, 16, 17)
    world, rank, per = 3, 1, 1000
    bad = idx.clone()
    bad[7], bad[4000] = -1, 3000
    counts = local.owner_counts(bad.cuda(), per, world).cpu().tolist()
    keep = np.ones(5000, bool)
    keep[[7, 4000]] = False
    assert counts == np.bincount(idx.numpy()[keep] // per, minlength=world).tolist() and sum(counts) == 4998
    state = local.route_begin(src.cuda(), bad.cuda(), rank * per, (rank + 1) * per)
    with pytest.raises(RuntimeError, match="do not add up"):
        local.route(state, 3000, rank * per, (rank + 1) * per, counts, rank)


@pytest.mark.parametrize("dname", ["f32", "bf16"])
@pytest.mark.parametrize("with_value", [True, False])
def test_spmm_split_matches_oracle(local, oracle, with_value, dname):
    from helpers import TORCH_DT, to_np

    nnz, n_total, k_local, D = 6000, 900, 70, 64
    g = torch.Generator().manual_seed(21)
    row = torch.randint(0, n_total, (nnz,), generator=g)
    row[row % 5 == 1] = 3
    col = torch.randint(0, k_local, (nnz,), generator=g)
    val = (torch.rand(nnz, generator=g) * 2 - 1).to(TORCH_DT[dname]) if with_value else None
    mat = (torch.rand(k_local, D, generator=g) * 2 - 1).to(TORCH_DT[dname])
    lo, hi = 300, 600
    own, ids, rows = local.spmm_split(row.cuda(), col.cuda(), None if val is None else val.cuda(), mat.cuda(), n_total, lo, hi)
    exp = oracle.spmm(np.stack([row.numpy(), col.numpy()]), None if val is None else to_np(val), n_total, k_local, to_np(mat),
                      dtype=dname)
    touched = np.bincount(row.numpy(), minlength=n_total) > 0
    remote = touched.copy()
    remote[lo:hi] = False
    assert np.array_equal(ids.cpu().numpy(), np.nonzero(remote)[0])
    assert np.array_equal(to_np(rows), exp[remote])
    assert np.array_equal(to_np(own()), exp[lo:hi])


def test_accumulate_and_combine(local, oracle):
    n_local, D = 500, 12
    g = torch.Generator().manual_seed(5)
    slab = torch.rand(n_local, D, generator=g)
    rows = torch.rand(800, D, generator=g)
    ids = torch.randint(0, n_local, (800,), generator=g)
    exp = slab.numpy().copy()
    np.add.at(exp, ids.numpy(), rows.numpy())
    got = local.accumulate(slab.cuda(), rows.cuda(), ids.cuda(), "sum").cpu().numpy()
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=1e-6)
    empty = local.accumulate(slab.cuda(), rows[:0].cuda(), ids[:0].cuda(), "sum")
    assert torch.equal(empty.cpu(), slab)
    for r in ("min", "max", "mul", "sum"):
        e = oracle.scatter(rows.numpy(), ids.numpy(), dim=0, dim_size=n_local, reduce=r)
        e = e[0] if isinstance(e, tuple) else e
        assert np.array_equal(local.combine(rows.cuda(), ids.cuda(), n_local, r).cpu().numpy(), e), r


def test_sharded_scatter_single_rank_rccl(oracle):
    """World size 1 over RCCL: no remote destinations, empty all-to-all — the whole call path on the device."""
    import os
    import tempfile

    import torch.distributed as dist

    from gnnops.dist import sharded_scatter

    src, idx = _inputs(3000, 400, 8, 3)
    with tempfile.TemporaryDirectory() as tmp:
        dist.init_process_group("nccl", init_method=f"file://{os.path.join(tmp, 'init')}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        try:
            for r in ("sum", "min", "mean"):
                got = sharded_scatter(src.cuda(), idx.cuda(), 400, r).cpu().numpy()
                exp = oracle.scatter(src.numpy(), idx.numpy(), dim=0, dim_size=400, reduce=r)
                exp = exp[0] if isinstance(exp, tuple) else exp
                np.testing.assert_allclose(got, exp, rtol=1e-6, atol=1e-6, err_msg=r)
            dense = sharded_scatter(src.cuda(), idx.cuda(), 400, "sum", exchange="dense").cpu().numpy()
            np.testing.assert_allclose(dense, oracle.scatter(src.numpy(), idx.numpy(), dim=0, dim_size=400, reduce="sum"),
                                       rtol=1e-6, atol=1e-6)
            # host read-backs per step: the compact exchange (min / max / mul, return_arg) reads back ONCE —
            # the per-owner counts — and sizes its lists from the host numbers; counted with PyTorch's sync debug mode
            import warnings

            d_src, d_idx = src.cuda(), idx.cuda()
            for kwargs in (dict(reduce="max"), dict(reduce="min", return_arg=True), dict(reduce="sum", exchange="compact")):
                sharded_scatter(d_src, d_idx, 400, **kwargs)              # warm: allocator, lazy initialisation
                torch.cuda.synchronize()
                torch.cuda.set_sync_debug_mode("warn")
                try:
                    with warnings.catch_warnings(record=True) as seen:
                        warnings.simplefilter("always")
                        sharded_scatter(d_src, d_idx, 400, **kwargs)
                finally:
                    torch.cuda.set_sync_debug_mode("default")
                syncs = [w for w in seen if "synchroniz" in str(w.message).lower()]
                assert len(syncs) <= 1, (kwargs, [str(w.message)[:120] for w in syncs])
        finally:
            dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,n_total,e_local,d", [(2, 6000, 20000, 128), (3, 192, 500, 8)])
def test_sharded_ops_gloo_ranks_on_one_gpu(world, n_total, e_local, d):
    """sharded_scatter / sharded_spmm end to end with the real per-GPU pieces and DEVICE tensors in the exchange: `world`
    gloo ranks sharing cuda:0 (RCCL refuses two ranks on one device); the first case takes the windowed partition."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import rehearse_dist_gloo_gpu

    rehearse_dist_gloo_gpu.check(world, n_total, e_local, d)
