"""CPU suite, part 3: the multi-GPU path's host logic (gnnops.dist) on 2 gloo ranks.

Both exchange forms — the dense reduce-scatter of per-rank partial [N, D] buffers and the sparse all-to-all-v of
compact (id, row) lists — and the row ownership are exercised for real; the local reductions are numpy / oracle
stand-ins (the HIP kernels need a GPU). The expected result is the oracle on the concatenation of every rank's
edges."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import dist_worker
from oracle import oracle


@pytest.mark.timeout(180)
def test_sharded_scatter_two_ranks():
    world, n_total, e_local, d = 2, 64, 500, 8
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "init")
        mp.spawn(dist_worker.run, args=(world, init_file, n_total, e_local, d, tmp), nprocs=world, join=True)
        srcs, idxs = zip(*(dist_worker.make_inputs(r, world, n_total, e_local, d) for r in range(world)))
        src = torch.cat(srcs).numpy()
        idx = torch.cat(idxs).numpy()
        for rank in range(world):
            got = np.load(os.path.join(tmp, f"rank{rank}.npz"))
            lo, hi = int(got["lo"]), int(got["hi"])
            assert (lo, hi) == (rank * n_total // world, (rank + 1) * n_total // world)
            for r in ("sum", "min", "max", "mean"):
                exp = oracle.scatter(src, idx, dim=0, dim_size=n_total, reduce=r)
                exp = exp[0] if isinstance(exp, tuple) else exp
                if r in ("min", "max"):
                    assert np.array_equal(got[r], exp[lo:hi]), r
                else:  # the order ranks are summed in differs from the single sequential pass
                    np.testing.assert_allclose(got[r], exp[lo:hi], rtol=1e-5, atol=1e-5, err_msg=r)
            assert (got["sum"][5 - lo] == 0).all() if lo <= 5 < hi else True
            for r in ("sum", "min", "max", "mean", "mul"):
                exp = oracle.scatter(src, idx, dim=0, dim_size=n_total, reduce=r)
                exp = exp[0] if isinstance(exp, tuple) else exp
                if r in ("min", "max"):
                    assert np.array_equal(got["sparse_" + r], exp[lo:hi]), r
                else:
                    np.testing.assert_allclose(got["sparse_" + r], exp[lo:hi], rtol=1e-5, atol=1e-5, err_msg=r)
            np.testing.assert_allclose(got["sparse_sum_out"], got["sparse_sum"], rtol=0, atol=0)
            n_sum, n_mean = got["a2a_calls_sum_mean"]
            assert n_sum == 3 and n_mean == n_sum, (n_sum, n_mean)      # counts, ids, rows — a mean adds no exchange
            assert (got["sparse_sum"][5 - lo] == 0).all() if lo <= 5 < hi else True
            exp = oracle.scatter(src, idx, dim=0, dim_size=n_total, reduce="sum")
            np.testing.assert_allclose(got["compact_sum"], exp[lo:hi], rtol=1e-5, atol=1e-5)
            # (value, index) pair reduction: arg = GLOBAL (rank-major) position of the extremum — the position in the
            # concatenation of every rank's edges, which is exactly what the oracle indexes
            for r in ("min", "max"):
                ev, ea = oracle.scatter(src, idx, dim=0, dim_size=n_total, reduce=r)
                assert np.array_equal(got["arg_" + r + "_val"], ev[lo:hi]), r
                assert np.array_equal(got["arg_" + r], ea[lo:hi]), r
        # source-partitioned SpMM: the global operand is the ranks' column blocks side by side
        parts = [dist_worker.make_spmm_inputs(r, world, n_total, e_local, 40, d) for r in range(world)]
        gidx = np.concatenate([np.stack([p[0][0].numpy(), p[0][1].numpy() + 40 * r]) for r, p in enumerate(parts)], axis=1)
        gval = np.concatenate([p[1].numpy() for p in parts])
        gmat = np.concatenate([p[2].numpy() for p in parts], axis=0)
        exp = oracle.spmm(gidx, gval, n_total, 40 * world, gmat)
        exp1 = oracle.spmm(gidx, None, n_total, 40 * world, gmat)
        for rank in range(world):
            got = np.load(os.path.join(tmp, f"rank{rank}.npz"))
            lo, hi = int(got["lo"]), int(got["hi"])
            np.testing.assert_allclose(got["spmm"], exp[lo:hi], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(got["spmm_noval"], exp1[lo:hi], rtol=1e-5, atol=1e-5)


@pytest.mark.timeout(180)
def test_sharded_scatter_three_ranks_sparse_only_cut():
    """Three ranks, more destinations than edges (most rows untouched): owner slices of the compact lists, empty
    shares and the rank in the middle (ids both below and above its range)."""
    world, n_total, e_local, d = 3, 300, 40, 4
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "init")
        mp.spawn(dist_worker.run, args=(world, init_file, n_total, e_local, d, tmp), nprocs=world, join=True)
        srcs, idxs = zip(*(dist_worker.make_inputs(r, world, n_total, e_local, d) for r in range(world)))
        src, idx = torch.cat(srcs).numpy(), torch.cat(idxs).numpy()
        for rank in range(world):
            got = np.load(os.path.join(tmp, f"rank{rank}.npz"))
            lo, hi = int(got["lo"]), int(got["hi"])
            for r in ("sum", "min", "max", "mean", "mul"):
                exp = oracle.scatter(src, idx, dim=0, dim_size=n_total, reduce=r)
                exp = exp[0] if isinstance(exp, tuple) else exp
                np.testing.assert_allclose(got["sparse_" + r], exp[lo:hi], rtol=1e-5, atol=1e-5, err_msg=r)
            for r in ("min", "max"):   # ties across ranks go to the smallest global position; empty groups get E_total
                ev, ea = oracle.scatter(src, idx, dim=0, dim_size=n_total, reduce=r)
                assert np.array_equal(got["arg_" + r + "_val"], ev[lo:hi]) and np.array_equal(got["arg_" + r], ea[lo:hi]), r


@pytest.mark.timeout(180)
def test_sharded_arg_reduction_breaks_ties_by_global_position(monkeypatch):
    """(value, index) pair reduction with ties everywhere (five distinct values): the arg must be the SMALLEST global
    (rank-major) position among the extrema — what one sequential pass over the concatenated edges yields."""
    monkeypatch.setenv("GNNOPS_TEST_TIES", "1")
    world, n_total, e_local, d = 3, 30, 400, 6
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "init")
        mp.spawn(dist_worker.run, args=(world, init_file, n_total, e_local, d, tmp), nprocs=world, join=True)
        srcs, idxs = zip(*(dist_worker.make_inputs(r, world, n_total, e_local, d) for r in range(world)))
        src, idx = torch.cat(srcs).numpy(), torch.cat(idxs).numpy()
        assert len(np.unique(src)) <= 5
        for rank in range(world):
            got = np.load(os.path.join(tmp, f"rank{rank}.npz"))
            lo, hi = int(got["lo"]), int(got["hi"])
            for r in ("min", "max"):
                ev, ea = oracle.scatter(src, idx, dim=0, dim_size=n_total, reduce=r)
                assert np.array_equal(got["arg_" + r + "_val"], ev[lo:hi]), r
                assert np.array_equal(got["arg_" + r], ea[lo:hi]), r
                if lo <= 5 < hi:
                    assert (got["arg_" + r][5 - lo] == world * e_local).all()


def test_owned_rows_requires_divisibility():
    from gnnops.dist import owned_rows

    assert owned_rows(80, 3, 8) == (30, 40)
    with pytest.raises(ValueError):
        owned_rows(10, 0, 3)
