"""Worker for test_dist_cpu.py: one gloo rank of the destination-partitioned scatter (gnnops.dist).
The HIP op cannot run on CPU, so the local reduction is injected: the oracle (test infrastructure)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def oracle_local_scatter(src, index, dim, out, dim_size, reduce):
    from oracle import oracle

    res = oracle.scatter(src.numpy(), index.numpy(), dim=dim, dim_size=dim_size, reduce=reduce)
    if isinstance(res, tuple):
        return torch.from_numpy(res[0]), torch.from_numpy(res[1])
    return torch.from_numpy(res)


class OracleLocal:
    """CPU stand-in for gnnops.dist.HipLocal (same contract), on numpy + the oracle: only the exchange logic of
    sharded_scatter(exchange="sparse") is under test here."""

    # ---- edge-list form (same contract as HipLocal.owner_counts / route_ready / route_begin / route)
    def owner_counts(self, index, per, world):
        idx = index.numpy()
        idx = idx[(idx >= 0) & (idx < per * world)]          # as gnnops_owner_counts: out-of-range ids belong to no owner
        return torch.from_numpy(np.bincount(idx // per, minlength=world).astype(np.int64))

    def route_ready(self, src, lo, hi):
        return src.dtype == torch.float32

    def route_begin(self, src, index, lo, hi):
        return (src, index, None)

    def route(self, state, n_total, lo, hi, send_splits, rank):
        src, index, _ = state
        idx = index.numpy()
        per = hi - lo
        assert int((idx // per == rank).sum()) == send_splits[rank] and sum(send_splits) == len(idx)
        remote = np.nonzero(idx // per != rank)[0]
        order = np.argsort(idx[remote] // per, kind="stable")
        pos = remote[order]
        send_ids = torch.from_numpy(idx[pos].astype(np.int64))
        send_rows = src[torch.from_numpy(pos)]

        def own(out=None):
            mine = np.nonzero(idx // per == rank)[0]
            slab = out if out is not None else torch.empty((per, src.shape[1]), dtype=src.dtype)
            slab.zero_()
            slab.index_add_(0, torch.from_numpy(idx[mine] - lo), src[torch.from_numpy(mine)])
            return slab

        return own, send_ids, send_rows

    def route_counts(self, state, lo, hi, n_own, recv_ids_local):
        src, index, _ = state
        idx = index.numpy()
        mine = idx[(idx >= lo) & (idx < hi)] - lo
        assert len(mine) == n_own
        cnt = np.bincount(mine, minlength=hi - lo) + np.bincount(recv_ids_local.numpy(), minlength=hi - lo)
        return torch.from_numpy(cnt.astype(np.float32))

    # ---- compact form, two-phase (the one-read-back protocol of HipLocal.split_counts / split_finish)
    def split_counts(self, src, index, n_total, lo, hi, world, own_dense):
        uniq = np.unique(index.numpy())
        remote = uniq[(uniq < lo) | (uniq >= hi)]
        counts = torch.from_numpy(np.bincount(remote // (hi - lo), minlength=world).astype(np.int64))
        extra = None if own_dense else torch.tensor([int(((uniq >= lo) & (uniq < hi)).sum())])
        return (src, index, n_total, lo, hi), counts, extra

    def split_finish(self, state, n_remote, n_lo, n_own, reduce, own_dense, want_arg=False):
        src, index, n_total, lo, hi = state
        own, ids, rows, args = self.split(src, index, n_total, lo, hi, reduce, own_dense, want_arg)
        assert ids.numel() == n_remote and int((ids < lo).sum()) == n_lo
        if not own_dense:
            assert own()[0].numel() == n_own
        return own, ids, rows, args

    # ---- compact form
    def split(self, src, index, n_total, lo, hi, reduce, own_dense, want_arg=False):
        from oracle import oracle

        idx = index.numpy()
        uniq, inv = np.unique(idx, return_inverse=True)
        red = oracle.scatter(src.numpy(), inv.astype(np.int64), dim=0, dim_size=len(uniq), reduce=reduce)
        arg = None
        if isinstance(red, tuple):
            red, arg = red
        remote = (uniq < lo) | (uniq >= hi)
        ids = torch.from_numpy(uniq[remote].astype(np.int64))
        rows = torch.from_numpy(np.ascontiguousarray(red[remote]))
        args = torch.from_numpy(np.ascontiguousarray(arg[remote])) if want_arg else None
        own_ids = torch.from_numpy((uniq[~remote] - lo).astype(np.int64))
        own_rows = torch.from_numpy(np.ascontiguousarray(red[~remote]))
        own_args = torch.from_numpy(np.ascontiguousarray(arg[~remote])) if want_arg else None

        def own(out=None):
            if not own_dense:
                return own_ids, own_rows, own_args
            slab = out if out is not None else torch.empty((hi - lo, src.shape[1]), dtype=src.dtype)
            slab.zero_()
            slab[own_ids] = own_rows
            return slab

        return own, ids, rows, args

    def spmm_split(self, row, col, value, mat, n_total, lo, hi):
        from oracle import oracle

        r, c = row.numpy(), col.numpy()
        uniq, inv = np.unique(r, return_inverse=True)
        red = oracle.spmm(np.stack([inv.astype(np.int64), c]), None if value is None else value.numpy(), len(uniq),
                          mat.shape[0], mat.numpy())
        remote = (uniq < lo) | (uniq >= hi)
        ids = torch.from_numpy(uniq[remote].astype(np.int64))
        rows = torch.from_numpy(np.ascontiguousarray(red[remote]))
        own_ids = torch.from_numpy((uniq[~remote] - lo).astype(np.int64))
        own_rows = torch.from_numpy(np.ascontiguousarray(red[~remote]))

        def own(out=None):
            slab = out if out is not None else torch.empty((hi - lo, mat.shape[1]), dtype=mat.dtype)
            slab.zero_()
            slab[own_ids] = own_rows
            return slab

        return own, ids, rows

    def accumulate(self, slab, rows, ids_local, reduce):
        slab.index_add_(0, ids_local, rows)
        return slab

    def combine(self, rows, ids_local, n_local, reduce, want_arg=False):
        from oracle import oracle

        res = oracle.scatter(rows.numpy(), ids_local.numpy(), dim=0, dim_size=n_local, reduce=reduce)
        if want_arg:
            return torch.from_numpy(res[0]), torch.from_numpy(res[1])
        return torch.from_numpy(res[0] if isinstance(res, tuple) else res)


def make_inputs(rank, world, n_total, e_local, d):
    g = torch.Generator().manual_seed(100 + rank)
    src = torch.rand(e_local, d, generator=g) * 2 - 1
    if os.environ.get("GNNOPS_TEST_TIES") == "1":   # few distinct values: minima / maxima tie within and across ranks
        src = torch.randint(-2, 3, (e_local, d), generator=g).float()
    idx = torch.randint(0, n_total, (e_local,), generator=g)
    idx[idx == 5] = 6  # global destination 5 receives nothing from anyone
    return src, idx


def make_spmm_inputs(rank, world, n_total, nnz_local, k_local, d):
    """Source-partitioned operand of rank `rank`: nonzeros (global row, local column), values, dense slab [k_local, d]."""
    g = torch.Generator().manual_seed(500 + rank)
    row = torch.randint(0, n_total, (nnz_local,), generator=g)
    col = torch.randint(0, k_local, (nnz_local,), generator=g)
    val = torch.rand(nnz_local, generator=g) * 2 - 1
    mat = torch.rand(k_local, d, generator=g) * 2 - 1
    return torch.stack([row, col]), val, mat


def run(rank, world, init_file, n_total, e_local, d, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        from gnnops.dist import owned_rows, sharded_scatter

        src, idx = make_inputs(rank, world, n_total, e_local, d)
        res = {}
        for r in ("sum", "min", "max", "mean"):
            res[r] = sharded_scatter(src, idx, n_total, r, local_scatter=oracle_local_scatter).numpy()
        for r in ("sum", "min", "max", "mean", "mul"):
            res["sparse_" + r] = sharded_scatter(src, idx, n_total, r, local=OracleLocal()).numpy()
        # reduce="mean" rides on the ONE exchange of the sums (the ids that travel give the counts): as many all-to-alls as a sum
        calls = {"n": 0}
        real_a2a = dist.all_to_all_single

        def counting(*a, **k):
            calls["n"] += 1
            return real_a2a(*a, **k)

        dist.all_to_all_single = counting
        try:
            sharded_scatter(src, idx, n_total, "sum", local=OracleLocal())
            n_sum, calls["n"] = calls["n"], 0
            sharded_scatter(src, idx, n_total, "mean", local=OracleLocal())
            n_mean = calls["n"]
        finally:
            dist.all_to_all_single = real_a2a
        res["a2a_calls_sum_mean"] = np.array([n_sum, n_mean])
        res["compact_sum"] = sharded_scatter(src, idx, n_total, "sum", local=OracleLocal(), exchange="compact").numpy()
        for r in ("min", "max"):
            val, arg = sharded_scatter(src, idx, n_total, r, local=OracleLocal(), return_arg=True)
            res["arg_" + r + "_val"], res["arg_" + r] = val.numpy(), arg.numpy()
        slab = torch.full((n_total // world, d), 7.0)
        got = sharded_scatter(src, idx, n_total, "sum", local=OracleLocal(), out_slab=slab)
        assert got is slab
        res["sparse_sum_out"] = slab.numpy()
        from gnnops.dist import sharded_spmm

        idx2, val, mat = make_spmm_inputs(rank, world, n_total, e_local, 40, d)
        res["spmm"] = sharded_spmm(idx2, val, n_total, mat, local=OracleLocal()).numpy()
        res["spmm_noval"] = sharded_spmm(idx2, None, n_total, mat, local=OracleLocal()).numpy()
        lo, hi = owned_rows(n_total, rank, world)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, **res)
    finally:
        dist.destroy_process_group()


def run_gpu(rank, world, init_file, n_total, e_local, d, out_dir):
    """Two gloo ranks sharing cuda:0 with the real per-GPU pieces (gnnops.dist.HipLocal): the exchange carries device
    tensors exactly as under RCCL (test_dist_gpu.py; a one-GPU box cannot hold two RCCL ranks)."""
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        import gnnops
        from gnnops.dist import owned_rows, sharded_scatter, sharded_spmm

        gnnops.load_library()
        src, idx = make_inputs(rank, world, n_total, e_local, d)
        src, idx = src.cuda(), idx.cuda()
        res = {}
        for r in ("sum", "min", "max", "mean", "mul"):
            res["sparse_" + r] = sharded_scatter(src, idx, n_total, r).cpu().numpy()
        res["compact_sum"] = sharded_scatter(src, idx, n_total, "sum", exchange="compact").cpu().numpy()
        for r in ("min", "max"):
            val, arg = sharded_scatter(src, idx, n_total, r, return_arg=True)
            res["arg_" + r + "_val"], res["arg_" + r] = val.cpu().numpy(), arg.cpu().numpy()
        slab = torch.full((n_total // world, d), 7.0, device="cuda")
        got = sharded_scatter(src, idx, n_total, "sum", out_slab=slab)
        assert got is slab
        res["sparse_sum_out"] = slab.cpu().numpy()
        res["dense_sum"] = sharded_scatter(src, idx, n_total, "sum", exchange="dense").cpu().numpy()
        idx2, val, mat = make_spmm_inputs(rank, world, n_total, e_local, 40, d)
        res["spmm"] = sharded_spmm(idx2.cuda(), val.cuda(), n_total, mat.cuda()).cpu().numpy()
        lo, hi = owned_rows(n_total, rank, world)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, **res)
    finally:
        dist.destroy_process_group()
