"""Destination-partitioned scatter across the GPUs of one node (BASELINE config 5).

One process per GPU. Edges are partitioned by position (rank g holds src_g [E_g, D] and index_g [E_g]
with GLOBAL destination ids anywhere in [0, n_total)); destination rows are partitioned contiguously
(rank g owns rows [g*n_total/G, (g+1)*n_total/G)). The path has exactly one exchange step, and two forms of it:

``exchange="sparse"`` (default) — a reduce-scatter of only what is there:
  1. one windowed partition of this rank's edges: those whose destination it owns are bucketed under their local id
     (stage 1 of the single-GPU one-shot scatter), all others come out set aside in source order;
  2. only the set-aside edges are sorted by destination and reduced per DISTINCT destination into compact
     (id, row) lists, grouped by owner (ids ascend, so the groups are contiguous slices);
  3. ONE all-to-all-v of those lists (RCCL over xGMI; `torch.distributed` backend "nccl" is RCCL on ROCm) —
     issued asynchronously, while
  4. the edges this rank owns itself are reduced straight into its slab;
  5. the received rows are scatter-reduced into the slab.
  (fp32 sums; other dtypes / reduces build one full plan over the global ids instead of 1-2, same exchange.)
  Bytes on the wire per rank = (#distinct remote destinations touched) x (row + 8), i.e. proportional to the
  edge cut of the partition, not to n_total. RCCL has no sparse reduce-scatter; this composes one from
  all_to_all_single and the local segment reduce.

``exchange="dense"`` — local scatter into a partial [n_total, D] buffer, then ONE `reduce_scatter_tensor`.
  Bytes on the wire per rank = (G-1)/G x n_total x row whatever the cut; kept for comparison and for
  graphs so dense that every rank touches every destination.

The reference has no distributed code; this module is the MI355X design for config 5.
The local pieces are injectable (`local=` / `local_scatter=`) so the exchange logic is testable on CPU with the
gloo backend (tests pass numpy/oracle stand-ins there); the product default is the HIP path and refuses CPU
tensors like everything else.
"""
import torch
import torch.distributed as dist

_REDUCE_OP = {"sum": dist.ReduceOp.SUM, "add": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}
_SPARSE_REDUCES = ("sum", "add", "min", "max", "mul")


def owned_rows(n_total, rank, world):
    """Row range [lo, hi) of the destination slab rank `rank` owns."""
    if n_total % world != 0:
        raise ValueError(f"n_total={n_total} must be divisible by world size {world}")
    per = n_total // world
    return rank * per, (rank + 1) * per


class HipLocal:
    """The local (per-GPU) pieces of the sparse exchange, on our kernels: plan build + segment reduce."""

    def split(self, src, index, n_total, lo, hi, reduce, own_dense):
        """Reduce this rank's edges per destination and split the result by ownership.

        Returns ``(own, ids, rows)``: ``ids`` int64 ascending = the distinct destinations outside [lo, hi) this rank
        touches, ``rows`` [len(ids), D] their reduced rows. ``own`` is a callable that, when invoked, produces the own
        part — the dense slab [hi-lo, D] if ``own_dense`` (sum: untouched rows are 0, the neutral element; written into
        ``out`` when one is passed), else the compact pair (ids_own - lo, rows_own) — so the caller can start the
        exchange first and overlap the two.
        """
        from .ops import _require_gpu

        _require_gpu(src, index)
        if src.dim() != 2 or index.dim() != 1 or index.numel() != src.size(0):
            raise ValueError("sharded_scatter: src must be [E, D] with a 1-D index of E destinations")
        if src.size(0) >= 2 ** 31 or n_total >= 2 ** 31:
            raise NotImplementedError("sharded_scatter(exchange='sparse'): E and n_total must be < 2^31")
        src = src.contiguous()
        index = index.contiguous()
        D = src.size(1)
        windowed = (own_dense and src.dtype == torch.float32 and hi - lo > 256 and src.size(0) > 0
                    and D % 4 == 0 and src.data_ptr() % 16 == 0)
        if windowed:
            return self._split_windowed(src, index, n_total, lo, hi, reduce)
        return self._split_planned(src, index, n_total, lo, hi, reduce, own_dense)

    @staticmethod
    def _seg(src, crow, perm, n_rows, out, reduce):
        """out[i] = reduce over src[perm[crow[i] : crow[i+1]]] (crow int32: absolute positions into perm)."""
        from . import _lib
        from .ops import REDUCE_CODE, _dtype_code, _stream, check

        if n_rows == 0:
            return
        E, D = src.shape
        L = _lib.load()
        rcode = REDUCE_CODE["sum" if reduce == "add" else reduce]
        hub_bytes = L.gnnops_hub_workspace_bytes(E, D, rcode)   # heavy destinations: csrc/hub.h
        hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=src.device) if hub_bytes else None
        with torch.cuda.device(src.device):
            check(L.gnnops_segment_reduce_hubs(src.data_ptr(), crow.data_ptr(), perm.data_ptr(), out.data_ptr(), None, 1,
                                               E, D, n_rows, _dtype_code(src, "sharded_scatter"), rcode, 0,
                                               hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes, _stream()),
                  "segment_reduce")

    def _split_windowed(self, src, index, n_total, lo, hi, reduce):
        """fp32 sums: ONE windowed partition serves both sides. Positions whose destination lies in [lo, hi) are bucketed
        under their local id and reduced straight into the slab (bucket.hip, the one-shot form of the single-GPU op);
        all other positions come out of the partition set aside in order, and only those (a `cut` fraction of E) are
        sorted by destination and reduced per distinct destination."""
        import ctypes

        from . import _lib
        from .ops import _stream, check
        from .sparse import sort

        L = _lib.load()
        E, D = src.shape
        n_loc = hi - lo
        dev = src.device
        ws_bytes = L.gnnops_bucket_workspace_bytes(E, n_loc)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(L.gnnops_bucket_partition_window(index.data_ptr(), E, lo, n_loc, ws.data_ptr(), ws_bytes, _stream()),
                  "bucket_partition_window")
        ko, vo, bo = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        check(L.gnnops_bucket_layout(E, n_loc, ctypes.byref(ko), ctypes.byref(vo), ctypes.byref(bo)), "bucket_layout")
        nb = (n_loc + 255) // 256
        bptr = ws[bo.value: bo.value + 4 * (nb + 1)].view(torch.int32)
        first_remote = int(bptr[nb].item())                       # positions set aside start here
        remote_pos = ws[vo.value + 4 * first_remote: vo.value + 4 * E].view(torch.int32)   # ascending (stable)
        if remote_pos.numel():
            rpos64 = remote_pos.long()
            sorted_ids, order = sort(index[rpos64].to(torch.int32))          # stable: source order inside a destination
            ids32, counts = torch.unique_consecutive(sorted_ids, return_counts=True)
            crow = torch.zeros(ids32.numel() + 1, dtype=torch.int32, device=dev)
            crow[1:] = torch.cumsum(counts, 0)
            perm = rpos64[order].to(torch.int32)
            ids = ids32.long()
            rows = torch.empty((ids.numel(), D), dtype=src.dtype, device=dev)
            self._seg(src, crow, perm, ids.numel(), rows, reduce)
        else:
            ids = torch.empty(0, dtype=torch.int64, device=dev)
            rows = torch.empty((0, D), dtype=src.dtype, device=dev)

        def own(out=None):
            slab = out if out is not None else torch.empty((n_loc, D), dtype=src.dtype, device=dev)
            hub_bytes = L.gnnops_hub_workspace_bytes(E, D, _lib.SUM)   # heavy destinations: csrc/hub.h
            hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=dev) if hub_bytes else None
            with torch.cuda.device(dev):
                check(L.gnnops_bucket_reduce_hubs(src.data_ptr(), ws.data_ptr(), slab.data_ptr(), None, E, D, n_loc, _lib.F32,
                                                  _lib.SUM, 0, hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes,
                                                  _stream()), "bucket_reduce")
            return slab

        return own, ids, rows

    def _split_planned(self, src, index, n_total, lo, hi, reduce, own_dense):
        """Any dtype / reduce: one full plan over the global ids; its row pointer restricted to touched rows is itself a
        CSR row pointer over perm (untouched rows are empty), so compact reductions are plain segment reductions."""
        from .ops import Plan

        E, D = src.shape
        dev = src.device
        plan = Plan(index, n_total)
        rowptr, perm = plan.rowptr, plan.perm

        def seg(rp, n_rows, out):
            self._seg(src, rp, perm, n_rows, out, reduce)

        touched = rowptr[1:] != rowptr[:-1]
        own_touched = None if own_dense else touched[lo:hi].nonzero().squeeze(1)
        touched[lo:hi] = False
        ids = touched.nonzero().squeeze(1)                       # int64, ascending, remote only
        n_lo = int(torch.searchsorted(ids, lo).item())           # ids below the own range (owners 0 .. rank-1)
        n_hi = ids.numel() - n_lo
        # compact rowptrs: untouched rows are empty, so consecutive touched rows are adjacent in perm
        crow = torch.cat([rowptr[ids[:n_lo]], rowptr[lo:lo + 1], rowptr[ids[n_lo:]], rowptr[n_total:n_total + 1]])
        rows = torch.empty((ids.numel(), D), dtype=src.dtype, device=dev)
        seg(crow[: n_lo + 1], n_lo, rows[:n_lo])
        seg(crow[n_lo + 1:], n_hi, rows[n_lo:])

        def own(out=None):
            if own_dense:
                slab = out if out is not None else torch.empty((hi - lo, D), dtype=src.dtype, device=dev)
                seg(rowptr[lo:hi + 1], hi - lo, slab)
                return slab
            orow = torch.cat([rowptr[lo:hi][own_touched], rowptr[hi:hi + 1]])
            orows = torch.empty((own_touched.numel(), D), dtype=src.dtype, device=dev)
            seg(orow, own_touched.numel(), orows)
            return own_touched, orows

        return own, ids, rows

    def spmm_split(self, row, col, value, mat, n_total, lo, hi):
        """Source-partitioned SpMM: this rank's nonzeros (global output row, LOCAL column into its slab `mat` [K_g, D],
        value or None) multiplied out per output row and split by ownership, like `split` with own_dense=True:
        returns (own, ids, rows). One plan over the rows; a row pointer restricted to touched rows is a CSR row pointer,
        so both the own slab and the compact remote rows are plain gnnops_spmm launches over slices of it."""
        from . import _lib
        from .ops import Plan, _dtype_code, _require_gpu, _stream, check

        _require_gpu(row, col, mat, value)
        if row.dim() != 1 or col.shape != row.shape or mat.dim() != 2:
            raise ValueError("sharded_spmm: row / col must be 1-D of equal length, mat 2-D")
        if row.numel() >= 2 ** 31 or n_total >= 2 ** 31:
            raise NotImplementedError("sharded_spmm: nnz and n_total must be < 2^31")
        if value is not None and value.dtype != mat.dtype:
            raise RuntimeError("sharded_spmm: value and mat must have the same dtype")
        row, col, mat = row.contiguous(), col.contiguous(), mat.contiguous()
        value = value.contiguous() if value is not None else None
        L = _lib.load()
        dt = _dtype_code(mat, "sharded_spmm")
        D = mat.size(1)
        dev = mat.device
        plan = Plan(row, n_total)
        rowptr, perm = plan.rowptr, plan.perm

        def mm(rp, n_rows, out):
            if n_rows == 0:
                return
            with torch.cuda.device(dev):
                check(L.gnnops_spmm(rp.data_ptr(), perm.data_ptr(), col.data_ptr(),
                                    value.data_ptr() if value is not None else None, mat.data_ptr(), out.data_ptr(), n_rows,
                                    D, row.numel(), mat.size(0), dt, _stream()), "spmm")

        touched = rowptr[1:] != rowptr[:-1]
        touched[lo:hi] = False
        ids = touched.nonzero().squeeze(1)
        n_lo = int(torch.searchsorted(ids, lo).item())
        crow = torch.cat([rowptr[ids[:n_lo]], rowptr[lo:lo + 1], rowptr[ids[n_lo:]], rowptr[n_total:n_total + 1]])
        rows = torch.empty((ids.numel(), D), dtype=mat.dtype, device=dev)
        mm(crow[: n_lo + 1], n_lo, rows[:n_lo])
        mm(crow[n_lo + 1:], ids.numel() - n_lo, rows[n_lo:])

        def own(out=None):
            slab = out if out is not None else torch.empty((hi - lo, D), dtype=mat.dtype, device=dev)
            mm(rowptr[lo:hi + 1], hi - lo, slab)
            return slab

        return own, ids, rows

    def accumulate(self, slab, rows, ids_local, reduce):
        """slab[ids_local[j]] (+)= rows[j], in place (sum only: the slab's untouched rows hold the neutral 0)."""
        from .ops import scatter

        if rows.size(0):
            scatter(rows, ids_local, 0, out=slab, reduce="sum")
        return slab

    def combine(self, rows, ids_local, n_local, reduce):
        """Dense slab [n_local, D] from compact contributions; destinations nobody touched read 0 (torch_scatter)."""
        from .ops import scatter

        res = scatter(rows, ids_local, 0, dim_size=n_local, reduce=reduce)
        return res[0] if isinstance(res, tuple) else res


def _exchange(ids, rows, per, rank, world, group):
    """All-to-all-v of compact (id, row) lists: ids ascend, so owner o's share is one contiguous slice."""
    dev = ids.device
    bounds = torch.searchsorted(ids, torch.arange(world + 1, device=dev, dtype=ids.dtype) * per)
    send_counts = (bounds[1:] - bounds[:-1]).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    send_splits, recv_splits = send_counts.tolist(), recv_counts.tolist()
    if send_splits[rank] != 0:
        raise RuntimeError("sharded_scatter: own destinations must not enter the exchange")
    n_recv = sum(recv_splits)
    recv_ids = torch.empty(n_recv, dtype=ids.dtype, device=dev)
    recv_rows = torch.empty((n_recv,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=dev)
    works = [dist.all_to_all_single(recv_ids, ids, recv_splits, send_splits, group=group, async_op=True),
             dist.all_to_all_single(recv_rows, rows, recv_splits, send_splits, group=group, async_op=True)]
    return recv_ids, recv_rows, works


def sharded_scatter(src_local, index_local, n_total, reduce="sum", group=None, local_scatter=None, out_slab=None,
                    exchange="sparse", local=None):
    """Scatter-reduce this rank's edges into global destinations and return the slab this rank owns.

    sum / min / max / mul: one exchange (see the module docstring); mean = sums and counts, then divide.
    (arg_out across ranks needs a (value, index) pair reduction — SURVEY.md §8f — and is not provided.)
    `local_scatter` given without `local` selects the dense form (it is all that form needs).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = owned_rows(n_total, rank, world)
    if local is None and local_scatter is not None:
        exchange = "dense"
    if exchange not in ("sparse", "dense"):
        raise ValueError(f"sharded_scatter: unknown exchange {exchange!r}")
    if reduce == "mean":
        kw = dict(group=group, local_scatter=local_scatter, exchange=exchange, local=local)
        sums = sharded_scatter(src_local, index_local, n_total, "sum", **kw)
        ones = torch.ones((src_local.shape[0], 1), dtype=src_local.dtype, device=src_local.device)
        cnt = sharded_scatter(ones, index_local, n_total, "sum", **kw)
        res = sums / cnt.clamp_(min=1)
        if out_slab is not None:
            out_slab.copy_(res)
            return out_slab
        return res

    if exchange == "sparse":
        if reduce not in _SPARSE_REDUCES:
            raise ValueError(f"sharded_scatter: reduce {reduce!r} is not supported")
        if local is None:
            local = HipLocal()
        own_dense = reduce in ("sum", "add")
        own, ids, rows = local.split(src_local, index_local, n_total, lo, hi, reduce, own_dense)
        recv_ids, recv_rows, works = _exchange(ids, rows, hi - lo, rank, world, group)
        direct = own_dense and out_slab is not None and out_slab.is_contiguous() and out_slab.dtype == src_local.dtype
        own_part = own(out_slab) if direct else own()          # runs while the all-to-all is in flight
        for w in works:
            w.wait()
        if own_dense:
            slab = local.accumulate(own_part, recv_rows, recv_ids - lo, reduce)
        else:
            own_ids, own_rows = own_part
            slab = local.combine(torch.cat([own_rows, recv_rows]), torch.cat([own_ids, recv_ids - lo]), hi - lo, reduce)
        if out_slab is not None and slab is not out_slab:
            out_slab.copy_(slab)
            return out_slab
        return slab

    # ---- dense: partial [n_total, D] + one reduce-scatter
    if local_scatter is None:
        from .ops import scatter as local_scatter
    if reduce not in _REDUCE_OP:
        raise ValueError(f"sharded_scatter: reduce {reduce!r} has no single-collective dense form")
    partial = local_scatter(src_local, index_local, 0, None, n_total, reduce)
    if isinstance(partial, tuple):
        partial = partial[0]
    if reduce in ("min", "max"):
        # local empties were zero-filled (torch_scatter convention); make them neutral for the collective
        cnt = local_scatter(torch.ones((src_local.shape[0], 1), dtype=src_local.dtype, device=src_local.device),
                            index_local, 0, None, n_total, "sum")
        neutral = float("inf") if reduce == "min" else float("-inf")
        partial = torch.where(cnt > 0, partial, torch.full_like(partial, neutral))
    if out_slab is None:
        out_slab = torch.empty((hi - lo,) + tuple(partial.shape[1:]), dtype=partial.dtype, device=partial.device)
    dist.reduce_scatter_tensor(out_slab, partial.contiguous(), op=_REDUCE_OP[reduce], group=group)
    if reduce in ("min", "max"):
        out_slab = torch.where(torch.isinf(out_slab), torch.zeros_like(out_slab), out_slab)
    return out_slab


def sharded_spmm(index_local, value_local, n_total, matrix_local, group=None, out_slab=None, local=None):
    """Source-partitioned SpMM across the GPUs of one node (SURVEY.md §8e "column(src)-partitioned A with B slabs").

    Rank g holds the slab ``matrix_local`` [K_g, D] of the dense operand (the features of the source nodes it owns) and
    the nonzeros whose column falls in that slab: ``index_local`` [2, nnz_g] = (GLOBAL output row, LOCAL column),
    ``value_local`` [nnz_g] or None. Returns the slab of ``A @ B`` this rank owns (rows [g*n_total/G, (g+1)*n_total/G)).
    Same single exchange as sharded_scatter(exchange="sparse"): partial output rows for rows other ranks own travel as
    compact (id, row) lists in one all-to-all-v while the own slab is multiplied; the received rows are added in.
    (Destination-partitioned A with a replicated B needs no collective: that is a plain local ``gnnops.spmm``.)
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = owned_rows(n_total, rank, world)
    if local is None:
        local = HipLocal()
    if index_local.dim() != 2 or index_local.size(0) != 2:
        raise ValueError("sharded_spmm: index_local must be [2, nnz]")
    own, ids, rows = local.spmm_split(index_local[0], index_local[1], value_local, matrix_local, n_total, lo, hi)
    recv_ids, recv_rows, works = _exchange(ids, rows, hi - lo, rank, world, group)
    direct = out_slab is not None and out_slab.is_contiguous() and out_slab.dtype == matrix_local.dtype
    slab = own(out_slab) if direct else own()
    for w in works:
        w.wait()
    slab = local.accumulate(slab, recv_rows, recv_ids - lo, "sum")
    if out_slab is not None and slab is not out_slab:
        out_slab.copy_(slab)
        return out_slab
    return slab
