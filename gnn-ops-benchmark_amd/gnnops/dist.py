"""Destination-partitioned scatter across the GPUs of one node (BASELINE config 5).

One process per GPU. Edges are partitioned by position (rank g holds src_g [E_g, D] and index_g [E_g]
with GLOBAL destination ids anywhere in [0, n_total)); destination rows are partitioned contiguously
(rank g owns rows [g*n_total/G, (g+1)*n_total/G)). The path has exactly one exchange step, in three forms:

``exchange="sparse"`` (default) — a reduce-scatter of only what is there. For sums (the headline path):
  1. `gnnops_owner_counts`: how many of this rank's edges go to each owner — one read of the index; the counts of all
     ranks are swapped (a [G]-element all-to-all) and read back: the ONE host synchronisation of the step, issued on a
     side stream so that step 2 is already running underneath it;
  2. one windowed partition of this rank's edges: those whose destination it owns are bucketed under their local id
     (stage 1 of the single-GPU one-shot scatter), all others come out set aside in source order;
  3. the set-aside positions are ordered by owner (one stable sort of their 3-bit owner ids) and their (destination id,
     source row) pairs gathered into the send lists — an EDGE list: nothing is summed before the wire, so every size is
     known from step 1 and no second read-back is needed (on a partitioned graph a rank rarely holds two edges to the
     same remote destination: compacting per destination would save ~7 % of the bytes at the price of a sort by
     destination, a unique and a second synchronisation — that is ``exchange="compact"``);
  4. the all-to-all-v of ids and rows (RCCL over xGMI; `torch.distributed` backend "nccl" is RCCL on ROCm), issued
     asynchronously, while
  5. the edges this rank owns itself are reduced straight into its slab (bucket.hip);
  6. the received rows are scatter-added into the slab (rows nothing touches are skipped, not rewritten).
  Bytes on the wire per rank = (#edges whose destination another rank owns) x (row + 8), i.e. proportional to the edge
  cut of the partition, not to n_total. RCCL has no sparse reduce-scatter; this composes one.

``exchange="compact"`` — the same exchange of per-DESTINATION partial rows: every rank first reduces its edges per distinct
  destination (plan + segment reduce) and sends compact (id, row) lists; what min / max / mul and `return_arg` use (a
  received row must be a finished partial result there), and what pays on graphs where a rank holds many edges to the same
  remote destination. One host read-back since round 3 (the per-owner counts of distinct destinations, taken from the
  touched-row mask before any list is built; `nonzero_static` sizes the lists from the host numbers).

``exchange="dense"`` — local scatter into a partial [n_total, D] buffer, then ONE `reduce_scatter_tensor`.
  Bytes on the wire per rank = (G-1)/G x n_total x row whatever the cut; kept for comparison and for
  graphs so dense that every rank touches every destination.

min / max with ``return_arg=True`` return the GLOBAL position of the extremum (rank-major: position e of rank g is
sum(E_h, h < g) + e; ties go to the smallest global position, empty groups get sum(E_h)): the (value, index) pair
reduction of SURVEY.md §8(f) — the positions travel beside the compact rows and the owner picks per destination.

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
    """The local (per-GPU) pieces of the sparse exchange, on our kernels."""

    # ---- edge-list form (sums) ----------------------------------------------------------------------------------
    def owner_counts(self, index, per, world):
        """Device int64 [world]: how many of this rank's positions go to each owner's slab of `per` rows."""
        from . import _lib
        from .ops import _on, _require_gpu, _stream, check

        _require_gpu(index)
        index = index.contiguous()
        counts = torch.empty(world, dtype=torch.int64, device=index.device)
        with _on(index.device):
            check(_lib.load().gnnops_owner_counts(index.data_ptr(), index.numel(), per, world, counts.data_ptr(), _stream()),
                  "owner_counts")
        return counts

    def route_ready(self, src, lo, hi):
        """Can `route` take this operand? (fp32 rows of whole 16-B lanes, a slab of more than one bucket.) The answer may
        differ between ranks: both forms speak the same exchange protocol."""
        return (src.is_cuda and src.dim() == 2 and src.dtype == torch.float32 and hi - lo > 256 and src.size(1) % 4 == 0
                and src.size(0) < 2 ** 31 and src.data_ptr() % 16 == 0)

    def route_begin(self, src, index, lo, hi):
        """Stage 1 of `route`: the windowed partition, enqueued BEFORE the host reads the owner counts back, so that the
        device is busy underneath that round trip. Returns the state `route` continues from."""
        from . import _lib
        from .ops import _on, _stream, check

        src, index = src.contiguous(), index.contiguous()
        E, n_loc = src.size(0), hi - lo
        if E == 0:
            return (src, index, None)
        L = _lib.load()
        ws_bytes = L.gnnops_bucket_workspace_bytes(E, n_loc)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=src.device)
        with _on(src.device):
            check(L.gnnops_bucket_partition_window(index.data_ptr(), E, lo, n_loc, ws.data_ptr(), ws_bytes, _stream()),
                  "bucket_partition_window")
        return (src, index, ws)

    def route(self, state, n_total, lo, hi, send_splits, rank):
        """Continue from `route_begin` with the owner counts known on the host: returns ``(own, send_ids, send_rows)`` —
        the (global destination id, source row) pairs of every position another rank owns, grouped by owner in source
        order (``send_splits[g]`` of them for owner g, none for ``rank``), and the callable that reduces the own part into
        a dense slab (written into ``out`` when one is passed)."""
        import ctypes

        from . import _lib
        from .ops import _on, _stream, check, index_select
        from .sparse import sort

        src, index, ws = state
        E, D = src.shape
        n_loc, dev = hi - lo, src.device
        per = n_loc
        L = _lib.load()
        n_own = send_splits[rank]
        R = E - n_own
        if sum(send_splits) != E:
            raise RuntimeError("sharded_scatter: owner counts do not add up to the number of edges (index out of range?)")
        if R:
            ko, vo, bo = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
            check(L.gnnops_bucket_layout(E, n_loc, ctypes.byref(ko), ctypes.byref(vo), ctypes.byref(bo)), "bucket_layout")
            # positions set aside by the windowed partition, ascending: everything behind the n_own own ones
            rpos = ws[vo.value + 4 * n_own: vo.value + 4 * E].view(torch.int32).long()
            ids = index_select(index, 0, rpos)
            owner = torch.div(ids, per, rounding_mode="floor").to(torch.int32)
            _, order = sort(owner, stable=True)                  # stable: source order inside an owner
            send_ids = index_select(ids, 0, order)
            send_rows = index_select(src, 0, index_select(rpos, 0, order))
        else:
            send_ids = torch.empty(0, dtype=torch.int64, device=dev)
            send_rows = torch.empty((0, D), dtype=src.dtype, device=dev)

        def own(out=None):
            slab = out if out is not None else torch.empty((n_loc, D), dtype=src.dtype, device=dev)
            if ws is None:
                return slab.zero_()
            hub_bytes = L.gnnops_hub_workspace_bytes(E, D, _lib.SUM)   # heavy destinations: csrc/hub.h
            hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=dev) if hub_bytes else None
            with _on(dev):
                check(L.gnnops_bucket_reduce_hubs(src.data_ptr(), ws.data_ptr(), slab.data_ptr(), None, E, D, n_loc, _lib.F32,
                                                  _lib.SUM, 0, hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes,
                                                  _stream()), "bucket_reduce")
            return slab

        return own, send_ids, send_rows

    def route_counts(self, state, lo, hi, n_own, recv_ids_local):
        """float32 [hi - lo]: contributions per owned destination = this rank's own edges (their local ids are the keys the
        windowed partition of `route_begin` left in the workspace) + the ids that arrived. What turns the sums of the ONE
        exchange into means: the ids travel anyway, so the counts need no second exchange."""
        import ctypes

        from . import _lib
        from .ops import check, scatter

        src, index, ws = state
        n_loc = hi - lo
        parts = [recv_ids_local]
        if ws is not None and n_own:
            ko, vo, bo = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
            check(_lib.load().gnnops_bucket_layout(src.size(0), n_loc, ctypes.byref(ko), ctypes.byref(vo), ctypes.byref(bo)),
                  "bucket_layout")
            parts.append(ws[ko.value: ko.value + 4 * n_own].view(torch.int32).long())    # local ids of the own edges
        ids = torch.cat(parts) if len(parts) > 1 else parts[0]
        if ids.numel() == 0:
            return torch.zeros(n_loc, dtype=torch.float32, device=src.device)
        return scatter(torch.ones(ids.numel(), dtype=torch.float32, device=src.device), ids, 0, None, n_loc, "sum")

    # ---- compact form (any reduce; what min / max / mul and return_arg use) -----------------------------------

    def split(self, src, index, n_total, lo, hi, reduce, own_dense, want_arg=False):
        """Reduce this rank's edges per destination and split the result by ownership.

        Returns ``(own, ids, rows, args)``: ``ids`` int64 ascending = the distinct destinations outside [lo, hi) this
        rank touches, ``rows`` [len(ids), D] their reduced rows, ``args`` (``want_arg``, min / max) int64 [len(ids), D] the
        LOCAL position of each extremum, else None. ``own`` is a callable that, when invoked, produces the own part — the
        dense slab [hi-lo, D] if ``own_dense`` (sum: untouched rows are 0, the neutral element; written into ``out`` when
        one is passed), else the compact triple (ids_own - lo, rows_own, args_own) — so the caller can start the exchange
        first and overlap the two. One plan over the global ids; its row pointer restricted to touched rows is itself a
        CSR row pointer over perm (untouched rows are empty), so compact reductions are plain segment reductions.
        """
        world = max(1, n_total // (hi - lo))
        state, counts, extra = self.split_counts(src, index, n_total, lo, hi, world, own_dense)
        host = torch.cat([counts, extra]).tolist() if extra is not None else counts.tolist()     # direct callers: read here
        n_lo = sum(host[: lo // (hi - lo)])
        return self.split_finish(state, sum(host[:world]), n_lo, host[world] if extra is not None else 0, reduce, own_dense, want_arg)

    def split_counts(self, src, index, n_total, lo, hi, world, own_dense):
        """First half of `split`, nothing read back: the plan over the global ids and, per owner, HOW MANY distinct remote
        destinations this rank touches — a device int64 [world] (``extra``: [1], the number of own destinations touched,
        when the own part is compact too). The caller reads these back together with the counts every other rank sends it
        (`_read_counts`): the ONE host synchronisation of the compact exchange; `split_finish` then sizes everything from
        the host numbers (`nonzero_static`, slices instead of masks)."""
        from .ops import Plan, _require_gpu

        _require_gpu(src, index)
        if src.dim() != 2 or index.dim() != 1 or index.numel() != src.size(0):
            raise ValueError("sharded_scatter: src must be [E, D] with a 1-D index of E destinations")
        if src.size(0) >= 2 ** 31 or n_total >= 2 ** 31:
            raise NotImplementedError("sharded_scatter(exchange='compact'): E and n_total must be < 2^31")
        src = src.contiguous()
        index = index.contiguous()
        plan = Plan(index, n_total)
        touched = plan.rowptr[1:] != plan.rowptr[:-1]
        own_mask = None if own_dense else touched[lo:hi].clone()
        touched[lo:hi] = False
        counts = touched.view(world, hi - lo).sum(1)
        extra = None if own_dense else own_mask.sum().view(1)
        return (src, plan, touched, own_mask, lo, hi, n_total), counts, extra

    def split_finish(self, state, n_remote, n_lo, n_own, reduce, own_dense, want_arg=False):
        """Second half of `split` (sizes known on the host: ``n_remote`` distinct remote destinations, ``n_lo`` of them below
        the own range, ``n_own`` own destinations touched): returns ``(own, ids, rows, args)`` as `split` documents."""
        src, plan, touched, own_mask, lo, hi, n_total = state
        E, D = src.shape
        dev = src.device
        rowptr, perm = plan.rowptr, plan.perm

        def seg(rp, n_rows, out, arg=None):
            self._seg(src, rp, perm, n_rows, out, reduce, arg)

        ids = torch.nonzero_static(touched, size=n_remote).squeeze(1)      # int64, ascending, remote only; no read-back
        own_touched = None if own_dense else torch.nonzero_static(own_mask, size=n_own).squeeze(1)
        # ids below the own range (owners 0 .. rank-1) and above it: two runs of the compact row pointer
        ids_lo, ids_hi = ids[:n_lo], ids[n_lo:]
        n_hi = n_remote - n_lo
        # compact rowptrs: untouched rows are empty, so consecutive touched rows are adjacent in perm
        crow = torch.cat([rowptr[ids_lo], rowptr[lo:lo + 1], rowptr[ids_hi], rowptr[n_total:n_total + 1]])
        rows = torch.empty((n_remote, D), dtype=src.dtype, device=dev)
        args = torch.empty((n_remote, D), dtype=torch.int64, device=dev) if want_arg else None
        seg(crow[: n_lo + 1], n_lo, rows[:n_lo], None if args is None else args[:n_lo])
        seg(crow[n_lo + 1:], n_hi, rows[n_lo:], None if args is None else args[n_lo:])

        def own(out=None):
            if own_dense:
                slab = out if out is not None else torch.empty((hi - lo, D), dtype=src.dtype, device=dev)
                seg(rowptr[lo:hi + 1], hi - lo, slab)
                return slab
            orow = torch.cat([rowptr[lo:hi][own_touched], rowptr[hi:hi + 1]])
            orows = torch.empty((n_own, D), dtype=src.dtype, device=dev)
            oargs = torch.empty((n_own, D), dtype=torch.int64, device=dev) if want_arg else None
            seg(orow, n_own, orows, oargs)
            return own_touched, orows, oargs

        return own, ids, rows, args

    @staticmethod
    def _seg(src, crow, perm, n_rows, out, reduce, arg=None):
        """out[i] = reduce over src[perm[crow[i] : crow[i+1]]] (crow int32: absolute positions into perm); ``arg`` (min /
        max) receives the position in src of each extremum."""
        from . import _lib
        from .ops import REDUCE_CODE, _dtype_code, _on, _stream, check

        if n_rows == 0:
            return
        E, D = src.shape
        L = _lib.load()
        rcode = REDUCE_CODE["sum" if reduce == "add" else reduce]
        hub_bytes = L.gnnops_hub_workspace_bytes(E, D, rcode)   # heavy destinations: csrc/hub.h
        hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=src.device) if hub_bytes else None
        with _on(src.device):
            check(L.gnnops_segment_reduce_hubs(src.data_ptr(), crow.data_ptr(), perm.data_ptr(), out.data_ptr(),
                                               arg.data_ptr() if arg is not None else None, 1,
                                               E, D, n_rows, _dtype_code(src, "sharded_scatter"), rcode, 0,
                                               hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes, _stream()),
                  "segment_reduce")

    def spmm_split(self, row, col, value, mat, n_total, lo, hi, counts_only=False):
        """Source-partitioned SpMM: this rank's nonzeros (global output row, LOCAL column into its slab `mat` [K_g, D],
        value or None) multiplied out per output row and split by ownership, like `split` with own_dense=True:
        returns (own, ids, rows). One plan over the rows; a row pointer restricted to touched rows is a CSR row pointer,
        so both the own slab and the compact remote rows are plain gnnops_spmm launches over slices of it.
        ``counts_only``: nothing is read back here — returns (own, per-owner counts of distinct remote rows [device], finish),
        and ``finish(n_remote, n_lo)`` builds (ids, rows) once the caller has the numbers on the host (`sharded_spmm`: the one
        read-back it shares with the exchange of the counts)."""
        from . import _lib
        from .ops import Plan, _dtype_code, _on, _require_gpu, _stream, check

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
            with _on(dev):
                check(L.gnnops_spmm(rp.data_ptr(), perm.data_ptr(), col.data_ptr(),
                                    value.data_ptr() if value is not None else None, mat.data_ptr(), out.data_ptr(), n_rows,
                                    D, row.numel(), mat.size(0), dt, _stream()), "spmm")

        touched = rowptr[1:] != rowptr[:-1]
        touched[lo:hi] = False
        counts = touched.view(max(1, n_total // (hi - lo)), hi - lo).sum(1)    # distinct remote rows per owner: device int64

        def finish(n_remote, n_lo):
            """Second half, sizes known on the host (one read-back, shared with the exchange of the counts)."""
            ids = torch.nonzero_static(touched, size=n_remote).squeeze(1)
            ids_lo, ids_hi = ids[:n_lo], ids[n_lo:]
            crow = torch.cat([rowptr[ids_lo], rowptr[lo:lo + 1], rowptr[ids_hi], rowptr[n_total:n_total + 1]])
            rows = torch.empty((n_remote, D), dtype=mat.dtype, device=dev)
            mm(crow[: n_lo + 1], n_lo, rows[:n_lo])
            mm(crow[n_lo + 1:], n_remote - n_lo, rows[n_lo:])
            return ids, rows

        def own(out=None):
            slab = out if out is not None else torch.empty((hi - lo, D), dtype=mat.dtype, device=dev)
            mm(rowptr[lo:hi + 1], hi - lo, slab)
            return slab

        if counts_only:
            return own, counts, finish
        host = counts.tolist()
        ids, rows = finish(sum(host), sum(host[: lo // (hi - lo)]))
        return own, ids, rows

    def accumulate(self, slab, rows, ids_local, reduce):
        """slab[ids_local[j]] (+)= rows[j], in place (sum only: the slab's untouched rows hold the neutral 0)."""
        from .ops import scatter

        if rows.size(0):
            scatter(rows, ids_local, 0, out=slab, reduce="sum")
        return slab

    def combine(self, rows, ids_local, n_local, reduce, want_arg=False):
        """Dense slab [n_local, D] from compact contributions; destinations nobody touched read 0 (torch_scatter). With
        ``want_arg`` (min / max) also the position in `rows` of each extremum (len(rows) where nothing arrived)."""
        from .ops import scatter

        res = scatter(rows, ids_local, 0, dim_size=n_local, reduce=reduce)
        if want_arg:
            return res
        return res[0] if isinstance(res, tuple) else res


def _read_counts(send_counts, world, group, extra=None):
    """Swap the per-owner counts with every rank and read both vectors back in ONE device-to-host copy: returns
    (send_splits, recv_splits) as lists — and, when ``extra`` (a small device int64 vector that needs no exchange) is
    given, its values as a third list from the same copy. Runs on the CURRENT stream (callers put it on a side stream to
    keep the device busy underneath)."""
    n_extra = 0 if extra is None else extra.numel()
    both = torch.empty(2 * world + n_extra, dtype=torch.int64, device=send_counts.device)
    both[:world] = send_counts
    if n_extra:
        both[2 * world:] = extra
    dist.all_to_all_single(both[world:2 * world], both[:world], group=group)
    host = both.tolist()
    if extra is None:
        return host[:world], host[world:2 * world]
    return host[:world], host[world:2 * world], host[2 * world:]


def _swap(tensors, send_splits, recv_splits, group):
    """All-to-all-v of row lists that are grouped by owner: returns (received tensors, work handles)."""
    n_recv = sum(recv_splits)
    outs, works = [], []
    for t in tensors:
        out = torch.empty((n_recv,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        works.append(dist.all_to_all_single(out, t, recv_splits, send_splits, group=group, async_op=True))
        outs.append(out)
    return outs, works


def _counts_by_owner(ids, per, world):
    """Device int64 [world]: sizes of the owner slices of an ascending id list."""
    bounds = torch.searchsorted(ids, torch.arange(world + 1, device=ids.device, dtype=ids.dtype) * per)
    return (bounds[1:] - bounds[:-1]).to(torch.int64)


def sharded_scatter(src_local, index_local, n_total, reduce="sum", group=None, local_scatter=None, out_slab=None,
                    exchange="sparse", local=None, return_arg=False):
    """Scatter-reduce this rank's edges into global destinations and return the slab this rank owns.

    sum / min / max / mul: one exchange (see the module docstring); mean = sums and counts, then divide.
    ``return_arg`` (min / max): returns ``(slab, arg)`` with the GLOBAL (rank-major) position of each extremum.
    `local_scatter` given without `local` selects the dense form (it is all that form needs).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = owned_rows(n_total, rank, world)
    per = hi - lo
    if local is None and local_scatter is not None:
        exchange = "dense"
    if exchange not in ("sparse", "compact", "dense"):
        raise ValueError(f"sharded_scatter: unknown exchange {exchange!r}")
    if return_arg and reduce not in ("min", "max"):
        raise ValueError("sharded_scatter: return_arg needs reduce='min' or 'max'")
    if return_arg and exchange == "dense":
        raise NotImplementedError("sharded_scatter: return_arg needs the sparse / compact exchange")
    want_mean = reduce == "mean"
    if want_mean:
        if local is None and exchange == "sparse":
            local = HipLocal()
        one_exchange = (exchange == "sparse" and hasattr(local, "route_counts") and local.route_ready(src_local, lo, hi))
        if _all_agree(one_exchange, src_local.device, group):
            reduce = "sum"      # the edge-list exchange below, plus counts from the ids that travel anyway
        else:
            kw = dict(group=group, local_scatter=local_scatter, exchange=exchange, local=local)
            sums = sharded_scatter(src_local, index_local, n_total, "sum", **kw)
            ones = torch.ones((src_local.shape[0], 1), dtype=src_local.dtype, device=src_local.device)
            cnt = sharded_scatter(ones, index_local, n_total, "sum", **kw)
            res = sums / cnt.clamp_(min=1)
            if out_slab is not None:
                out_slab.copy_(res)
                return out_slab
            return res

    if exchange in ("sparse", "compact"):
        if reduce not in _SPARSE_REDUCES:
            raise ValueError(f"sharded_scatter: reduce {reduce!r} is not supported")
        if local is None:
            local = HipLocal()
        is_sum = reduce in ("sum", "add")
        edge_list = (exchange == "sparse" and is_sum and hasattr(local, "route") and local.route_ready(src_local, lo, hi))
        direct = is_sum and out_slab is not None and out_slab.is_contiguous() and out_slab.dtype == src_local.dtype
        send_args = None
        e_sizes_host = None
        if edge_list:
            # the owner counts go round on a side stream while the partition of step 2 is already queued behind them on the
            # main one: the host waits for the counts only, the device never idles
            counts = local.owner_counts(index_local, per, world)
            ready = _mark(counts)                                  # right behind the counting kernel
            state = local.route_begin(src_local, index_local, lo, hi)
            send_splits, recv_splits = _side_stream_counts(counts, ready, world, group)
            own, send_ids, send_rows = local.route(state, n_total, lo, hi, send_splits, rank)
        elif hasattr(local, "split_counts"):
            # compact form, ONE host read-back: the per-owner counts of distinct remote destinations (and the number of own
            # ones) come from the touched-row mask before any list exists; the lists are then sized from the host numbers
            state, counts, extra = local.split_counts(src_local, index_local, n_total, lo, hi, world, is_sum)
            parts = [] if extra is None else [extra]
            if return_arg:   # every rank's edge count (positions travel as GLOBAL, rank-major positions): same read-back
                e_dev = torch.empty(world, dtype=torch.int64, device=counts.device)
                dist.all_gather_into_tensor(e_dev, torch.full((1,), src_local.shape[0], dtype=torch.int64, device=counts.device), group=group)
                parts.append(e_dev)
            if parts:
                send_splits, recv_splits, host_extra = _read_counts(counts, world, group, torch.cat(parts) if len(parts) > 1 else parts[0])
            else:
                send_splits, recv_splits = _read_counts(counts, world, group)
                host_extra = []
            n_own = host_extra[0] if extra is not None else 0
            if return_arg:
                e_sizes_host = host_extra[-world:]
            own, send_ids, send_rows, send_args = local.split_finish(state, sum(send_splits), sum(send_splits[:rank]), n_own, reduce,
                                                                     is_sum, want_arg=return_arg)
        else:
            own, send_ids, send_rows, send_args = local.split(src_local, index_local, n_total, lo, hi, reduce, is_sum,
                                                              want_arg=return_arg)
            send_splits, recv_splits = _read_counts(_counts_by_owner(send_ids, per, world), world, group)
        if send_splits[rank] != 0 and not edge_list:
            raise RuntimeError("sharded_scatter: own destinations must not enter the exchange")
        if want_mean and not edge_list:
            raise RuntimeError("sharded_scatter: ranks disagree on the exchange form of reduce='mean'")
        if edge_list:
            n_own_edges = send_splits[rank]
            send_splits = list(send_splits)
            send_splits[rank] = 0          # the own edges stay here (recv_splits[rank] is the mirror of it)
            recv_splits = list(recv_splits)
            recv_splits[rank] = 0
        payload = [send_ids, send_rows]
        if return_arg:
            if e_sizes_host is None:      # a `local` without the two-phase protocol: its own exchange and read-back
                e_sizes = torch.empty(world, dtype=torch.int64, device=src_local.device)
                dist.all_gather_into_tensor(e_sizes, torch.full((1,), src_local.shape[0], dtype=torch.int64, device=src_local.device),
                                            group=group)
                e_sizes_host = e_sizes.tolist()
            e_off, e_total = sum(e_sizes_host[:rank]), sum(e_sizes_host)
            payload.append(send_args + e_off)          # positions travel as GLOBAL (rank-major) positions
        (recv_ids, recv_rows, *recv_rest), works = _swap(payload, send_splits, recv_splits, group)
        own_part = own(out_slab) if direct else own()          # runs while the all-to-all is in flight
        for w in works:
            w.wait()
        if is_sum:
            slab = local.accumulate(own_part, recv_rows, recv_ids - lo, reduce)
            if want_mean:   # sums -> means: divide by the number of contributions (own edges + received ids), at least 1
                cnt = local.route_counts(state, lo, hi, n_own_edges, recv_ids - lo)
                slab.div_(cnt.clamp_(min=1).unsqueeze(1))
        else:
            own_ids, own_rows, own_args = own_part
            # contributions in RANK order (received from ranks below, own, received from ranks above): the first
            # minimiser among ties is then the one with the smallest global position
            n_before = sum(recv_splits[:rank])
            rows = torch.cat([recv_rows[:n_before], own_rows, recv_rows[n_before:]])
            ids_local = torch.cat([recv_ids[:n_before] - lo, own_ids, recv_ids[n_before:] - lo])
            if return_arg:
                slab, apos = local.combine(rows, ids_local, per, reduce, want_arg=True)
                args = torch.cat([recv_rest[0][:n_before], own_args + e_off, recv_rest[0][n_before:]])
                n_rows = rows.size(0)
                if n_rows:
                    arg = torch.where(apos < n_rows, args.gather(0, apos.clamp(max=n_rows - 1)),
                                      torch.full((), e_total, dtype=torch.int64, device=apos.device))
                else:
                    arg = torch.full_like(apos, e_total)
            else:
                slab = local.combine(rows, ids_local, per, reduce)
        if out_slab is not None and slab is not out_slab:
            out_slab.copy_(slab)
            slab = out_slab
        return (slab, arg) if return_arg else slab

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


def _all_agree(flag, device, group):
    """True when `flag` holds on EVERY rank (the two forms of a mean are different protocols: all ranks take the same)."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())


_side_stream = {}


def _mark(t):
    """An event on the current stream right behind the kernel that produced the device tensor `t` (None on CPU)."""
    if not t.is_cuda:
        return None
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(t.device))
    return ev


def _side_stream_counts(counts, ready, world, group):
    """`_read_counts` on a side stream that waits for `ready` only: the host blocks until the counts are back, the main
    stream keeps running what was queued behind them. CPU tensors (the gloo tests' stand-ins) have no streams: plain call."""
    if ready is None:
        return _read_counts(counts, world, group)
    dev = counts.device
    side = _side_stream.get(dev)
    if side is None:
        side = _side_stream[dev] = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        side.wait_event(ready)
        counts.record_stream(side)
        return _read_counts(counts, world, group)


def sharded_spmm(index_local, value_local, n_total, matrix_local, group=None, out_slab=None, local=None):
    """Source-partitioned SpMM across the GPUs of one node (SURVEY.md §8e "column(src)-partitioned A with B slabs").

    Rank g holds the slab ``matrix_local`` [K_g, D] of the dense operand (the features of the source nodes it owns) and
    the nonzeros whose column falls in that slab: ``index_local`` [2, nnz_g] = (GLOBAL output row, LOCAL column),
    ``value_local`` [nnz_g] or None. Returns the slab of ``A @ B`` this rank owns (rows [g*n_total/G, (g+1)*n_total/G)).
    Same single exchange as sharded_scatter(exchange="compact"): partial output rows for rows other ranks own travel as
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
    import inspect

    if "counts_only" in inspect.signature(local.spmm_split).parameters:      # one host read-back for the whole call
        own, counts, finish = local.spmm_split(index_local[0], index_local[1], value_local, matrix_local, n_total, lo, hi, counts_only=True)
        send_splits, recv_splits = _read_counts(counts, world, group)
        ids, rows = finish(sum(send_splits), sum(send_splits[:rank]))
    else:
        own, ids, rows = local.spmm_split(index_local[0], index_local[1], value_local, matrix_local, n_total, lo, hi)
        send_splits, recv_splits = _read_counts(_counts_by_owner(ids, hi - lo, world), world, group)
    if send_splits[rank] != 0:
        raise RuntimeError("sharded_spmm: own rows must not enter the exchange")
    (recv_ids, recv_rows), works = _swap([ids, rows], send_splits, recv_splits, group)
    direct = out_slab is not None and out_slab.is_contiguous() and out_slab.dtype == matrix_local.dtype
    slab = own(out_slab) if direct else own()
    for w in works:
        w.wait()
    slab = local.accumulate(slab, recv_rows, recv_ids - lo, "sum")
    if out_slab is not None and slab is not out_slab:
        out_slab.copy_(slab)
        return out_slab
    return slab
