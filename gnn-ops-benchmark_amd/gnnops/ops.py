"""Tensor-level front end of the HIP kernels: the reference's op signatures, on ROCm device tensors.

Each function mirrors one call the reference's ``op_*`` bodies make (file:line under the reference
root) — same names, argument order and meaning, same error behaviour where it is observable:

  scatter / scatter_add / scatter_mean / scatter_min / scatter_max / scatter_mul
      ~ torch_scatter (op_bm_scripts/benchmark_scatter_add.py:18, benchmark_scatter_mean.py:17,
        benchmark_scatter_min.py:17, benchmark_scatter_max.py:17)
  index_select ~ torch.index_select (benchmark_native_index_select.py:14)
  index_add_   ~ Tensor.index_add_  (benchmark_native_index_add_.py:15)
  gather       ~ torch.gather       (benchmark_native_gather.py:16)
  scatter_add_ / scatter_reduce_mul_ ~ native in-place forms (benchmark_scatter_add.py:24,
        benchmark_scatter_multiply.py:44)
  index_select_sum ~ torch.index_select(...).sum() (benchmark_fused_index_select_reduce.py:12-20)

PyTorch is used for device memory and streams only. Nothing here computes on the CPU: a CPU tensor
or a missing HIP library raises.
"""
import weakref

import torch

from . import _lib
from ._lib import F32, F16, BF16, REDUCE_CODE, check

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}


# At the reference's smallest shapes ((223, 223): benchmark_scatter_add.py:40-46) a call is host-bound: what the Timer sees is how
# long Python takes to enqueue it. `torch.cuda.current_stream().cuda_stream` costs ~8 us and `with _on(d)` ~3 us
# per use (device-index resolution in Python): the raw bindings below return the same values in ~0.3 us
# (tools/host_overhead.py: torch_scatter.scatter_add at (223, 223) 43 -> 2x us per call).
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current stream of the current device, as the integer the C ABI takes."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


class _Here:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_HERE = _Here()


def _on(device):
    """`with _on(t.device):` = `with _on(t.device):`, free when that device is the current one already."""
    if _raw_device is not None and (device.index is None or device.index == _raw_device()):
        return _HERE
    return torch.cuda.device(device)


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "gnnops: expected a ROCm device tensor (device='cuda'); there is no CPU path in this library"
            )


def _refuse_grad(what, *tensors):
    """The raw ops fill `torch.empty` outputs through ctypes: autograd does not see them. Called with a tensor that
    requires grad (outside a torch.autograd.Function, where grad mode is off) they would hand back a result cut off from
    the graph — refuse instead. The differentiable entry points are gnnops.autograd.* (what `gnnops.scatter` etc. and
    the torch_scatter / torch_sparse shims export)."""
    if torch.is_grad_enabled():
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.requires_grad:
                raise NotImplementedError(
                    f"gnnops.ops.{what}: an operand requires grad but this raw entry point has no backward; call the "
                    f"gnnops.{what.split('(')[0]} / torch_scatter / torch_sparse name (autograd-aware) or detach the operand")


def _dtype_code(t, what):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise NotImplementedError(f"gnnops.{what}: dtype {t.dtype} is not supported (float32/float16/bfloat16)")


def _bek(shape, dim):
    B = 1
    for s in shape[:dim]:
        B *= s
    K = 1
    for s in shape[dim + 1:]:
        K *= s
    return B, shape[dim], K


def _norm_dim(dim, ndim, what):
    if ndim == 0:
        raise IndexError(f"{what}: 0-dim tensors are not supported")
    if dim < -ndim or dim >= ndim:
        raise IndexError(f"Dimension out of range (expected to be in range of [{-ndim}, {ndim - 1}], but got {dim})")
    return dim % ndim


def _check_index(index, what):
    if index.dtype != torch.int64:
        raise RuntimeError(f"{what}: expected index of dtype int64, got {index.dtype}")


# --------------------------------------------------------------------------------------------------
# Plan: the inverted index of a destination index (rowptr, perm), built on device by plan.hip
# --------------------------------------------------------------------------------------------------
class Plan:
    """Stable inverted index of a 1-D int64 ``index`` with values in [0, N).

    ``rowptr`` int32 [N+1], ``perm`` int32 [E]: positions e with index[e] == n are
    perm[rowptr[n]:rowptr[n+1]], ascending. Reusable across every op that shares the index
    (scatter_*, index_add_, index_select push form) — build it once per static edge_index.
    """

    __slots__ = ("rowptr", "perm", "E", "N", "col", "_csr", "__weakref__")

    def __init__(self, index, N, companion=None):
        """``companion`` (optional int64 [E], e.g. the source row of the edge list whose destination row is ``index``): small
        inputs are planned by ONE launch that also emits ``self.col`` = companion in plan order (None otherwise)."""
        _require_gpu(index)
        _check_index(index, "Plan")
        if index.dim() != 1:
            raise ValueError("Plan: index must be 1-D")
        index = index.contiguous()
        L = _lib.load()
        self.E = index.numel()
        self.N = int(N)
        dev = index.device
        self.rowptr = torch.empty(self.N + 1, dtype=torch.int32, device=dev)
        self.perm = torch.empty(max(self.E, 1), dtype=torch.int32, device=dev)
        self.col = None
        if L.gnnops_plan_small_fits(self.E, self.N):   # one workgroup, one launch (csrc/plan.hip plan_small_kernel)
            if companion is not None:
                companion = companion.contiguous()
                self.col = torch.empty(self.E, dtype=torch.int64, device=dev)
            with _on(dev):
                rc = L.gnnops_plan_build_small(index.data_ptr(), companion.data_ptr() if companion is not None else None, self.E,
                                               self.N, self.rowptr.data_ptr(), self.perm.data_ptr(),
                                               self.col.data_ptr() if self.col is not None else None, _stream())
            check(rc, "plan_build_small")
            return
        ws_bytes = L.gnnops_plan_workspace_bytes(self.E, self.N)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        with _on(dev):
            rc = L.gnnops_plan_build(index.data_ptr(), self.E, self.N, self.rowptr.data_ptr(), self.perm.data_ptr(),
                                     ws.data_ptr(), ws_bytes, _stream())
        check(rc, "plan_build")
        # ws is returned to the caching allocator here; stream-ordered reuse keeps that safe.

    # --- persistence: a plan of a static edge_index can be built once and shipped with the dataset ---
    def state_dict(self):
        return {"rowptr": self.rowptr.cpu(), "perm": self.perm[: max(self.E, 0)].cpu(), "E": self.E, "N": self.N, "format": 1}

    @classmethod
    def from_state_dict(cls, state, device="cuda"):
        if state.get("format") != 1:
            raise ValueError("Plan.from_state_dict: unknown format")
        self = object.__new__(cls)
        self.col = None
        self.E, self.N = int(state["E"]), int(state["N"])
        self.rowptr = state["rowptr"].to(device=device, dtype=torch.int32).contiguous()
        perm = state["perm"].to(device=device, dtype=torch.int32).contiguous()
        self.perm = perm if perm.numel() else torch.empty(1, dtype=torch.int32, device=device)
        if self.rowptr.numel() != self.N + 1 or perm.numel() != self.E:
            raise ValueError("Plan.from_state_dict: inconsistent sizes")
        return self

    def save(self, path):
        torch.save(self.state_dict(), path)

    @classmethod
    def load(cls, path, device="cuda"):
        return cls.from_state_dict(torch.load(path, map_location="cpu", weights_only=True), device)


_plan_cache = {}          # id(index tensor) -> (weakref, version, N, Plan)
_plan_cache_enabled = True
_PLAN_CACHE_MAX = 8


def set_plan_cache(enabled):
    """Enable/disable reuse of plans across calls that pass the same (unmodified) index tensor object."""
    global _plan_cache_enabled
    _plan_cache_enabled = bool(enabled)
    if not enabled:
        _plan_cache.clear()
        _narrow_cache.clear()
        _max_cache.clear()


def clear_plan_cache():
    _plan_cache.clear()
    _narrow_cache.clear()
    _max_cache.clear()


def _version_of(t):
    """Version counter of a tensor, or None when it has none: tensors created under torch.inference_mode() do not
    track in-place writes (``t._version`` raises), so nothing derived from their contents may be cached."""
    return None if t.is_inference() else t._version


def get_plan(index, N, owner=None, tag=0, companion=None):
    """Plan for ``index`` (cached per tensor object + version counter while the cache is enabled).

    ``owner`` (default: ``index`` itself) is the tensor object the cache entry is tied to: a row of a COO
    ``edge_index`` is a fresh view object on every call, so its plan is cached under the [2, E] parent, with ``tag``
    telling the rows apart. An index created under ``torch.inference_mode()`` has no version counter, so an in-place
    write to it could not be noticed: its plan is rebuilt on every call (pass an explicit ``Plan`` to reuse one)."""
    if isinstance(index, Plan):
        if index.N != N:
            raise ValueError(f"Plan was built for N={index.N}, op needs N={N}")
        return index
    owner = index if owner is None else owner
    if not _plan_cache_enabled or owner.is_inference():
        return Plan(index, N, companion)
    key = (id(owner), tag)
    hit = _plan_cache.get(key)
    if hit is not None:
        ref, version, n, plan = hit
        if ref() is owner and version == owner._version and n == N:
            return plan
    plan = Plan(index, N, companion)
    if len(_plan_cache) >= _PLAN_CACHE_MAX:
        _plan_cache.pop(next(iter(_plan_cache)))

    def _drop(_ref, key=key):
        _plan_cache.pop(key, None)

    _plan_cache[key] = (weakref.ref(owner, _drop), owner._version, N, plan)
    return plan


# ---- narrowed copies of full-shape (layout F) indices -------------------------------------------------------------
# At the reference's own shapes (benchmark_scatter_add.py:67-84: src fp16 (L, L), index int64 (L, L)) the index is 8 of every
# 10 bytes the op reads. An index tensor that is passed a second time (unchanged: same object, same version counter) gets a
# 2-byte (N <= 65536) or 4-byte copy, kept while the tensor lives; the element kernels then stream that. The first call
# pays nothing extra, the second the narrowing pass (12 B per entry once), every later one reads 2-4 B instead of 8.
_narrow_cache = {}        # id(index) -> [weakref, version, bound, None | (narrow tensor, bytes) | "unsupported"]
_NARROW_CACHE_MAX = 8


def _narrowed(index, full, bound):
    """(tensor, bytes per entry) to hand to the element kernels for the full-shape index `full` of the user's `index`."""
    if not _plan_cache_enabled or bound >= 2 ** 31 or _version_of(index) is None:
        return full, 8
    key = id(index)
    what = (bound, tuple(full.shape))     # an index that broadcasts ([1, K]) can meet a src of another E: the copy is per SHAPE
    hit = _narrow_cache.get(key)
    if hit is None or hit[0]() is not index or hit[1] != index._version or hit[2] != what:
        if len(_narrow_cache) >= _NARROW_CACHE_MAX:
            _narrow_cache.pop(next(iter(_narrow_cache)))
        _narrow_cache[key] = [weakref.ref(index, lambda _r, key=key: _narrow_cache.pop(key, None)), index._version, what, None]
        return full, 8                                  # first sighting: nothing to amortise yet
    if hit[3] == "unsupported":
        return full, 8
    if hit[3] is None:
        nbytes = 2 if bound <= 65535 else 4             # 0xFFFF / -1 mark ids outside [0, bound): never a valid id
        narrow = torch.empty(full.shape, dtype=torch.int16 if nbytes == 2 else torch.int32, device=full.device)
        with _on(full.device):
            check(_lib.load().gnnops_narrow_index(full.data_ptr(), narrow.data_ptr(), full.numel(), nbytes, bound, _stream()), "narrow_index")
        hit[3] = (narrow, nbytes)
    return hit[3]


def _narrow_unsupported(index):
    hit = _narrow_cache.get(id(index))
    if hit is not None:
        hit[3] = "unsupported"


_max_cache = {}           # id(index) -> (weakref, version, max): torch_scatter's implicit dim_size of an index seen before (warm only)


def index_max(index):
    """int(index.max()) computed by our reduction kernel; -1 for an empty index. Synchronises (like the
    reference's implicit ``index.max()`` in torch_scatter when dim_size is None). While the plan cache is on, the value is
    remembered per index tensor object + version counter like everything else derived from an index (a full read of the
    index and a host round trip: 15-35 % of a call at the reference's (6708, 6708) shapes); `set_plan_cache(False)` — the
    cold numbers — computes it every time."""
    _require_gpu(index)
    _check_index(index, "index_max")
    value = _remembered_index_max(index)
    if value is None:
        value = _index_max_now(index)
        _remember_index_max(index, value)
    return value


def _remember_index_max(index, value):
    if _plan_cache_enabled and _version_of(index) is not None:
        if len(_max_cache) >= _PLAN_CACHE_MAX:
            _max_cache.pop(next(iter(_max_cache)))
        key = id(index)
        _max_cache[key] = (weakref.ref(index, lambda _r, key=key: _max_cache.pop(key, None)), index._version, value)


def _remembered_index_max(index):
    if _plan_cache_enabled and _version_of(index) is not None:
        hit = _max_cache.get(id(index))
        if hit is not None and hit[0]() is index and hit[1] == index._version:
            return hit[2]
    return None


def _index_max_now(index):
    index = index.contiguous()
    out = torch.empty(1, dtype=torch.int64, device=index.device)
    with _on(index.device):
        rc = _lib.load().gnnops_index_max(index.data_ptr(), index.numel(), out.data_ptr(), _stream())
    check(rc, "index_max")
    return int(out.item())


# --------------------------------------------------------------------------------------------------
# scatter family
# --------------------------------------------------------------------------------------------------
_SCATTER1D_MIN_N = 1 << 22     # 1-D min / max over at least this many destinations takes the carried-value form (scatter1d.hip)
_LDS_STRIP_BYTES = 160 * 1024 - 512   # csrc/scatter_elem.hip LDS_BUDGET: what one workgroup's strip of destinations may take
_FUSED_MAX_MIN_NUMEL = 1 << 26        # from here on a pass over the index costs more than the extra small read-back that saves it
_FUSED_MAX_SAMPLE = 1 << 20           # ids looked at to guess whether the destinations will fit an LDS strip


def _row_index_of(index, src, dim):
    """Return a contiguous 1-D index if `index` is (a broadcast of) a row index along `dim`, else None."""
    if index.dim() == 1 and index.numel() == src.size(dim):
        return index.contiguous()
    if index.dim() == src.dim() and index.shape == src.shape:
        if index.is_contiguous() and index.numel() != src.size(dim):
            return None     # a contiguous full-shape index with more than one column is no broadcast (host time matters at small shapes)
        # index.view(-1,1).expand_as(src) and friends: stride 0 everywhere but `dim`
        if all(index.stride(d) == 0 or index.size(d) == 1 for d in range(index.dim()) if d != dim):
            sl = [0] * index.dim()
            sl[dim] = slice(None)
            return index[tuple(sl)].contiguous()
    return None


def _broadcast_index(index, src, dim):
    """torch_scatter.utils.broadcast: lift a lower-rank index to src's shape."""
    if index.dim() == 1 and src.dim() > 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        index = index.view(shape)
    while index.dim() < src.dim():
        index = index.unsqueeze(-1)
    return index.expand(src.shape)


def _scatter_transposed(src2, full, N, reduce, rcode, dt, L):
    """The dim-0 route of a large full-index scatter (src2, full: [E, K]; N destinations per column; the reference's
    (38000, 38000) shapes): reduce along the LAST dim of the transposed operands, where the element kernel streams whole rows.
    Of the 69 GB that route moved in transposes at (38000)^2, 46 were the int64 index going in and the int64 arg coming out:
    the index is narrowed to int32 INSIDE its transpose and the positions come out of the kernel as int32 rows, widened inside
    the transpose back (gnnops_transpose2d_cvt, gnnops_scatter_elementwise_ixa). Returns out [N, K] (and arg), or None when
    the kernel does not take the narrowed operands (the caller then takes the int64 route). N = None: the size is
    index.max() + 1, found inside the index's transpose; an int (that size) is returned instead of None when the route does
    not apply after all."""
    from .sparse import transpose_contiguous

    E, K = src2.shape
    if E >= 2 ** 31 or (N is not None and N >= 2 ** 31):
        return None
    dev = src2.device
    want_arg = rcode in (_lib.MIN, _lib.MAX)
    src_t = transpose_contiguous(src2)                                          # [K, E]
    idx_t = torch.empty((K, E), dtype=torch.int32, device=dev)
    if N is None:
        # torch_scatter's implicit dim_size: the index's transpose reads every id anyway and leaves the largest behind — one
        # host read-back as before, without the extra pass over the index (1.7 of 21.5 ms at (38000, 38000))
        top = torch.empty(1, dtype=torch.int64, device=dev)
        with _on(dev):
            check(L.gnnops_transpose2d_cvt_max(full.data_ptr(), idx_t.data_ptr(), E, K, top.data_ptr(), _stream()), "transpose2d_cvt_max")
        N = int(top.item()) + 1
        if N >= 2 ** 31:
            return N            # the int32 copy is meaningless: the caller takes the int64 route with the size it now knows
        narrowed = True
    else:
        narrowed = False
    out_t = torch.empty((K, N), dtype=src2.dtype, device=dev)
    arg_t = torch.empty((K, N), dtype=torch.int32, device=dev) if want_arg else None
    ws_bytes = L.gnnops_scatter_elementwise_workspace_bytes(K, N, 1, dt, rcode)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    with _on(dev):
        if not narrowed:
            check(L.gnnops_transpose2d_cvt(full.data_ptr(), idx_t.data_ptr(), E, K, 0, _stream()), "transpose2d_cvt")
        rc = L.gnnops_scatter_elementwise_ixa(src_t.data_ptr(), idx_t.data_ptr(), 4, out_t.data_ptr(),
                                              arg_t.data_ptr() if want_arg else None, 4 if want_arg else 8, K, E, 1, N, dt, rcode, 0,
                                              ws.data_ptr(), ws_bytes, _stream())
        if rc == _lib.EUNSUPPORTED:
            return N if narrowed else None
        check(rc, "scatter_elementwise")
        out = transpose_contiguous(out_t)                                       # [N, K]
        if not want_arg:
            return out
        arg = torch.empty((N, K), dtype=torch.int64, device=dev)
        check(L.gnnops_transpose2d_cvt(arg_t.data_ptr(), arg.data_ptr(), K, N, 1, _stream()), "transpose2d_cvt")
    return out, arg


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    """torch_scatter.scatter(src, index, dim, out, dim_size, reduce). min/max return (out, arg_out)."""
    if reduce not in REDUCE_CODE:
        raise ValueError(f"scatter: unknown reduce {reduce!r}")
    rcode = REDUCE_CODE[reduce]
    is_plan = isinstance(index, Plan)
    _require_gpu(src, out, None if is_plan else index)
    _refuse_grad("scatter", src, out)
    dt = _dtype_code(src, "scatter")
    dim = _norm_dim(dim, src.dim(), "scatter")
    src = src.contiguous()
    B, E, K = _bek(src.shape, dim)
    L = _lib.load()

    if is_plan:
        row_index = index
        if index.E != E:
            raise ValueError(f"scatter: plan has {index.E} positions, src.size({dim}) is {E}")
    else:
        _check_index(index, "scatter")
        row_index = _row_index_of(index, src, dim)

    if out is not None:
        if out.dtype != src.dtype or not out.is_contiguous():
            raise RuntimeError("scatter: out must be contiguous and of src's dtype")
        N = out.size(dim)
        init = 1
        if reduce == "mean":
            # torch_scatter.scatter_mean with out=: the sums are accumulated INTO out, then out is divided by the per-group
            # count (clamped to 1) — out = (out + sum) / max(count, 1). The count is one more (row) scatter of ones.
            scatter(src, index, dim, out, None, "sum")
            if row_index is not None:
                cnt = scatter(torch.ones(E, dtype=torch.float32, device=src.device), row_index, 0, None, N, "sum")
                shape = [1] * out.dim()
                shape[dim] = N
                cnt = cnt.view(shape)
            else:
                cnt = scatter(torch.ones(src.shape, dtype=torch.float32, device=src.device), index, dim, None, N, "sum")
            out.div_(cnt.clamp_(min=1))
            return out
    else:
        init = 0
        if dim_size is not None:
            N = int(dim_size)
        elif is_plan:
            N = index.N
        else:
            N = None
            if (row_index is None and B == 1 and K > 1 and E < 2 ** 31 and src.numel() >= _FUSED_MAX_MIN_NUMEL
                    and index.shape == src.shape and index.is_contiguous() and _remembered_index_max(index) is None):
                # A big layout-F scatter along dim 0 with an implicit dim_size. If its destinations will not fit an LDS strip
                # it takes the transposed route below, whose index transpose can find index.max() on the way. Whether they
                # fit is a question about that same max: the first 2^20 ids answer it (uniform ids: to a part in a million);
                # a wrong guess costs time, never correctness — both routes are complete for any N.
                cell = 4 if (reduce in ("sum", "add", "mul") or (reduce in ("min", "max") and src.element_size() == 2 and E < 65535)) else 8
                guess = _index_max_now(index.view(-1)[:_FUSED_MAX_SAMPLE]) + 1
                if guess * cell > _LDS_STRIP_BYTES:
                    res = _scatter_transposed(src.view(E, K), index.view(E, K), None, reduce, rcode, dt, L)
                    if isinstance(res, int):
                        N = res                      # route refused after the size was found: carry on with the size
                        _remember_index_max(index, N - 1)
                    else:
                        n_out = (res[0] if isinstance(res, tuple) else res).size(0)
                        _remember_index_max(index, n_out - 1)
                        shape = list(src.shape)
                        shape[dim] = n_out
                        if isinstance(res, tuple):
                            return res[0].view(shape), res[1].view(shape)
                        return res.view(shape)
            if N is None:
                N = index_max(index if row_index is None else row_index) + 1 if index.numel() else 0
        if row_index is None and B == 1 and K > 1 and E < 2 ** 31:
            cell = 4 if (reduce in ("sum", "add", "mul") or (reduce in ("min", "max") and src.element_size() == 2 and E < 65535)) else 8
            if N * cell > _LDS_STRIP_BYTES:
                # layout F along dim 0 of a matrix whose destinations do not fit an LDS strip (the reference's (38000, 38000)
                # shapes, data/scatter_max.csv:32-33): the element kernel would re-scan every column strip once per chunk of
                # destinations, through 16-B-wide column pieces. Along the LAST dim the same problem streams whole rows, so:
                # tile-transpose src and index, reduce along dim 1, transpose the results back (streaming copies).
                from .sparse import transpose_contiguous

                shape = list(src.shape)
                shape[dim] = N
                full = _broadcast_index(index, src, dim).contiguous().view(E, K)      # B == 1: [E, K] around `dim`
                res = _scatter_transposed(src.view(E, K), full, N, reduce, rcode, dt, L)
                if res is None:      # the element kernel does not take the narrowed operands at this shape: int64 all the way
                    res = scatter(transpose_contiguous(src.view(E, K)), transpose_contiguous(full), 1, None, N, reduce)
                    res = tuple(transpose_contiguous(r) for r in res) if isinstance(res, tuple) else transpose_contiguous(res)
                if isinstance(res, tuple):
                    return res[0].view(shape), res[1].view(shape)
                return res.view(shape)
        shape = list(src.shape)
        shape[dim] = N
        out = torch.empty(shape, dtype=src.dtype, device=src.device)
    want_arg = rcode in (_lib.MIN, _lib.MAX)
    arg = torch.empty(out.shape, dtype=torch.int64, device=src.device) if want_arg else None

    if row_index is not None and not is_plan and (E >= 2 ** 31 or N >= 2 ** 31):
        row_index = None  # beyond the plan's int32 range: the element-wise kernel takes int64 sizes
    if (want_arg and init == 0 and row_index is not None and not is_plan and B == 1 and K == 1 and _SCATTER1D_MIN_N <= N < 2 ** 31 - 32768
            and 0 < E < 2 ** 31):
        # a long 1-D min / max (the reference's 1.47e9-element shapes): the value travels with its destination through a PARTIAL
        # sort and a workgroup finishes each bucket of 32768 destinations in LDS (scatter1d.hip) — no full sort, no random gather
        ws_bytes = L.gnnops_scatter1d_workspace_bytes(E, N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=src.device)
        with _on(src.device):
            rc = L.gnnops_scatter1d_minmax(src.data_ptr(), row_index.data_ptr(), out.data_ptr(), arg.data_ptr(), E, N, dt, rcode,
                                           ws.data_ptr(), ws_bytes, _stream())
        if rc != _lib.EUNSUPPORTED:
            check(rc, "scatter1d_minmax")
            return out, arg
    if (not want_arg and init == 0 and row_index is not None and not is_plan and B == 1 and K == 1 and _SCATTER1D_MIN_N <= N < 2 ** 31 - 32768
            and 0 < E < 2 ** 31):
        # a long 1-D sum / mean / product: the same carried-value partial sort, one pass further (buckets of 256 destinations),
        # finished on chip in source order — bit-identical to the sequential loop (scatter1d.hip)
        ws_bytes = L.gnnops_scatter1d_workspace_bytes(E, N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=src.device)
        with _on(src.device):
            rc = L.gnnops_scatter1d_sum(src.data_ptr(), row_index.data_ptr(), out.data_ptr(), E, N, dt, rcode, ws.data_ptr(), ws_bytes, _stream())
        if rc != _lib.EUNSUPPORTED:
            check(rc, "scatter1d_sum")
            return out
    with _on(src.device):
        vec = 16 // src.element_size()
        rows_ok = K % vec == 0 and src.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0     # rows of whole 16-B lanes
        if row_index is not None and not is_plan and B == 1 and not _plan_cache_enabled and E > 0 and rows_ok and N > 256 and (
                src.dtype == torch.float32 or want_arg):
            # nothing will reuse a plan: fold the end of its construction into the reduction (bucket.hip). 16-bit
            # sums / means / products stay on the plan path (one rounding of the fp32 accumulator, whatever the skew).
            ws_bytes = L.gnnops_bucket_workspace_bytes(E, N)
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=src.device)
            rc = L.gnnops_bucket_partition(row_index.data_ptr(), E, N, ws.data_ptr(), ws_bytes, _stream())
            if rc == _lib.OK:
                hub_bytes = L.gnnops_hub_workspace_bytes(E, K, rcode)   # heavy destinations are reduced piecewise (hub.h)
                hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=src.device) if hub_bytes else None
                rc = L.gnnops_bucket_reduce_hubs(src.data_ptr(), ws.data_ptr(), out.data_ptr(),
                                                 arg.data_ptr() if want_arg else None, E, K, N, dt, rcode, init,
                                                 hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes, _stream())
            if rc != _lib.EUNSUPPORTED:
                check(rc, "scatter_rows_oneshot")
                return (out, arg) if want_arg else out
        if row_index is not None:
            plan = get_plan(row_index, N)
            hub_bytes = L.gnnops_hub_workspace_bytes(E, K, rcode) if B == 1 else 0   # heavy destinations: hub.h
            hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=src.device) if hub_bytes else None
            rc = L.gnnops_segment_reduce_hubs(src.data_ptr(), plan.rowptr.data_ptr(), plan.perm.data_ptr(), out.data_ptr(),
                                              arg.data_ptr() if want_arg else None, B, E, K, N, dt, rcode, init,
                                              hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes, _stream())
            check(rc, "segment_reduce")
        else:
            full = index if (index.shape == src.shape and index.is_contiguous()) else _broadcast_index(index, src, dim).contiguous()
            ws_bytes = L.gnnops_scatter_elementwise_workspace_bytes(B, N, K, dt, rcode)
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=src.device)
            ix, ix_bytes = _narrowed(index, full, N)
            rc = L.gnnops_scatter_elementwise_ix(src.data_ptr(), ix.data_ptr(), ix_bytes, out.data_ptr(),
                                                 arg.data_ptr() if want_arg else None, B, E, K, N, dt, rcode, init,
                                                 ws.data_ptr(), ws_bytes, _stream())
            if rc == _lib.EUNSUPPORTED and ix_bytes != 8:   # not the LDS-strip form: the int64 index, and remember it
                _narrow_unsupported(index)
                rc = L.gnnops_scatter_elementwise_ix(src.data_ptr(), full.data_ptr(), 8, out.data_ptr(),
                                                     arg.data_ptr() if want_arg else None, B, E, K, N, dt, rcode, init,
                                                     ws.data_ptr(), ws_bytes, _stream())
            check(rc, "scatter_elementwise")
    return (out, arg) if want_arg else out


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "sum")


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "sum")


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "mean")


def scatter_mul(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "mul")


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "min")


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "max")


def scatter_add_(self, dim, index, src):
    """Tensor.scatter_add_(dim, index, src): in place, returns self."""
    _inplace_scatter(self, dim, index, src, "sum", "scatter_add_")
    return self


def scatter_reduce_mul_(self, dim, index, src):
    """Tensor.scatter_(dim, index, src, reduce='multiply'): in place, returns self."""
    _inplace_scatter(self, dim, index, src, "mul", "scatter_")
    return self


def _inplace_scatter(self, dim, index, src, reduce, what):
    _require_gpu(self, index, src)
    _check_index(index, what)
    if self.dtype != src.dtype:
        raise RuntimeError(f"{what}: expected self and src to have the same dtype")
    if not self.is_contiguous():
        raise NotImplementedError(f"{what}: self must be contiguous")
    dim = _norm_dim(dim, self.dim(), what)
    if index.dim() != src.dim() or index.dim() != self.dim():
        raise RuntimeError(f"{what}: index, src and self must have the same number of dimensions")
    if index.shape != src.shape:
        # ATen allows index smaller than src; take the matching corner of src
        src = src[tuple(slice(0, s) for s in index.shape)]
    for d in range(self.dim()):
        if d != dim and index.size(d) != self.size(d):
            raise NotImplementedError(f"{what}: index must span self outside dim {dim}")
    scatter(src, index, dim, out=self, reduce=reduce)


# --------------------------------------------------------------------------------------------------
# gathers
# --------------------------------------------------------------------------------------------------
def _elem_bytes(t, what):
    eb = t.element_size()
    if eb not in (1, 2, 4, 8) or t.is_complex():
        raise NotImplementedError(f"gnnops.{what}: element size {eb} is not supported (1/2/4/8-byte real types)")
    return eb


# index_select picks the push form by itself when it pays: the table is far larger than the 256 MiB
# Infinity Cache (so re-reads of a row go to HBM) and every row is selected several times on average.
_PUSH_MIN_TABLE_BYTES = 1 << 30
_PUSH_MIN_REUSE = 3


def index_select(input, dim, index, plan=None):
    """torch.index_select(input, dim, index).

    Pull form: one gathered row per output row (E*row random reads). Push form (a Plan of ``index`` over
    input.size(dim), given or built here when the heuristic above says so): every input row is read once
    and stored to all output rows that select it — N*row reads instead of E*row."""
    _require_gpu(input, index)
    _refuse_grad("index_select", input)
    if index.dtype == torch.int32:  # ATen takes IntTensor indices here; the kernels read int64
        index = index.long()
    _check_index(index, "index_select")
    if index.dim() > 1:
        raise IndexError("index_select(): Index is supposed to be a vector")
    dim = _norm_dim(dim, input.dim(), "index_select")
    eb = _elem_bytes(input, "index_select")
    input = input.contiguous()
    index = index.contiguous().view(-1)
    B, N, K = _bek(input.shape, dim)
    E = index.numel()
    shape = list(input.shape)
    shape[dim] = E
    out = torch.empty(shape, dtype=input.dtype, device=input.device)
    L = _lib.load()
    row_bytes = K * eb
    if (plan is None and row_bytes % 16 == 0 and N * row_bytes * B >= _PUSH_MIN_TABLE_BYTES
            and E >= _PUSH_MIN_REUSE * N and input.data_ptr() % 16 == 0):
        if B == 1 and not _plan_cache_enabled and 256 < N < 2 ** 31 and E < 2 ** 31:
            # push form with no plan to keep: partition by bucket, finish the sort on chip inside the copy (bucket.hip)
            ws_bytes = L.gnnops_bucket_workspace_bytes(E, N)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=input.device)
            with _on(input.device):
                check(L.gnnops_bucket_partition(index.data_ptr(), E, N, ws.data_ptr(), ws_bytes, _stream()), "bucket_partition")
                hub_bytes = L.gnnops_hub_workspace_bytes(E, 0, 0)   # hot rows are written by whole workgroups (hub.h)
                hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=input.device) if hub_bytes else None
                check(L.gnnops_bucket_select_hubs(input.data_ptr(), ws.data_ptr(), out.data_ptr(), N, K, E, eb,
                                                  hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes, _stream()),
                      "bucket_select")
            return out
        plan = get_plan(index, N)
    with _on(input.device):
        if plan is not None:
            if plan.E != E or plan.N != N:
                raise ValueError("index_select: plan does not match index / input.size(dim)")
            hub_bytes = L.gnnops_hub_workspace_bytes(E, 0, 0) if B == 1 else 0   # hot rows: hub.h
            hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=input.device) if hub_bytes else None
            rc = L.gnnops_index_select_planned_hubs(input.data_ptr(), plan.rowptr.data_ptr(), plan.perm.data_ptr(),
                                                    out.data_ptr(), B, N, K, E, eb,
                                                    hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes, _stream())
        else:
            rc = L.gnnops_index_select(input.data_ptr(), index.data_ptr(), out.data_ptr(), B, N, K, E, eb, _stream())
    check(rc, "index_select")
    return out


def gather(input, dim, index):
    """torch.gather(input, dim, index) for index spanning input outside ``dim``."""
    _require_gpu(input, index)
    _refuse_grad("gather", input)
    _check_index(index, "gather")
    if index.dim() != input.dim():
        raise RuntimeError("gather(): Index tensor must have the same number of dimensions as input tensor")
    dim = _norm_dim(dim, input.dim(), "gather")
    eb = _elem_bytes(input, "gather")
    for d in range(input.dim()):
        if d != dim and index.size(d) != input.size(d):
            raise NotImplementedError("gnnops.gather: index must span input outside `dim`")
    input = input.contiguous()
    index = index.contiguous()
    B, N, K = _bek(input.shape, dim)
    E = index.size(dim)
    out = torch.empty(index.shape, dtype=input.dtype, device=input.device)
    with _on(input.device):
        rc = _lib.load().gnnops_gather(input.data_ptr(), index.data_ptr(), out.data_ptr(), B, N, K, E, eb, _stream())
    check(rc, "gather")
    return out


def index_add_(self, dim, index, source, alpha=1):
    """Tensor.index_add_(dim, index, source): self.select(dim, index[j]) += source.select(dim, j). Returns self."""
    is_plan = isinstance(index, Plan)
    _require_gpu(self, source, None if is_plan else index)
    if alpha != 1:
        raise NotImplementedError("gnnops.index_add_: alpha != 1 is not supported")
    if self.dtype != source.dtype:
        raise RuntimeError("index_add_(): self and source must have the same dtype")
    if not self.is_contiguous():
        raise NotImplementedError("gnnops.index_add_: self must be contiguous")
    dim = _norm_dim(dim, self.dim(), "index_add_")
    if not is_plan:
        if index.dtype == torch.int32:  # ATen takes IntTensor indices here; the kernels read int64
            index = index.long()
        _check_index(index, "index_add_")
        if index.dim() != 1:
            raise IndexError("index_add_(): Index is supposed to be a vector")
        if index.numel() != source.size(dim):
            raise IndexError("index_add_(): Number of indices should be equal to source.size(dim)")
    scatter(source, index, dim, out=self, reduce="sum")
    return self


def index_add(input, dim, index, source):
    """torch.index_add (out of place): clone + index_add_ (benchmark_fused_index_add_reduce.py:13)."""
    return index_add_(input.clone(), dim, index, source)


def index_select_sum(input, dim, index):
    """fp32 value of ``torch.index_select(input, dim, index).sum()`` without materialising the gather
    (benchmark_fused_index_select_reduce.py:12-20). Returns a 0-dim float32 device tensor."""
    _require_gpu(input, index)
    _refuse_grad("index_select_sum", input)
    _check_index(index, "index_select_sum")
    dt = _dtype_code(input, "index_select_sum")
    dim = _norm_dim(dim, input.dim(), "index_select_sum")
    input = input.contiguous()
    index = index.contiguous().view(-1)
    B, N, K = _bek(input.shape, dim)
    L = _lib.load()
    out = torch.empty((), dtype=torch.float32, device=input.device)
    ws_bytes = L.gnnops_fused_select_sum_workspace_bytes()
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=input.device)
    with _on(input.device):
        rc = L.gnnops_fused_index_select_sum(input.data_ptr(), index.data_ptr(), out.data_ptr(), B, N, K,
                                             index.numel(), dt, ws.data_ptr(), ws_bytes, _stream())
    check(rc, "fused_index_select_sum")
    return out


def index_add_select_sum(input, dim, index, other):
    """fp32 value of ``torch.index_select(torch.index_add(input, dim, index, other), dim, index).sum(dim)``
    (benchmark_fused_index_add_reduce.py:12-20) in one pass; nothing of input's size is materialised."""
    is_plan = isinstance(index, Plan)
    _require_gpu(input, other, None if is_plan else index)
    _refuse_grad("index_add_select_sum", input, other)
    dt = _dtype_code(input, "index_add_select_sum")
    if other.dtype != input.dtype:
        raise RuntimeError("index_add_select_sum: input and other must have the same dtype")
    dim = _norm_dim(dim, input.dim(), "index_add_select_sum")
    input = input.contiguous()
    other = other.contiguous()
    B, N, K = _bek(input.shape, dim)
    Bo, E, Ko = _bek(other.shape, dim)
    if (Bo, Ko) != (B, K):
        raise RuntimeError("index_add_select_sum: other must match input outside `dim`")
    if not is_plan:
        _check_index(index, "index_add_select_sum")
        if index.numel() != E:
            raise IndexError("index_add_select_sum: index length must equal other.size(dim)")
    plan = get_plan(index if is_plan else index.contiguous().view(-1), N)
    shape = list(input.shape)
    del shape[dim]
    out = torch.empty(shape, dtype=torch.float32, device=input.device)
    L = _lib.load()
    ws_bytes = L.gnnops_fused_index_add_select_sum_workspace_bytes(B, K)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=input.device)
    with _on(input.device):
        rc = L.gnnops_fused_index_add_select_sum(input.data_ptr(), other.data_ptr(), plan.rowptr.data_ptr(),
                                                 plan.perm.data_ptr(), out.data_ptr(), B, N, E, K, dt, ws.data_ptr(),
                                                 ws_bytes, _stream())
    check(rc, "fused_index_add_select_sum")
    return out


def addmm(input, mat1, mat2, *, beta=1, alpha=1):
    """torch.addmm(input, mat1, mat2) for float16 / bfloat16 / float32 matrices (benchmark_native_addmm.py:13-16;
    fp32 is the dtype of the older data/native_addmm.csv sweep)."""
    if beta != 1 or alpha != 1:
        raise NotImplementedError("gnnops.addmm: beta and alpha must be 1")
    _require_gpu(input, mat1, mat2)
    _refuse_grad("addmm", input, mat1, mat2)
    if mat1.dim() != 2 or mat2.dim() != 2:
        raise RuntimeError("addmm: mat1 and mat2 must be matrices")
    if mat1.size(1) != mat2.size(0):
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({mat1.size(0)}x{mat1.size(1)} and "
                           f"{mat2.size(0)}x{mat2.size(1)})")
    if mat1.dtype != mat2.dtype or (input is not None and input.dtype != mat1.dtype):
        raise RuntimeError("addmm: operands must have the same dtype")
    if mat1.dtype not in (torch.float16, torch.bfloat16, torch.float32):
        raise NotImplementedError(f"gnnops.addmm: dtype {mat1.dtype} is not supported (float16/bfloat16/float32)")
    dt = _DT[mat1.dtype]
    M, K = mat1.shape
    N = mat2.size(1)
    ld_input = N
    if input is not None:
        if input.numel() == N and (input.dim() == 1 or input.size(0) == 1) and M > 1:
            input, ld_input = input.contiguous(), 0          # one row for every output row (a Linear bias): never expanded
        else:
            input = input.expand(M, N).contiguous()
    mat1 = mat1.contiguous()
    mat2 = mat2.contiguous()
    out = torch.empty((M, N), dtype=mat1.dtype, device=mat1.device)
    L = _lib.load()
    ws_bytes = L.gnnops_addmm_workspace_bytes(M, N, K) if mat1.dtype != torch.float32 else 0   # fp32 kernels use none
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=mat1.device) if ws_bytes else None
    with _on(mat1.device):
        rc = L.gnnops_addmm_ld(input.data_ptr() if input is not None else None, ld_input, mat1.data_ptr(), mat2.data_ptr(),
                               out.data_ptr(), M, N, K, dt, ws.data_ptr() if ws is not None else None, ws_bytes, _stream())
    check(rc, "addmm")
    return out


def matmul(input, other):
    """torch.matmul(input, other) for 2-D float16 / bfloat16 matrices (benchmark_native_matmul.py:13-16)."""
    return addmm(None, input, other)
