"""Differentiable wrappers (torch.autograd.Function) around the forward kernels, so the shim `torch_scatter`
ops and index_select / gather can sit inside a training graph like the upstream extension's do
(prerequisite for OpProfiler's training loop, graph_benchmark/profile/OpProfiler.py:259-292).

Backward formulas are the upstream ones:
  scatter sum   grad_src = grad_out gathered at index
  scatter mean  grad_src = (grad_out / max(count,1)) gathered at index
  scatter min/max  grad_src = grad_out routed to the arg positions only
  index_select  grad_input = index_add_ of grad_out
  gather        grad_input = scatter_add_ of grad_out
Every backward step is one of our own forward ops; nothing falls back to stock kernels.
"""
import torch

from . import ops


def _gather_back(grad_out, index, dim, src_shape):
    """grad_out[.., index, ..] laid out like src: a row index uses index_select, a full index uses gather."""
    if index.dim() == 1:
        return ops.index_select(grad_out, dim, index)
    return ops.gather(grad_out, dim, index.expand(src_shape) if index.shape != src_shape else index)


class _ScatterSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, dim, dim_size):
        out = ops.scatter(src, index, dim, None, dim_size, "sum")
        ctx.save_for_backward(index)
        ctx.dim, ctx.src_shape = dim % src.dim(), src.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (index,) = ctx.saved_tensors
        return _gather_back(grad_out.contiguous(), index, ctx.dim, ctx.src_shape), None, None, None


class _ScatterMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, dim, dim_size):
        dim = dim % src.dim()
        out = ops.scatter(src, index, dim, None, dim_size, "mean")
        ones = torch.ones((src.size(dim),) if index.dim() == 1 else src.shape, dtype=src.dtype, device=src.device)
        count = ops.scatter(ones, index, 0 if index.dim() == 1 else dim, None, out.size(dim), "sum").clamp_(min=1)
        if index.dim() == 1:
            shape = [1] * src.dim()
            shape[dim] = -1
            count = count.view(shape)
        ctx.save_for_backward(index, count)
        ctx.dim, ctx.src_shape = dim, src.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        index, count = ctx.saved_tensors
        return _gather_back((grad_out / count).contiguous(), index, ctx.dim, ctx.src_shape), None, None, None


class _ScatterMinMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, dim, dim_size, reduce):
        dim = dim % src.dim()
        out, arg = ops.scatter(src, index, dim, None, dim_size, reduce)
        ctx.save_for_backward(arg)
        ctx.dim, ctx.src_shape = dim, src.shape
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        shape = list(ctx.src_shape)
        shape[ctx.dim] += 1  # slot E swallows the groups nothing reached (arg == E), as upstream does
        grad_src = torch.zeros(shape, dtype=grad_out.dtype, device=grad_out.device)
        ops.scatter_add_(grad_src, ctx.dim, arg, grad_out.contiguous())
        return grad_src.narrow(ctx.dim, 0, shape[ctx.dim] - 1), None, None, None, None


class _IndexSelect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, dim, index, plan):
        ctx.save_for_backward(index)
        ctx.dim, ctx.in_shape, ctx.plan = dim % input.dim(), input.shape, plan
        return ops.index_select(input, dim, index, plan=plan)

    @staticmethod
    def backward(ctx, grad_out):
        (index,) = ctx.saved_tensors
        grad_in = torch.zeros(ctx.in_shape, dtype=grad_out.dtype, device=grad_out.device)
        # the plan of `index` over input.size(dim) is exactly the plan the scatter-add back needs
        ops.index_add_(grad_in, ctx.dim, ctx.plan if ctx.plan is not None else index, grad_out.contiguous())
        return grad_in, None, None, None


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, dim, index):
        ctx.save_for_backward(index)
        ctx.dim, ctx.in_shape = dim % input.dim(), input.shape
        return ops.gather(input, dim, index)

    @staticmethod
    def backward(ctx, grad_out):
        (index,) = ctx.saved_tensors
        grad_in = torch.zeros(ctx.in_shape, dtype=grad_out.dtype, device=grad_out.device)
        ops.scatter_add_(grad_in, ctx.dim, index, grad_out.contiguous())
        return grad_in, None, None


class _Composite(torch.autograd.Function):
    """scatter_softmax / scatter_log_softmax / scatter_logsumexp over a row index along `dim`, differentiable in src.
      softmax      dx = y * (g - sum_group(g * y))
      log_softmax  dx = g - exp(y) * sum_group(g)
      logsumexp    dx = g[group] * exp(x - out[group])
      std          dx = g[group] * (x - mean[group]) / ((c' + 1e-6) * out[group]),  c' = max(cnt - ddof, 1); 0 where out == 0
    The group sums and the broadcasts back are our segment-reduce and index_select kernels over the same plan."""

    @staticmethod
    def forward(ctx, src, index, dim, dim_size, mode, eps):
        from . import segment

        dim = dim % src.dim()
        row = ops._row_index_of(index, src, dim)
        if row is None:
            raise NotImplementedError("gnnops: composite ops take a row index")
        N = int(dim_size) if dim_size is not None else (ops.index_max(row) + 1 if row.numel() else 0)
        plan = ops.get_plan(row, N)
        out = segment._composite(src, plan, dim, N, mode, eps)
        ctx.save_for_backward(src, out, row)
        ctx.plan, ctx.dim, ctx.mode, ctx.param = plan, dim, mode, eps
        return out

    @staticmethod
    def backward(ctx, g):
        src, out, row = ctx.saved_tensors
        plan, dim, mode = ctx.plan, ctx.dim, ctx.mode
        g = g.contiguous()
        if mode == "softmax":
            s = ops.scatter(g * out, plan, dim, None, None, "sum")
            dx = out * (g - ops.index_select(s, dim, row))
        elif mode == "log_softmax":
            s = ops.scatter(g, plan, dim, None, None, "sum")
            dx = g - out.exp() * ops.index_select(s, dim, row)
        elif mode == "std":  # out has the group shape; eps carries the unbiased flag (composite.hip: param != 0)
            mean = ops.scatter(src, plan, dim, None, None, "mean")
            cnt = (plan.rowptr[1:] - plan.rowptr[:-1]).to(torch.float32)
            shape = [1] * src.dim()
            shape[dim] = -1
            denom = ((cnt - (1.0 if ctx.param != 0 else 0.0)).clamp_(min=1.0) + 1e-6).view(shape) * out.float()
            scale = torch.where(out != 0, g.float() / denom, torch.zeros_like(denom)).to(src.dtype)
            dx = ops.index_select(scale, dim, row) * (src - ops.index_select(mean, dim, row))
        else:  # logsumexp: out has the group shape
            dx = ops.index_select(g, dim, row) * (src - ops.index_select(out, dim, row)).exp()
        return dx, None, None, None, None, None


class _SegmentCSR(torch.autograd.Function):
    """segment_csr / segment_coo over dim 0. Backward = the gather of the group's gradient back to its members:
      sum   dx[e] = g[seg(e)]           mean  dx[e] = g[seg(e)] / max(count, 1)
      min / max   dx = g routed to the arg positions only
    seg(e) comes from gnnops_rowptr_expand (CSR) or is the sorted index itself (COO); positions no segment holds get 0."""

    @staticmethod
    def forward(ctx, src, indptr, index, dim_size, reduce):
        from . import segment

        if index is not None:
            res = segment._segment_coo_raw(src, index, dim_size, reduce)
        else:
            res = segment._segment_csr_raw(src, indptr, reduce)
        ctx.reduce, ctx.E, ctx.src_shape = reduce, src.size(0), src.shape
        if reduce in ("min", "max"):
            out, arg = res
            ctx.save_for_backward(arg)
            ctx.mark_non_differentiable(arg)
            return out, arg
        N = res.size(0)
        seg = index if index is not None else segment.expand_rowptr(indptr, src.size(0))
        if reduce == "mean":
            if index is not None:
                cnt = ops.scatter(torch.ones(src.size(0), dtype=src.dtype, device=src.device), index, 0, None, N, "sum")
            else:
                cnt = (indptr[1:] - indptr[:-1]).to(src.dtype)
            ctx.save_for_backward(seg, cnt.clamp_(min=1))
        else:
            ctx.save_for_backward(seg)
        return res

    @staticmethod
    def backward(ctx, g, *_):
        g = g.contiguous()
        if ctx.reduce in ("min", "max"):
            (arg,) = ctx.saved_tensors
            shape = list(ctx.src_shape)
            shape[0] += 1            # slot E swallows the groups nothing reached (arg == E)
            dx = torch.zeros(shape, dtype=g.dtype, device=g.device)
            ops.scatter_add_(dx, 0, arg, g)
            return dx[: ctx.E], None, None, None, None
        if ctx.reduce == "mean":
            seg, cnt = ctx.saved_tensors
            g = g / cnt.view([-1] + [1] * (g.dim() - 1))
        else:
            (seg,) = ctx.saved_tensors
        g_ext = torch.cat([g, g.new_zeros((1,) + tuple(g.shape[1:]))])     # row N: positions outside every segment
        return ops.index_select(g_ext, 0, seg), None, None, None, None


def segment_csr(src, indptr, reduce="sum"):
    if reduce not in ("sum", "add", "mean", "min", "max"):
        raise NotImplementedError(f"gnnops: backward of segment_csr(reduce={reduce!r}) is not implemented")
    return _SegmentCSR.apply(src, indptr, None, None, "sum" if reduce == "add" else reduce)


def segment_coo(src, index, dim_size=None, reduce="sum"):
    if reduce not in ("sum", "add", "mean", "min", "max"):
        raise NotImplementedError(f"gnnops: backward of segment_coo(reduce={reduce!r}) is not implemented")
    return _SegmentCSR.apply(src, None, index, dim_size, "sum" if reduce == "add" else reduce)


class _GatherCSR(torch.autograd.Function):
    """gather_csr / gather_coo: out[e] = src[seg(e)]; backward = the segment sum of the gradient."""

    @staticmethod
    def forward(ctx, src, indptr, index, E):
        from . import segment

        ctx.N = src.size(0)
        if index is None:
            index = segment.expand_rowptr(indptr, E)
            ctx.csr = True
            ext = torch.cat([src, src.new_zeros((1,) + tuple(src.shape[1:]))])   # row N: positions below indptr[0]
            ctx.save_for_backward(indptr)
            return ops.index_select(ext, 0, index)
        ctx.csr = False
        ctx.save_for_backward(index)
        return ops.index_select(src, 0, index)

    @staticmethod
    def backward(ctx, g):
        from . import segment

        (ix,) = ctx.saved_tensors
        g = g.contiguous()
        if ctx.csr:
            return segment._segment_csr_raw(g, ix, "sum"), None, None, None
        dx = torch.zeros((ctx.N,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        return ops.index_add_(dx, 0, ix, g), None, None, None     # gather_coo takes any index: a scatter-add back


def gather_csr(src, indptr, E):
    return _GatherCSR.apply(src, indptr, None, E)


def gather_coo(src, index):
    return _GatherCSR.apply(src, None, index, index.numel())


class _SpMM(torch.autograd.Function):
    """torch_sparse.spmm (transposed=False: out rows follow index[0]) and its transposed form spmm_t (out rows follow
    index[1]). With A the sparse operand:  d matrix = A^T @ g  — the OTHER form over the same index, i.e. one more
    launch of the same row-split kernel —  and  d value[k] = <g[out_row[k]], matrix[gathered_row[k]]>  (gnnops_sddmm)."""

    @staticmethod
    def forward(ctx, index, value, m, n, matrix, transposed):
        from . import sparse

        ctx.m, ctx.n, ctx.transposed = m, n, transposed
        ctx.save_for_backward(index, value, matrix)
        fn = sparse._spmm_t_raw if transposed else sparse._spmm_raw
        return fn(index, value, m, n, matrix)

    @staticmethod
    def backward(ctx, g):
        from . import sparse

        index, value, matrix = ctx.saved_tensors
        g = g.contiguous()
        d_value = d_matrix = None
        if ctx.needs_input_grad[4]:
            other = sparse._spmm_raw if ctx.transposed else sparse._spmm_t_raw
            d_matrix = other(index, value, ctx.m, ctx.n, g)
        if value is not None and ctx.needs_input_grad[1]:
            out_rows, in_rows = (index[1], index[0]) if ctx.transposed else (index[0], index[1])
            d_value = sparse.sddmm(out_rows, in_rows, g, matrix)
        return None, d_value, None, None, d_matrix, None


def spmm(index, value, m, n, matrix):
    return _SpMM.apply(index, value, m, n, matrix, False)


def spmm_t(index, value, m, n, matrix):
    return _SpMM.apply(index, value, m, n, matrix, True)


def composite(src, index, dim, dim_size, mode, eps):
    """Differentiable entry for the composite ops (used by gnnops.segment when src requires grad)."""
    return _Composite.apply(src, index, dim, dim_size, mode, eps)


def _needs_grad(t):
    return torch.is_grad_enabled() and t.requires_grad


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    """torch_scatter.scatter with autograd for sum/add/mean/min/max when src requires grad. The forms with no backward
    (``out=`` given, a ``Plan`` as index, reduce='mul') raise instead of returning a tensor cut off from the graph."""
    if _needs_grad(src) and (out is not None or isinstance(index, ops.Plan)):
        raise NotImplementedError("gnnops.scatter: src requires grad, but the out= / Plan-index forms have no backward; "
                                  "pass the index tensor and no out=, or detach src")
    if out is None and _needs_grad(src) and not isinstance(index, ops.Plan):
        if reduce in ("sum", "add"):
            return _ScatterSum.apply(src, index, dim, dim_size)
        if reduce == "mean":
            return _ScatterMean.apply(src, index, dim, dim_size)
        if reduce in ("min", "max"):
            return _ScatterMinMax.apply(src, index, dim, dim_size, reduce)
        raise NotImplementedError(f"gnnops: backward of scatter(reduce={reduce!r}) is not implemented")
    return ops.scatter(src, index, dim, out, dim_size, reduce)


def index_select(input, dim, index, plan=None):
    if _needs_grad(input):
        return _IndexSelect.apply(input, dim, index, plan)
    return ops.index_select(input, dim, index, plan=plan)


def gather(input, dim, index):
    return _Gather.apply(input, dim, index) if _needs_grad(input) else ops.gather(input, dim, index)


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


class _AddMM(torch.autograd.Function):
    """addmm / matmul on the MFMA kernels: d input = g (summed over broadcast dims), d mat1 = g @ mat2^T,
    d mat2 = mat1^T @ g; the transposes are our tiled transpose copy, the products the same GEMM kernel."""

    @staticmethod
    def forward(ctx, input, mat1, mat2):
        ctx.save_for_backward(mat1, mat2)
        ctx.in_shape = None if input is None else input.shape
        return ops.addmm(input, mat1, mat2)

    @staticmethod
    def backward(ctx, g):
        from .sparse import transpose_contiguous

        mat1, mat2 = ctx.saved_tensors
        g = g.contiguous()
        d_in = d1 = d2 = None
        if ctx.in_shape is not None and ctx.needs_input_grad[0]:
            d_in = g
            if tuple(ctx.in_shape) != tuple(g.shape):       # input was broadcast to [M, N]
                d_in = g.sum_to_size(ctx.in_shape)
        if ctx.needs_input_grad[1]:
            d1 = ops.matmul(g, transpose_contiguous(mat2))
        if ctx.needs_input_grad[2]:
            d2 = ops.matmul(transpose_contiguous(mat1), g)
        return d_in, d1, d2


def addmm(input, mat1, mat2, *, beta=1, alpha=1):
    if beta != 1 or alpha != 1:
        raise NotImplementedError("gnnops.addmm: beta and alpha must be 1")
    if _needs_grad(mat1) or _needs_grad(mat2) or (input is not None and _needs_grad(input)):
        return _AddMM.apply(input, mat1, mat2)
    return ops.addmm(input, mat1, mat2)


def matmul(input, other):
    return addmm(None, input, other)
