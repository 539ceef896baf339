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
    def forward(ctx, input, dim, index):
        ctx.save_for_backward(index)
        ctx.dim, ctx.in_shape = dim % input.dim(), input.shape
        return ops.index_select(input, dim, index)

    @staticmethod
    def backward(ctx, grad_out):
        (index,) = ctx.saved_tensors
        grad_in = torch.zeros(ctx.in_shape, dtype=grad_out.dtype, device=grad_out.device)
        ops.index_add_(grad_in, ctx.dim, index, grad_out.contiguous())
        return grad_in, None, None


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


def composite(src, index, dim, dim_size, mode, eps):
    """Differentiable entry for the composite ops (used by gnnops.segment when src requires grad)."""
    return _Composite.apply(src, index, dim, dim_size, mode, eps)


def _needs_grad(t):
    return torch.is_grad_enabled() and t.requires_grad


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    """torch_scatter.scatter with autograd for sum/add/mean/min/max when src requires grad (out= not given)."""
    if out is None and _needs_grad(src) and not isinstance(index, ops.Plan):
        if reduce in ("sum", "add"):
            return _ScatterSum.apply(src, index, dim, dim_size)
        if reduce == "mean":
            return _ScatterMean.apply(src, index, dim, dim_size)
        if reduce in ("min", "max"):
            return _ScatterMinMax.apply(src, index, dim, dim_size, reduce)
        raise NotImplementedError(f"gnnops: backward of scatter(reduce={reduce!r}) is not implemented")
    return ops.scatter(src, index, dim, out, dim_size, reduce)


def index_select(input, dim, index):
    return _IndexSelect.apply(input, dim, index) if _needs_grad(input) else ops.index_select(input, dim, index)


def gather(input, dim, index):
    return _Gather.apply(input, dim, index) if _needs_grad(input) else ops.gather(input, dim, index)
