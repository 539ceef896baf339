"""Single message-passing layers, forward pass: the five layers the reference's app benchmarks time
(app_bm/benchmark_convs.py:146-246: FiLMConv, GINConv, CGConv, PNAConv, SAGEConv; app_bm/groq_script.py:15-111 holds the
text of CGConv itself) — SURVEY.md §8(f) rank 4.

Constructor arguments, parameter names and shapes follow torch_geometric.nn.conv (2.0.2, requirements.txt:211), so a
state_dict moves between the two; the forward is NOT MessagePassing.propagate. A layer here is
    one dense product  x @ [all the layer's per-node weight blocks]      (gemm.hip, MFMA)
    one edge pass      gnnops_edge_reduce (csrc/conv.hip): message + aggregation(s) + degree scalers + residual
    (PNA / GIN / SAGE) one dense product on the aggregate                (gemm.hip)
because every Linear a message applies to cat([x_i, x_j, e]) splits into per-node products: z W = x_i W_i + x_j W_j + e W_e.
The [E, .] tensors of propagate (x_i, x_j, z, the messages) never exist.

Training (the reference's OpProfiler.py:259-292 profiles a train loop): GINConv, SAGEConv, CGConv and FiLMConv are
differentiable — the dense products through gnnops.autograd.addmm, the edge pass through `_EdgeReduce` below, whose backward
is the forward's own machinery run the other way: the output gradient gathered along the TRANSPOSED plan (copy messages), or
one streaming kernel that writes the per-edge gradient of the message (gnnops_edge_grad: cgconv / film) followed by two
segment sums over the plans the forward already holds (by destination for the p side, by source for the q side). Packed
weight operands are cached only while nothing requires grad. PNAConv (min / max / std aggregators with degree scalers): the
fused multi-aggregator pass has no backward, so in a graph that needs gradients the layer runs the propagate-order chain
(`_forward_train`: per-edge messages, one differentiable scatter per aggregator) on this package's differentiable ops.
"""
import ctypes

import torch

from . import _lib, ops
from ._lib import check
from .ops import _dtype_code, _on, _require_gpu, _stream, get_plan
from .sparse import _coo_rows_cols, _csr_arrays

FUNCTORS = {"copy": 0, "add": 1, "cgconv": 2, "film": 3}
_PARTS = {"copy": (1, 0, 0), "add": (1, 1, 1), "cgconv": (2, 2, 2), "film": (1, 2, 0)}   # K-wide parts per row of q, p, w
AGGREGATORS = {"sum": 0, "add": 0, "mean": 1, "min": 2, "max": 3, "std": 4}
SCALERS = {"identity": 0, "amplification": 1, "attenuation": 2, "linear": 3, "inverse_linear": 4}


def _rows(t, what, parts, K):
    """A [rows, >= parts*K] operand whose rows are contiguous runs (column blocks of a wider matrix are fine)."""
    if t is None:
        return None, 0
    if t.dim() != 2 or (t.size(1) > 1 and t.stride(1) != 1) or t.size(1) < parts * K:
        raise RuntimeError(f"edge_reduce: {what} must be 2-D with unit column stride and at least {parts * K} columns")
    return t, (t.stride(0) if t.size(0) > 1 else t.size(1))


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def edge_reduce(functor, q, edge_index, num_dst, p=None, w=None, add=None, aggr=("sum",), scalers=(), avg_deg=None, out=None,
                flip=False):
    """out[i] = [scaler_s(deg_i) * AGGR_a_{(j -> i) in edge_index} f(p[i], q[j], w[e])  for s in scalers for a in aggr].

    edge_index int64 [2, E] = (source j, destination i), PyG's flow="source_to_target" (``flip=True``: the transposed graph,
    row 0 = destinations — what the backward of a copy message runs). See include/gnnops.h (gnnops_edge_reduce) for the
    functors. ``out`` may be a column block of a wider buffer (what the layer would cat into). The destination plan
    (rowptr, perm) and the plan-ordered source ids are cached under the edge_index tensor. Differentiable in q, p, w and
    add for one sum / mean aggregator without scalers and ``out=None`` (copy / cgconv / film messages)."""
    if _wants_grad(q, p, w, add):
        if len(aggr) != 1 or aggr[0] not in ("sum", "add", "mean") or scalers or out is not None or flip or functor == "add":
            raise NotImplementedError("gnnops.conv.edge_reduce: an operand requires grad, but only one sum / mean aggregator "
                                      "without scalers of a copy / cgconv / film message has a backward (PNAConv is forward-only)")
        return _EdgeReduce.apply(functor, "mean" if aggr[0] == "mean" else "sum", edge_index, num_dst, q, p, w, add)
    _require_gpu(q, edge_index, p, w, add, out)
    edge_index, src_rows, dst_rows = _coo_rows_cols(edge_index, "edge_reduce")
    if flip:
        src_rows, dst_rows = dst_rows, src_rows
    nq, np_, nw = _PARTS[functor]
    if q.dim() != 2 or q.size(1) % nq:
        raise RuntimeError(f"edge_reduce: q must be [rows, {nq} * K]")
    K = q.size(1) // nq
    dt = _dtype_code(q, "edge_reduce")
    for t in (p, w, add, out):
        if t is not None and t.dtype != q.dtype:
            raise RuntimeError("edge_reduce: operands must have the same dtype")
    if functor != "copy" and p is None:
        raise RuntimeError(f"edge_reduce: the {functor} message needs the per-destination rows p")
    q, ldq = _rows(q, "q", nq, K)
    p, ldp = _rows(p, "p", np_, K)
    w, ldw = _rows(w, "w", nw, K)
    add, ldadd = _rows(add, "add", 1, K)
    E = edge_index.size(1)
    if p is not None and p.size(0) != num_dst or add is not None and add.size(0) != num_dst:
        raise RuntimeError("edge_reduce: p and add have one row per destination")
    if w is not None and w.size(0) != E:
        raise RuntimeError("edge_reduce: w has one row per edge")
    aggr_ids = [AGGREGATORS[a] for a in aggr]
    scal_ids = [SCALERS[s] for s in scalers]
    width = max(len(scal_ids), 1) * len(aggr_ids) * K
    if out is None:
        out = torch.empty((num_dst, width), dtype=q.dtype, device=q.device)
    elif out.size(0) != num_dst:
        raise RuntimeError("edge_reduce: out has one row per destination")
    out, ldo = _rows(out, "out", 1, width)
    avg_log, avg_lin = (1.0, 1.0) if avg_deg is None else (float(avg_deg["log"]), float(avg_deg["lin"]))
    # plans are cached under the [2, E] tensor: tag 1 = plan of row 1 (destinations), tag 0 = plan of row 0 (the flipped graph)
    plan = get_plan(dst_rows, num_dst, owner=edge_index, tag=0 if flip else 1, companion=src_rows)   # small graphs: col comes with the plan
    if plan.col is not None or E == 0:
        col = plan.col if E else src_rows
    else:
        col, _ = _csr_arrays(plan, src_rows, None, owner=edge_index, tag=1 if flip else 0)
    c_aggr = (ctypes.c_int * len(aggr_ids))(*aggr_ids)
    c_scal = (ctypes.c_int * max(len(scal_ids), 1))(*scal_ids)
    ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    L = _lib.load()
    hub_bytes = L.gnnops_edge_reduce_hub_workspace_bytes(E, K)   # destinations with more than 8192 edges: reduced piecewise
    hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=q.device) if hub_bytes else None
    with _on(q.device):
        rc = L.gnnops_edge_reduce_hubs(FUNCTORS[functor], ptr(q), ldq, ptr(p), ldp, ptr(w), ldw, ptr(add), ldadd,
                                       plan.rowptr.data_ptr(), plan.perm.data_ptr(), col.data_ptr(), out.data_ptr(), ldo,
                                       num_dst, E, K, c_aggr, len(aggr_ids), c_scal, len(scal_ids), avg_log, avg_lin, dt,
                                       ptr(hub_ws), hub_bytes, _stream())
    check(rc, "edge_reduce")
    return out


class _EdgeReduce(torch.autograd.Function):
    """One sum / mean edge pass, differentiable in q [N_src, .], p [N_dst, .], w [E, .] and add [N_dst, K].

    backward, with g = grad_out (for mean: divided by max(deg, 1) per destination):
      copy    d q[j] = sum over the edges OUT of j of g[i]        = the same edge pass over the transposed graph
      cgconv  gz[e] = g[i] * d message / d z (gnnops_edge_grad), z = p[i] + q[j] + w[e]:
              d p = segment sum of gz by destination, d q = segment sum by source, d w = gz
      film    gp[e], gq[e] from the same kernel; d p = sum of gp by destination, d q = sum of gq by source
      d add = grad_out."""

    @staticmethod
    def forward(ctx, functor, aggr, edge_index, num_dst, q, p, w, add):
        out = edge_reduce(functor, q, edge_index, num_dst, p=p, w=w, add=add, aggr=(aggr,))
        ctx.functor, ctx.aggr, ctx.num_dst, ctx.n_src = functor, aggr, num_dst, q.size(0)
        ctx.has = (p is not None, w is not None, add is not None)
        ctx.save_for_backward(edge_index, q, *(t for t in (p, w) if t is not None))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        edge_index, q, *rest = ctx.saved_tensors
        has_p, has_w, has_add = ctx.has
        p = rest.pop(0) if has_p else None
        w = rest.pop(0) if has_w else None
        need_q, need_p, need_w, need_add = ctx.needs_input_grad[4:8]
        g = grad_out.contiguous()
        d_add = g if (has_add and need_add) else None
        index, src_rows, dst_rows = _coo_rows_cols(edge_index, "edge_reduce")
        E = index.size(1)
        plan_dst = get_plan(dst_rows, ctx.num_dst, owner=edge_index, tag=1, companion=src_rows)
        if ctx.aggr == "mean":
            deg = (plan_dst.rowptr[1:] - plan_dst.rowptr[:-1]).clamp(min=1).to(g.dtype)
            g = g / deg.unsqueeze(1)
        d_q = d_p = d_w = None
        K = g.size(1)
        if ctx.functor == "copy":
            if need_q:
                d_q = edge_reduce("copy", g, edge_index, ctx.n_src, flip=True)
            return None, None, None, None, d_q, None, None, d_add
        if E == 0:
            d_q = torch.zeros_like(q) if need_q else None
            d_p = torch.zeros_like(p) if need_p else None
            d_w = torch.zeros_like(w) if (has_w and need_w) else None
            return None, None, None, None, d_q, d_p, d_w, d_add
        q_, ldq = _rows(q, "q", _PARTS[ctx.functor][0], K)
        p_, ldp = _rows(p, "p", 2, K)
        w_, ldw = _rows(w, "w", 2, K) if has_w else (None, 0)
        gp = torch.empty((E, 2 * K), dtype=g.dtype, device=g.device)
        gq = torch.empty((E, K), dtype=g.dtype, device=g.device) if ctx.functor == "film" else None
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        with _on(g.device):
            check(_lib.load().gnnops_edge_grad(FUNCTORS[ctx.functor], ptr(p_), ldp, ptr(q_), ldq, ptr(w_), ldw, g.data_ptr(), g.stride(0),
                                               src_rows.data_ptr(), dst_rows.data_ptr(), gp.data_ptr(), ptr(gq), E, K,
                                               _dtype_code(g, "edge_grad"), _stream()), "edge_grad")
        if need_p:
            d_p = ops.scatter(gp, plan_dst, 0, None, None, "sum")
            if p.size(1) != 2 * K:       # p was a column block wider than its 2K parts (never the case for the layers here)
                d_p = torch.nn.functional.pad(d_p, (0, p.size(1) - 2 * K))
        if need_q:
            plan_src = get_plan(src_rows, ctx.n_src, owner=edge_index, tag=0)
            d_q = ops.scatter(gq if gq is not None else gp, plan_src, 0, None, None, "sum")
        if has_w and need_w:
            d_w = gp
        return None, None, None, None, d_q, d_p, d_w, d_add


# ---- dense side: every weight block a layer applies per node, as ONE operand of the MFMA product ------------------------
class _Packed:
    """Weights of several nn.Linear maps packed for one product, rebuilt only when a parameter changes (identity + version
    counter of the parameters named in ``params``), so transposes and concatenation are paid once per set of weights.

    side by side (default): x @ [W_0^T | W_1^T | ...] — one input, several maps; bias = the concatenated bias rows
    stacked  (stack=True) : [h_0 | h_1 | ...] @ [W_0^T ; W_1^T ; ...] — several inputs summed into one output"""

    def __init__(self):
        self.key = self.weight = self.bias = self._alive = None

    def get(self, params, blocks, stack=False):
        """params: the Parameters the blocks are cut from; blocks: [(weight or a column slice of it [out, in], bias or None)]."""
        # Module.to() / .half() swap a parameter's data without touching its version counter: the pointer and dtype are in the key
        # tensors created under torch.inference_mode() have no version counter: nothing derived from them is cached
        if _wants_grad(*params):     # training: the packed operand is part of the graph (cat / t are differentiable) and never cached
            return self._pack(blocks, stack)
        cacheable = all(ops._version_of(t) is not None for t in params if t is not None)
        key = tuple((id(t), t._version, t.data_ptr(), t.dtype) for t in params if t is not None) if cacheable else None
        if key is None or key != self.key:
            with torch.no_grad():
                self.weight, self.bias = self._pack(blocks, stack)
            self.key = key
            self._alive = list(params)   # the key holds ids: keep the tensors they name alive so that no id is handed out again
        return self.weight, self.bias

    @staticmethod
    def _pack(blocks, stack):
        if stack:
            weight = torch.cat([wt.t() for wt, _ in blocks], dim=0).contiguous()
            biases = [b for _, b in blocks if b is not None]
            bias = sum(biases[1:], biases[0]).contiguous() if biases else None
        else:
            weight = torch.cat([wt.t() for wt, _ in blocks], dim=1).contiguous()
            if any(b is not None for _, b in blocks):
                bias = torch.cat([b if b is not None else wt.new_zeros(wt.size(0)) for wt, b in blocks]).contiguous()
            else:
                bias = None
        return weight, bias


def _dense(x, packed):
    """x [N, D_in] @ weight [D_in, W] (+ bias row) on the MFMA kernels (gemm.hip)."""
    from . import autograd

    weight, bias = packed
    return autograd.addmm(bias, x, weight) if bias is not None else autograd.matmul(x, weight)


def _linear(x, lin, cache):
    return _dense(x, cache.get([lin.weight, lin.bias], [(lin.weight, lin.bias)]))


def _pair(x):
    return x if isinstance(x, (tuple, list)) else (x, x)


class _Layer(torch.nn.Module):
    """Every layer is trainable (module docstring); `_freeze` / `_forward_only` remain for callers that want an inference-only copy."""

    def _freeze(self):
        self.requires_grad_(False)

    def _forward_only(self, *tensors):
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                        any(t is not None and t.requires_grad for t in tensors)):
            raise RuntimeError(f"gnnops.conv.{type(self).__name__} is forward-only: freeze its parameters / call it under "
                               "torch.no_grad() (trainable: GINConv, SAGEConv, CGConv, FiLMConv)")


class GINConv(_Layer):
    """x'_i = nn((1 + eps) * x_i + sum_j x_j)  (torch_geometric GINConv; benchmark_convs.py:163 GINConv(Linear(11, 2048)))."""

    def __init__(self, nn, eps=0.0, train_eps=False):
        super().__init__()
        self.nn = nn
        self.initial_eps = eps
        if train_eps:
            self.eps = torch.nn.Parameter(torch.tensor([float(eps)]))
        else:
            self.register_buffer("eps", torch.tensor([float(eps)]))
        self._packed = _Packed()
        self._eps_key, self._eps_host = None, float(eps)

    def _eps_value(self):
        """eps as a host number, read back from the device only when the tensor changed (a read-back per forward would
        synchronise every call and cannot be captured into a graph)."""
        t = self.eps
        key = (ops._version_of(t), t.data_ptr())
        if key[0] is None or key != self._eps_key:
            self._eps_host = float(t)
            self._eps_key = key
        return self._eps_host

    def forward(self, x, edge_index, size=None):
        x_src, x_dst = _pair(x)
        n_dst = x_dst.size(0) if size is None else size[1]
        if _wants_grad(self.eps):       # train_eps: eps is part of the graph
            root = x_dst * (1.0 + self.eps)
        else:
            eps = self._eps_value()
            root = x_dst if eps == 0.0 else x_dst * (1.0 + eps)
        h = edge_reduce("copy", x_src.contiguous(), edge_index, n_dst, add=root.contiguous())
        return _linear(h, self.nn, self._packed) if isinstance(self.nn, torch.nn.Linear) else self.nn(h)


class SAGEConv(_Layer):
    """x'_i = W_l mean_j x_j + W_r x_i  (torch_geometric SAGEConv; benchmark_convs.py:231 SAGEConv(-1, 2048)).
    One edge pass writes the mean into the left block of [mean | x]; one product with [W_l^T ; W_r^T] finishes the layer."""

    def __init__(self, in_channels, out_channels, normalize=False, root_weight=True, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.normalize, self.root_weight, self._bias = normalize, root_weight, bias
        self.lin_l = self.lin_r = None
        if not (isinstance(in_channels, int) and in_channels < 0):
            self._build(*((in_channels, in_channels) if isinstance(in_channels, int) else in_channels))
        self._packed = _Packed()

    def _build(self, in_src, in_dst, like=None):
        kw = {} if like is None else {"device": like.device, "dtype": like.dtype}
        self.lin_l = torch.nn.Linear(in_src, self.out_channels, bias=self._bias, **kw)
        if self.root_weight:
            self.lin_r = torch.nn.Linear(in_dst, self.out_channels, bias=False, **kw)

    def forward(self, x, edge_index, size=None):
        x_src, x_dst = _pair(x)
        if self.lin_l is None:   # in_channels = -1: sized by the first input, like PyG's lazy Linear
            self._build(x_src.size(1), x_dst.size(1), like=x_src)
        n_dst = x_dst.size(0) if size is None else size[1]
        d_src = x_src.size(1)
        if self.root_weight:
            if _wants_grad(x_src, x_dst):     # in a graph the mean is a tensor of its own (the out= form has no backward)
                h = torch.cat([edge_reduce("copy", x_src.contiguous(), edge_index, n_dst, aggr=("mean",)), x_dst[:n_dst]], dim=1)
            else:
                h = torch.empty((n_dst, d_src + x_dst.size(1)), dtype=x_src.dtype, device=x_src.device)
                h[:, d_src:] = x_dst[:n_dst]
                edge_reduce("copy", x_src.contiguous(), edge_index, n_dst, aggr=("mean",), out=h[:, :d_src])
            packed = self._packed.get([self.lin_l.weight, self.lin_r.weight, self.lin_l.bias],
                                      [(self.lin_l.weight, self.lin_l.bias), (self.lin_r.weight, None)], stack=True)
        else:
            h = edge_reduce("copy", x_src.contiguous(), edge_index, n_dst, aggr=("mean",))
            packed = self._packed.get([self.lin_l.weight, self.lin_l.bias], [(self.lin_l.weight, self.lin_l.bias)])
        out = _dense(h, packed)
        return torch.nn.functional.normalize(out, p=2.0, dim=-1) if self.normalize else out


class CGConv(_Layer):
    """x'_i = x_i + sum_j sigmoid(z_ij W_f + b_f) * softplus(z_ij W_s + b_s),  z_ij = [x_i, x_j, e_ij]
    (app_bm/groq_script.py:15-111; forward :91-102, message :104-109)."""

    def __init__(self, channels, dim=0, aggr="add", batch_norm=False, bias=True):
        super().__init__()
        self.channels, self.dim, self.aggr, self.batch_norm = channels, dim, aggr, batch_norm
        self._ch = (channels, channels) if isinstance(channels, int) else tuple(channels)
        self.lin_f = torch.nn.Linear(sum(self._ch) + dim, self._ch[1], bias=bias)
        self.lin_s = torch.nn.Linear(sum(self._ch) + dim, self._ch[1], bias=bias)
        self.bn = torch.nn.BatchNorm1d(self._ch[1]) if batch_norm else None
        self._pk_both, self._pk_dst, self._pk_src, self._pk_edge = _Packed(), _Packed(), _Packed(), _Packed()

    def forward(self, x, edge_index, edge_attr=None):
        x_src, x_dst = _pair(x)
        c_src, c_dst = self._ch
        K = c_dst
        Wf, Ws, bf, bs = self.lin_f.weight, self.lin_s.weight, self.lin_f.bias, self.lin_s.bias
        params = [Wf, Ws, bf, bs]
        # the weight columns follow cat([x_i, x_j, e]) (groq_script.py:105-108)
        dst_blocks = [(Wf[:, :c_dst], bf), (Ws[:, :c_dst], bs)]                                # -> p = [f | s], biases included
        src_blocks = [(Wf[:, c_dst:c_dst + c_src], None), (Ws[:, c_dst:c_dst + c_src], None)]  # -> q = [f | s]
        n_dst = x_dst.size(0)
        if x_src is x_dst:   # one product for both sides: [p | q] = x @ [W_f,i | W_s,i | W_f,j | W_s,j]
            pq = _dense(x_dst.contiguous(), self._pk_both.get(params, dst_blocks + src_blocks))   # its own cache: 4K columns, not 2K
            p, q = pq[:, :2 * K], pq[:, 2 * K:]
        else:
            p = _dense(x_dst.contiguous(), self._pk_dst.get(params, dst_blocks))
            q = _dense(x_src.contiguous(), self._pk_src.get(params, src_blocks))
        w = None
        if edge_attr is not None:
            if edge_attr.dim() == 1:
                edge_attr = edge_attr.unsqueeze(-1)
            w = _dense(edge_attr.contiguous(), self._pk_edge.get(params, [(Wf[:, c_dst + c_src:], None), (Ws[:, c_dst + c_src:], None)]))
        aggr = "sum" if self.aggr == "add" else self.aggr
        if self.bn is None:
            return edge_reduce("cgconv", q, edge_index, n_dst, p=p, w=w, add=x_dst.contiguous(), aggr=(aggr,))
        out = self.bn(edge_reduce("cgconv", q, edge_index, n_dst, p=p, w=w, aggr=(aggr,)))
        out += x_dst
        return out


class FiLMConv(_Layer):
    """x'_i = sum_r mean_{j in N_r(i)} relu(gamma_r,i * W_r x_j + beta_r,i) + relu(gamma_s,i * W_s x_i + beta_s,i)
    (torch_geometric FiLMConv, aggr="mean", act=ReLU; benchmark_convs.py:146 FiLMConv(in_channels=11, out_channels=2048)).
    One product gives [beta_s | gamma_s | W_s x | (beta_r | gamma_r | W_r x) for every relation] per node."""

    def __init__(self, in_channels, out_channels, num_relations=1, nn=None, act=torch.nn.ReLU(), aggr="mean"):
        super().__init__()
        if nn is not None:
            raise NotImplementedError("gnnops.conv.FiLMConv: a custom film network is not fused; pass nn=None")
        if not isinstance(act, torch.nn.ReLU):
            raise NotImplementedError("gnnops.conv.FiLMConv: act must be ReLU (the edge pass applies it)")
        if isinstance(in_channels, (tuple, list)):
            raise NotImplementedError("gnnops.conv.FiLMConv: bipartite input")
        self.in_channels, self.out_channels, self.num_relations = in_channels, out_channels, max(num_relations, 1)
        self.act, self.aggr = act, aggr
        R = self.num_relations
        self.lins = torch.nn.ModuleList([torch.nn.Linear(in_channels, out_channels, bias=False) for _ in range(R)])
        self.films = torch.nn.ModuleList([torch.nn.Linear(in_channels, 2 * out_channels) for _ in range(R)])
        self.lin_skip = torch.nn.Linear(in_channels, out_channels, bias=False)
        self.film_skip = torch.nn.Linear(in_channels, 2 * out_channels, bias=False)
        self._packed = _Packed()

    def forward(self, x, edge_index, edge_type=None):
        if isinstance(x, (tuple, list)):
            raise NotImplementedError("gnnops.conv.FiLMConv: bipartite input")
        o, R = self.out_channels, self.num_relations
        blocks = [(self.film_skip.weight, None), (self.lin_skip.weight, None)]
        for r in range(R):
            blocks += [(self.films[r].weight, self.films[r].bias), (self.lins[r].weight, None)]
        y = _dense(x.contiguous(), self._packed.get(list(self.parameters()), blocks))   # [N, 3 o (1 + R)]
        beta_s, gamma_s, xs = y[:, :o], y[:, o:2 * o], y[:, 2 * o:3 * o]
        out = torch.relu_(gamma_s * xs + beta_s)
        aggr = "sum" if self.aggr == "add" else self.aggr
        n = x.size(0)
        for r in range(R):
            base = 3 * o * (1 + r)     # film(x) = [beta | gamma] (FiLMConv.forward: .split(out_channels, dim=-1)), then W_r x
            ei = edge_index if R == 1 else edge_index[:, edge_type == r].contiguous()
            out = edge_reduce("film", y[:, base + 2 * o:base + 3 * o], ei, n, p=y[:, base:base + 2 * o], add=out, aggr=(aggr,))
        return out


class PNAConv(_Layer):
    """Principal neighbourhood aggregation (torch_geometric PNAConv; benchmark_convs.py:197-206: in 1, out 2048,
    aggregators mean/min/max/std, scalers identity/amplification/attenuation, deg = in-degree histogram).
    message = pre_nn([x_i, x_j (, enc(e))]) with ONE pre-layer is p_i + q_j (+ w_e); the aggregators and scalers come out of
    one edge pass, written next to x into the [N, (1 + A S) F] operand of the post layer (inference; when a gradient is needed
    the layer runs `_forward_train`: the same mathematics as a chain of this package's differentiable ops). pre_layers > 1: the first layer is
    still split per node, the rest of the MLP runs on per-edge rows (see forward); post_layers > 1: more node-row products."""

    def __init__(self, in_channels, out_channels, aggregators, scalers, deg, edge_dim=None, towers=1, pre_layers=1,
                 post_layers=1, divide_input=False):
        super().__init__()
        if pre_layers < 1 or post_layers < 1:
            raise ValueError("PNAConv: pre_layers and post_layers must be >= 1")
        self.pre_layers, self.post_layers = pre_layers, post_layers
        if divide_input and in_channels % towers or out_channels % towers:
            raise ValueError("PNAConv: channels must divide by towers")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aggregators, self.scalers = list(aggregators), list(scalers)
        self.edge_dim, self.towers, self.divide_input = edge_dim, towers, divide_input
        self.F_in = in_channels // towers if divide_input else in_channels
        self.F_out = out_channels // towers
        deg = deg.to(torch.float)
        total = deg.sum()
        bins = torch.arange(deg.numel(), dtype=torch.float)
        self.avg_deg = {"lin": float((bins * deg).sum() / total), "log": float(((bins + 1).log() * deg).sum() / total),
                        "exp": float((bins.exp() * deg).sum() / total)}
        F = self.F_in
        if edge_dim is not None:
            self.edge_encoder = torch.nn.Linear(edge_dim, F)
        def mlp(first_in, width_out, layers):   # PyG's PNAConv: Linear, then (ReLU, Linear) per extra layer
            mods = [torch.nn.Linear(first_in, width_out)]
            for _ in range(layers - 1):
                mods += [torch.nn.ReLU(), torch.nn.Linear(width_out, width_out)]
            return torch.nn.Sequential(*mods)

        self.pre_nns = torch.nn.ModuleList([mlp((3 if edge_dim else 2) * F, F, pre_layers) for _ in range(towers)])
        width = (len(self.aggregators) * len(self.scalers) + 1) * F
        self.post_nns = torch.nn.ModuleList([mlp(width, self.F_out, post_layers) for _ in range(towers)])
        self.lin = torch.nn.Linear(out_channels, out_channels)
        self._pk = {}

    def _cache(self, name):
        if name not in self._pk:
            self._pk[name] = _Packed()
        return self._pk[name]

    def _forward_train(self, x, edge_index, edge_attr):
        """In a graph that needs gradients: the propagate-order chain — per-edge messages [E, F], one differentiable scatter per
        aggregator (min / max route the gradient to their arg, std through its two means), the degree scalers as constant
        factors — built from the differentiable front ends of this package's own kernels (gnnops.autograd: index_select,
        scatter, addmm). The fused multi-aggregator edge pass has no backward; inference keeps it."""
        from . import autograd as ad

        F, T = self.F_in, self.towers
        n = x.size(0)
        src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
        xt = x.view(n, T, F) if self.divide_input else x.view(n, 1, F).expand(n, T, F)

        def lin(z, layer):
            return ad.addmm(layer.bias, z.contiguous(), layer.weight.t().contiguous())

        def mlp(z, seq):
            z = lin(z, seq[0])
            for li in range(1, (len(seq) + 1) // 2):
                z = lin(torch.relu(z), seq[2 * li])
            return z

        e = lin(edge_attr, self.edge_encoder) if self.edge_dim is not None else None
        deg = ad.scatter(torch.ones(dst.numel(), 1, dtype=x.dtype, device=x.device), dst, 0, None, n, "sum").clamp_(min=1)
        logd = torch.log(deg + 1)
        fac = {"identity": None, "amplification": logd / self.avg_deg["log"], "attenuation": self.avg_deg["log"] / logd,
               "linear": deg / self.avg_deg["lin"], "inverse_linear": self.avg_deg["lin"] / deg}
        outs = []
        for t in range(T):
            xin = xt[:, t].contiguous()
            z = [ad.index_select(xin, 0, dst), ad.index_select(xin, 0, src)] + ([e] if e is not None else [])
            m = mlp(torch.cat(z, dim=1), self.pre_nns[t])
            aggs = []
            for a in self.aggregators:
                if a == "std":
                    mean = ad.scatter(m, dst, 0, None, n, "mean")
                    aggs.append(torch.sqrt(torch.relu(ad.scatter(m * m, dst, 0, None, n, "mean") - mean * mean) + 1e-5))
                else:
                    r = ad.scatter(m, dst, 0, None, n, "sum" if a == "add" else a)
                    aggs.append(r[0] if isinstance(r, tuple) else r)
            out = torch.cat(aggs, dim=1)
            out = torch.cat([out if fac[sc] is None else out * fac[sc] for sc in self.scalers], dim=1) if self.scalers else out
            outs.append(mlp(torch.cat([xin, out], dim=1), self.post_nns[t]))
        return lin(outs[0] if T == 1 else torch.cat(outs, dim=1), self.lin)

    def forward(self, x, edge_index, edge_attr=None):
        if self.edge_dim is not None and edge_attr is None:
            raise RuntimeError("PNAConv: edge_attr is required when edge_dim is set")
        if _wants_grad(x, edge_attr, *self.parameters()):
            return self._forward_train(x, edge_index, edge_attr)
        F, T = self.F_in, self.towers
        n = x.size(0)
        xt = x.view(n, T, F) if self.divide_input else x.view(n, 1, F).expand(n, T, F)
        A, S = len(self.aggregators), len(self.scalers)
        e = None
        if self.edge_dim is not None:
            if edge_attr is None:
                raise RuntimeError("PNAConv: edge_attr is required when edge_dim is set")
            e = _linear(edge_attr.contiguous(), self.edge_encoder, self._cache("enc"))
        outs = []
        for t in range(T):
            pre, post = self.pre_nns[t][0], self.post_nns[t][0]
            xin = xt[:, t].contiguous()
            Wp = pre.weight                                              # [F, 2F or 3F]: columns follow cat([x_i, x_j, e])
            pq = _dense(xin, self._cache(f"pre{t}").get([Wp, pre.bias], [(Wp[:, :F], pre.bias), (Wp[:, F:2 * F], None)]))
            w = _dense(e, self._cache(f"edge{t}").get([Wp], [(Wp[:, 2 * F:], None)])) if e is not None else None
            h = torch.empty((n, (1 + A * S) * F), dtype=x.dtype, device=x.device)
            h[:, :F] = xin
            if self.pre_layers == 1:
                edge_reduce("add", pq[:, F:], edge_index, n, p=pq[:, :F], w=w, aggr=self.aggregators, scalers=self.scalers,
                            avg_deg=self.avg_deg, out=h[:, F:])
            else:
                # a deeper pre-MLP is not linear in [x_i, x_j]: only its FIRST layer leaves the edge loop (the split product
                # above); its output is materialised per edge, the remaining (ReLU, Linear) pairs run as [E, F] products, and
                # the aggregators take the finished messages as "copy" rows of a graph whose sources are the edges themselves
                src_rows, dst_rows = edge_index[0].contiguous(), edge_index[1].contiguous()
                m = ops.index_select(pq[:, :F].contiguous(), 0, dst_rows) + ops.index_select(pq[:, F:].contiguous(), 0, src_rows)
                if w is not None:
                    m = m + w
                for li in range(1, self.pre_layers):
                    m = _linear(torch.relu_(m), self.pre_nns[t][2 * li], self._cache(f"pre{t}.{li}"))
                per_edge = torch.stack([torch.arange(m.size(0), device=m.device), dst_rows])
                edge_reduce("copy", m, per_edge, n, aggr=self.aggregators, scalers=self.scalers, avg_deg=self.avg_deg, out=h[:, F:])
            o = _linear(h, post, self._cache(f"post{t}"))
            for li in range(1, self.post_layers):
                o = _linear(torch.relu_(o), self.post_nns[t][2 * li], self._cache(f"post{t}.{li}"))
            outs.append(o)
        out = outs[0] if T == 1 else torch.cat(outs, dim=1)
        return _linear(out, self.lin, self._cache("lin"))
