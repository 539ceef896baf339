"""CPU oracle for the single-layer forward passes. TEST INFRASTRUCTURE ONLY: only ``tests/`` may
import this module; the product package never does.

Each function restates a layer the reference times (app_bm/benchmark_convs.py:146-246) in the order PyG's
MessagePassing.propagate runs it — gather x_i / x_j per edge, concatenate, apply the layer's Linear maps PER EDGE, message,
scatter-reduce by destination — in float64 numpy. That is deliberately NOT how the product computes them (it splits every
Linear into per-node products and never builds a per-edge tensor), so agreement checks the algebra as well as the kernel.

Pinning status
  cg_conv    follows the text of the layer in the reference: app_bm/groq_script.py:91-102 (forward: propagate, optional
             batch norm, `out += x[1]`) and :104-109 (message: z = cat([x_i, x_j, edge_attr]); lin_f(z).sigmoid() *
             softplus(lin_s(z))); softplus = torch's (beta 1, threshold 20). Aggregation "add" (:62).
  gin_conv, sage_conv, film_conv, pna_conv   torch_geometric 2.0.2 (requirements.txt:211) is absent from /root/reference and
             from this image, and the reference holds no output of these layers: restated from the published layer
             definitions — PARITY UNPINNED. Degree scalers and the std aggregator follow PNAConv.aggregate:
             std = sqrt(relu(mean(m^2) - mean(m)^2) + 1e-5); amplification = log(deg + 1) / avg_deg_log; attenuation =
             avg_deg_log / log(deg + 1); deg clamped to >= 1; groups without edges aggregate to 0 (torch_scatter).
edge_index is int64 [2, E] = (source j, destination i) — flow "source_to_target".
"""
import numpy as np


def _lin(z, W, b=None):
    y = z @ np.asarray(W, np.float64).T
    return y if b is None else y + np.asarray(b, np.float64)


def scatter(m, index, n, reduce):
    """torch_scatter.scatter(m, index, 0, dim_size=n, reduce): groups nothing reaches are 0 for every reduction."""
    m = np.asarray(m, np.float64)
    out = np.zeros((n,) + m.shape[1:], np.float64)
    if reduce in ("sum", "add", "mean"):
        np.add.at(out, index, m)
        if reduce == "mean":
            cnt = np.bincount(index, minlength=n).astype(np.float64)
            out /= np.maximum(cnt, 1.0).reshape((n,) + (1,) * (m.ndim - 1))
        return out
    init = np.inf if reduce == "min" else -np.inf
    out[:] = init
    (np.minimum if reduce == "min" else np.maximum).at(out, index, m)
    out[out == init] = 0.0
    return out


def softplus(x):
    return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0))))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gin_conv(x, edge_index, W, b, eps=0.0):
    x = np.asarray(x, np.float64)
    src, dst = edge_index
    h = scatter(x[src], dst, x.shape[0], "sum") + (1.0 + eps) * x
    return _lin(h, W, b)


def sage_conv(x, edge_index, W_l, b_l, W_r=None, normalize=False):
    x = np.asarray(x, np.float64)
    src, dst = edge_index
    out = _lin(scatter(x[src], dst, x.shape[0], "mean"), W_l, b_l)
    if W_r is not None:
        out = out + _lin(x, W_r)
    if normalize:
        out = out / np.maximum(np.linalg.norm(out, axis=-1, keepdims=True), 1e-12)
    return out


def cg_conv(x, edge_index, W_f, b_f, W_s, b_s, edge_attr=None, aggr="add"):
    """groq_script.py:91-109."""
    x = np.asarray(x, np.float64)
    src, dst = edge_index
    parts = [x[dst], x[src]]                          # x_i, x_j
    if edge_attr is not None:
        ea = np.asarray(edge_attr, np.float64)
        parts.append(ea.reshape(len(src), -1))
    z = np.concatenate(parts, axis=-1)
    m = sigmoid(_lin(z, W_f, b_f)) * softplus(_lin(z, W_s, b_s))
    return scatter(m, dst, x.shape[0], "sum" if aggr == "add" else aggr) + x


def film_conv(x, edge_index, lins, films, lin_skip, film_skip, edge_type=None, aggr="mean"):
    """lins: [W_r]; films: [(W, b)] giving [beta | gamma]; lin_skip: W; film_skip: W (no bias)."""
    x = np.asarray(x, np.float64)
    src, dst = edge_index
    o = np.asarray(lin_skip).shape[0]
    fs = _lin(x, film_skip)
    beta, gamma = fs[:, :o], fs[:, o:]
    out = np.maximum(gamma * _lin(x, lin_skip) + beta, 0.0)
    for r, (W_r, (W_film, b_film)) in enumerate(zip(lins, films)):
        f = _lin(x, W_film, b_film)
        beta, gamma = f[:, :o], f[:, o:]
        sel = np.ones(len(src), bool) if edge_type is None or len(lins) == 1 else (np.asarray(edge_type) == r)
        s, d = src[sel], dst[sel]
        m = np.maximum(gamma[d] * _lin(x, W_r)[s] + beta[d], 0.0)
        out = out + scatter(m, d, x.shape[0], "sum" if aggr == "add" else aggr)
    return out


def _mlp(z, layers):
    """PNAConv's pre / post networks: Linear, then (ReLU, Linear) per extra layer; `layers` = (W, b) or a list of them."""
    if isinstance(layers, tuple):
        layers = [layers]
    z = _lin(z, *layers[0])
    for W, b in layers[1:]:
        z = _lin(np.maximum(z, 0.0), W, b)
    return z


def pna_conv(x, edge_index, pre, post, lin, aggregators, scalers, avg_deg, edge_attr=None, edge_encoder=None, towers=1,
             divide_input=False):
    """pre / post: per tower (W, b) of a single pre / post layer, or the list of (W, b) of a deeper one; lin: (W, b);
    edge_encoder: (W, b) or None."""
    x = np.asarray(x, np.float64)
    src, dst = edge_index
    n = x.shape[0]
    F = x.shape[1] // towers if divide_input else x.shape[1]
    xt = x.reshape(n, towers, F) if divide_input else np.repeat(x.reshape(n, 1, F), towers, axis=1)
    e = None
    if edge_attr is not None:
        e = _lin(np.asarray(edge_attr, np.float64).reshape(len(src), -1), *edge_encoder)
    deg = np.maximum(np.bincount(dst, minlength=n).astype(np.float64), 1.0).reshape(n, 1)
    outs = []
    for t in range(towers):
        xi, xj = xt[dst, t], xt[src, t]
        h = np.concatenate([xi, xj] + ([e] if e is not None else []), axis=-1)
        m = _mlp(h, pre[t])
        aggs = []
        for a in aggregators:
            if a == "std":
                mean = scatter(m, dst, n, "mean")
                msq = scatter(m * m, dst, n, "mean")
                aggs.append(np.sqrt(np.maximum(msq - mean * mean, 0.0) + 1e-5))
            else:
                aggs.append(scatter(m, dst, n, a))
        out = np.concatenate(aggs, axis=-1)
        scaled = []
        for s in scalers:
            if s == "identity":
                scaled.append(out)
            elif s == "amplification":
                scaled.append(out * (np.log(deg + 1.0) / avg_deg["log"]))
            elif s == "attenuation":
                scaled.append(out * (avg_deg["log"] / np.log(deg + 1.0)))
            elif s == "linear":
                scaled.append(out * (deg / avg_deg["lin"]))
            elif s == "inverse_linear":
                scaled.append(out * (avg_deg["lin"] / deg))
            else:
                raise ValueError(s)
        out = np.concatenate([xt[:, t]] + scaled, axis=-1)
        outs.append(_mlp(out, post[t]))
    return _lin(np.concatenate(outs, axis=1), *lin)
