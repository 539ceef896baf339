"""GPU parity for the SURVEY.md §8(f) rank-4 widening: gnnops_edge_reduce (csrc/conv.hip) and the five single-layer
forward passes the reference's app benchmarks time (app_bm/benchmark_convs.py:146-246; CGConv's text:
app_bm/groq_script.py:91-109).

The oracle (oracle/conv_oracle.py) runs every layer the way MessagePassing.propagate does — per-edge gather, concat, Linear
per edge, message, scatter — in float64; the product splits the Linear maps into per-node products and runs one fused
edge pass. Bars (stated per test): fp32 within 2e-5 of the value scale (fp32 sums of <= a few hundred terms in a different
order + device exp/log), fp16 / bf16 within a few storage-type ulps of the value scale (the per-node projections are
rounded to the storage type once before the edge pass; the reference's fp16 run rounds after every op).
CGConv is pinned by the reference's own layer text; GIN / SAGE / FiLM / PNA are restated from the published PyG 2.0.2
definitions — parity unpinned (oracle/conv_oracle.py header).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}


@pytest.fixture(scope="module")
def conv():
    import gnnops
    from gnnops import conv as c

    gnnops.load_library()
    return c


@pytest.fixture(scope="module")
def ora():
    from oracle import conv_oracle

    return conv_oracle


def _graph(seed, n, e, n_src=None, isolated=True):
    g = torch.Generator().manual_seed(seed)
    n_src = n if n_src is None else n_src
    src = torch.randint(0, n_src, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    if isolated and n > 8 and e:
        dst[dst == 3] = 4          # node 3 has no incoming edge
        dst[:40] = 5               # node 5 is a (small) hub
    return torch.stack([src, dst])


def _rand(g, *shape, dtype=torch.float32, scale=1.0):
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(dtype)


def _f64(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _close(got, want, dtype, what, scale=None):
    got = _f64(got)
    scale = max(np.abs(want).max(), 1e-6) if scale is None else scale
    err = np.abs(got - want).max() / scale
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert err <= TOL[dtype], f"{what}: max error {err:.3e} of the value scale exceeds {TOL[dtype]:.1e}"


# ---------------------------------------------------------------------------------------------------------------------
# the edge pass itself, functor by functor, against a brute-force loop
# ---------------------------------------------------------------------------------------------------------------------
def _brute(functor, q, p, w, add, ei, n, K, aggr, scalers, avg):
    src, dst = ei
    q, p, w = (None if t is None else _f64(t) for t in (q, p, w))
    if functor == "copy":
        m = q[src]
    elif functor == "add":
        m = q[src] + (p[dst] if p is not None else 0) + (w if w is not None else 0)
    elif functor == "cgconv":
        z = q[src] + (p[dst] if p is not None else 0) + (w if w is not None else 0)
        f, s = z[:, :K], z[:, K:]
        m = 1 / (1 + np.exp(-f)) * np.where(s > 20, s, np.log1p(np.exp(np.minimum(s, 20))))
    else:
        m = np.maximum(p[dst][:, K:] * q[src] + p[dst][:, :K], 0)
    from oracle.conv_oracle import scatter

    blocks = []
    deg = np.maximum(np.bincount(dst, minlength=n), 1).astype(np.float64).reshape(n, 1)
    for a in aggr:
        if a == "std":
            mean = scatter(m, dst, n, "mean")
            blocks.append(np.sqrt(np.maximum(scatter(m * m, dst, n, "mean") - mean * mean, 0) + 1e-5))
        else:
            blocks.append(scatter(m, dst, n, a))
    out = np.concatenate(blocks, -1)
    avg = avg or {"log": 1.0, "lin": 1.0}
    fac = {"identity": 1.0, "amplification": np.log(deg + 1) / avg["log"], "attenuation": avg["log"] / np.log(deg + 1),
           "linear": deg / avg["lin"], "inverse_linear": avg["lin"] / deg}
    out = np.concatenate([out * fac[s] for s in scalers], -1) if scalers else out
    if add is not None:
        out[:, :K] += _f64(add)
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("functor", ["copy", "add", "cgconv", "film"])
@pytest.mark.parametrize("K", [64, 1, 13, 200])   # whole 16-B lanes; the element form (K = 1 is PNA on MNIST); ragged; > one chunk
def test_edge_reduce_functors(conv, functor, dtype, K):
    n, e = 300, 2500
    g = torch.Generator().manual_seed(K * 7 + len(functor))
    ei = _graph(K, n, e)
    nq, npp, nw = conv._PARTS[functor]
    q = _rand(g, n, nq * K, dtype=dtype, scale=2.0)
    p = _rand(g, n, npp * K, dtype=dtype, scale=2.0) if npp else None
    w = _rand(g, e, nw * K, dtype=dtype) if nw else None
    add = _rand(g, n, K, dtype=dtype)
    dev = lambda t: None if t is None else t.cuda()   # noqa: E731
    got = conv.edge_reduce(functor, dev(q), ei.cuda(), n, p=dev(p), w=dev(w), add=dev(add), aggr=("sum",))
    want = _brute(functor, q, p, w, add, ei.numpy(), n, K, ("sum",), (), None)
    _close(got, want, dtype, f"{functor} sum K={K}")
    # the PNA set in one pass, with every scaler
    aggr, scal = ("mean", "min", "max", "std"), ("identity", "amplification", "attenuation", "linear", "inverse_linear")
    avg = {"log": 1.7, "lin": 6.0}
    got = conv.edge_reduce(functor, dev(q), ei.cuda(), n, p=dev(p), w=dev(w), aggr=aggr, scalers=scal, avg_deg=avg)
    want = _brute(functor, q, p, w, None, ei.numpy(), n, K, aggr, scal, avg)
    assert got.shape == (n, 20 * K)
    _close(got, want, dtype, f"{functor} multi K={K}")


def test_edge_reduce_column_blocks_and_empty(conv):
    """Operands as column blocks of wider matrices (row pitch != row length), an output block inside a wider buffer, no
    edges at all, and destinations nothing reaches (0 for every aggregator, sqrt(1e-5) for std)."""
    n, e, K = 64, 300, 16
    g = torch.Generator().manual_seed(3)
    ei = _graph(3, n, e)
    wide = _rand(g, n, 5 * K).cuda()
    p, q = wide[:, K:3 * K], wide[:, 3 * K:]
    buf = torch.full((n, 4 * K), 7.0, device="cuda")
    conv.edge_reduce("cgconv", q, ei.cuda(), n, p=p, aggr=("sum", "max"), out=buf[:, K:3 * K])
    want = _brute("cgconv", q.cpu(), p.cpu(), None, None, ei.numpy(), n, K, ("sum", "max"), (), None)
    _close(buf[:, K:3 * K], want, torch.float32, "column blocks")
    assert bool((buf[:, :K] == 7).all()) and bool((buf[:, 3 * K:] == 7).all()), "wrote outside its block"
    none = torch.zeros((2, 0), dtype=torch.int64, device="cuda")
    got = conv.edge_reduce("copy", wide[:, :K].contiguous(), none, n, aggr=("sum", "mean", "min", "max", "std"))
    assert bool((got[:, :4 * K] == 0).all())
    assert torch.allclose(got[:, 4 * K:], torch.full((n, K), 1e-5 ** 0.5, device="cuda"))


@pytest.mark.parametrize("K,dtype", [(64, torch.float16), (13, torch.float32), (128, torch.float16)])
def test_edge_reduce_hub_destinations(conv, K, dtype):
    """Power-law graphs: destinations with more than 8192 edges are set aside and reduced in pieces of 2048 edges (fp32 partial
    accumulators combined in piece order). A 60 000-edge hub, a 9 000-edge one and one of exactly 8192 (not a hub) among
    ordinary rows; sum with residual and the full PNA set; cgconv at K = 128 takes the one-row-per-wave form."""
    n, e = 3000, 200_000
    g = torch.Generator().manual_seed(K)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    dst[:60_000] = 17
    dst[60_000:69_000] = 1234
    dst[dst == 99] = 100
    dst[69_000:69_000 + 8192] = 99
    perm = torch.randperm(e, generator=g)
    ei = torch.stack([src, dst])[:, perm]
    functor = "cgconv" if K == 128 else "add"
    nq, npp, nw = conv._PARTS[functor]
    q = _rand(g, n, nq * K, dtype=dtype)
    p = _rand(g, n, npp * K, dtype=dtype)
    add = _rand(g, n, K, dtype=dtype)
    got = conv.edge_reduce(functor, q.cuda(), ei.cuda(), n, p=p.cuda(), add=add.cuda(), aggr=("sum",))
    want = _brute(functor, q, p, None, add, ei.numpy(), n, K, ("sum",), (), None)
    err = np.abs(_f64(got) - want) / np.maximum(np.abs(want), 1.0)          # per element, relative to max(|value|, 1): hubs sum 60 000 terms
    assert err.max() <= (2e-5 if dtype == torch.float32 else 2e-3), err.max()
    aggr, scal = ("mean", "min", "max", "std"), ("identity", "amplification", "attenuation")
    avg = {"log": 3.0, "lin": 60.0}
    got = conv.edge_reduce(functor, q.cuda(), ei.cuda(), n, p=p.cuda(), aggr=aggr, scalers=scal, avg_deg=avg)
    want = _brute(functor, q, p, None, None, ei.numpy(), n, K, aggr, scal, avg)
    _close(got, want, dtype, "hub multi")
    for row in (17, 1234, 99):
        _close(got[row:row + 1], want[row:row + 1], dtype, f"row {row}")


def test_edge_reduce_refuses_bad_arguments(conv):
    x = torch.rand(10, 8, device="cuda")
    ei = torch.randint(0, 10, (2, 30), device="cuda")
    with pytest.raises(RuntimeError):
        conv.edge_reduce("film", x, ei, 10)                                      # no [beta | gamma]
    with pytest.raises(RuntimeError):
        conv.edge_reduce("cgconv", x[:, :7], ei, 10)                             # odd width for a two-part row
    with pytest.raises(RuntimeError):
        conv.edge_reduce("copy", x, ei, 10, add=torch.rand(9, 8, device="cuda"))  # one row per destination
    with pytest.raises(RuntimeError):
        conv.edge_reduce("copy", x.half(), ei, 10, add=x)                         # dtype mismatch
    with pytest.raises((RuntimeError, ValueError)):
        conv.edge_reduce("copy", x.cpu(), ei.cpu(), 10)                           # CPU tensors are refused, not emulated
    xg = x.clone().requires_grad_(True)
    with pytest.raises(NotImplementedError):                                          # a form without a backward refuses a gradient
        conv.edge_reduce("copy", xg, ei, 10, aggr=("max",))
    assert conv.edge_reduce("copy", xg, ei, 10).requires_grad                        # sum / mean of a copy message has one


@pytest.mark.parametrize("E,N", [(1, 1), (40, 21), (1023, 7), (1024, 1024), (1025, 40000), (18744, 9134), (24576, 300), (24576, 40000)])
def test_one_launch_plan_equals_the_radix_plan(E, N, monkeypatch):
    """csrc/plan.hip plan_small_kernel (one workgroup, counters in LDS) against the radix build: rowptr, perm (stable:
    ascending positions inside every destination) and the companion column, bit for bit; a hub destination included."""
    import gnnops

    g = torch.Generator().manual_seed(E + N)
    idx = torch.randint(0, N, (E,), generator=g)
    if E > 100:
        idx[torch.randint(0, E, (E // 3,), generator=g)] = N // 2        # a hub: a third of all positions
    comp = torch.randint(0, 1 << 40, (E,), generator=g)
    small = gnnops.Plan(idx.cuda(), N, comp.cuda())
    assert small.col is not None, "the one-launch build did not run"
    monkeypatch.setenv("GNNOPS_PLAN_SMALL", "0")
    radix = gnnops.Plan(idx.cuda(), N, comp.cuda())
    assert radix.col is None
    assert torch.equal(small.rowptr, radix.rowptr)
    assert torch.equal(small.perm[:E], radix.perm[:E])
    assert torch.equal(small.col.cpu(), comp[radix.perm[:E].long().cpu()])
    order = torch.sort(idx, stable=True).indices
    assert torch.equal(small.perm[:E].cpu().long(), order)


# ---------------------------------------------------------------------------------------------------------------------
# the layers
# ---------------------------------------------------------------------------------------------------------------------
def _np_params(mod):
    return {k: _f64(v) for k, v in mod.state_dict().items()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("channels,dim", [(11, 0), (64, 0), (32, 5), ((24, 16), 3)])
def test_cgconv_matches_the_reference_layer_text(conv, ora, dtype, channels, dim):
    """groq_script.py:119-127: CGConv(num_features=11, dim 0) on x [29, 11] fp16 — plus wider, edge-featured and bipartite."""
    torch.manual_seed(1)
    c_src, c_dst = (channels, channels) if isinstance(channels, int) else channels
    n_src, n_dst, e = (29, 29, 56) if channels == 11 else (500, 500 if isinstance(channels, int) else 350, 4000)
    layer = conv.CGConv(channels, dim).to(dtype).cuda()
    g = torch.Generator().manual_seed(5)
    ei = _graph(2, n_dst, e, n_src=n_src, isolated=n_dst > 100)
    x_src = _rand(g, n_src, c_src, dtype=dtype)
    x_dst = x_src if isinstance(channels, int) else _rand(g, n_dst, c_dst, dtype=dtype)
    ea = _rand(g, e, dim, dtype=dtype) if dim else None
    with torch.no_grad():
        x = x_src.cuda() if isinstance(channels, int) else (x_src.cuda(), x_dst.cuda())
        got = layer(x, ei.cuda(), None if ea is None else ea.cuda())
    P = _np_params(layer)
    if isinstance(channels, int):
        want = ora.cg_conv(_f64(x_src), ei.numpy(), P["lin_f.weight"], P["lin_f.bias"], P["lin_s.weight"], P["lin_s.bias"],
                           None if ea is None else _f64(ea))
    else:   # bipartite: x_i from the destination side, x_j from the source side
        src, dst = ei.numpy()
        z = np.concatenate([_f64(x_dst)[dst], _f64(x_src)[src]] + ([_f64(ea)] if ea is not None else []), -1)
        m = ora.sigmoid(z @ P["lin_f.weight"].T + P["lin_f.bias"]) * ora.softplus(z @ P["lin_s.weight"].T + P["lin_s.bias"])
        want = ora.scatter(m, dst, n_dst, "sum") + _f64(x_dst)
    _close(got, want, dtype, "CGConv")


def test_cgconv_batch_norm_and_mean(conv, ora):
    torch.manual_seed(2)
    layer = conv.CGConv(16, 0, aggr="mean", batch_norm=True).cuda().eval()
    layer.bn.running_mean.uniform_(-0.1, 0.1)
    layer.bn.running_var.uniform_(0.5, 1.5)
    g = torch.Generator().manual_seed(6)
    ei = _graph(4, 200, 1500)
    x = _rand(g, 200, 16)
    with torch.no_grad():
        got = layer(x.cuda(), ei.cuda())
    P = _np_params(layer)
    agg = ora.cg_conv(_f64(x), ei.numpy(), P["lin_f.weight"], P["lin_f.bias"], P["lin_s.weight"], P["lin_s.bias"], aggr="mean") - _f64(x)
    want = (agg - P["bn.running_mean"]) / np.sqrt(P["bn.running_var"] + 1e-5) * P["bn.weight"] + P["bn.bias"] + _f64(x)
    _close(got, want, torch.float32, "CGConv bn")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_gin_conv(conv, ora, dtype):
    """benchmark_convs.py:163: GINConv(Linear(11, 2048)); plus eps != 0."""
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(7)
    n, e = 700, 3000
    ei = _graph(5, n, e)
    x = _rand(g, n, 11, dtype=dtype)
    for eps in (0.0, 0.25):
        layer = conv.GINConv(torch.nn.Linear(11, 2048), eps=eps).to(dtype).cuda()
        with torch.no_grad():
            got = layer(x.cuda(), ei.cuda())
        P = _np_params(layer)
        _close(got, ora.gin_conv(_f64(x), ei.numpy(), P["nn.weight"], P["nn.bias"], eps), dtype, f"GIN eps={eps}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_sage_conv(conv, ora, dtype):
    """benchmark_convs.py:231: SAGEConv(-1, 2048) on IMDB-MULTI with OneHotDegree(88) features (89 wide)."""
    torch.manual_seed(4)
    g = torch.Generator().manual_seed(8)
    n, e = 650, 6000
    ei = _graph(6, n, e)
    x = _rand(g, n, 89, dtype=dtype)
    layer = conv.SAGEConv(-1, 2048)
    with torch.no_grad():
        got = layer(x.cuda(), ei.cuda())
    assert layer.lin_l.weight.shape == (2048, 89) and layer.lin_l.weight.dtype == dtype
    P = _np_params(layer)
    _close(got, ora.sage_conv(_f64(x), ei.numpy(), P["lin_l.weight"], P["lin_l.bias"], P["lin_r.weight"]), dtype, "SAGE")
    layer = conv.SAGEConv(89, 64, normalize=True, root_weight=False).to(dtype).cuda()
    with torch.no_grad():
        got = layer(x.cuda(), ei.cuda())
    P = _np_params(layer)
    _close(got, ora.sage_conv(_f64(x), ei.numpy(), P["lin_l.weight"], P["lin_l.bias"], None, normalize=True), dtype, "SAGE normalised")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("relations", [1, 3])
def test_film_conv(conv, ora, dtype, relations):
    """benchmark_convs.py:146: FiLMConv(in_channels=11, out_channels=2048)."""
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(9)
    n, e, o = 400, 2500, 2048 if relations == 1 else 96
    ei = _graph(7, n, e)
    et = torch.randint(0, relations, (e,), generator=g)
    x = _rand(g, n, 11, dtype=dtype)
    layer = conv.FiLMConv(11, o, num_relations=relations).to(dtype).cuda()
    with torch.no_grad():
        got = layer(x.cuda(), ei.cuda(), et.cuda() if relations > 1 else None)
    P = _np_params(layer)
    want = ora.film_conv(_f64(x), ei.numpy(), [P[f"lins.{r}.weight"] for r in range(relations)],
                         [(P[f"films.{r}.weight"], P[f"films.{r}.bias"]) for r in range(relations)], P["lin_skip.weight"],
                         P["film_skip.weight"], et.numpy())
    _close(got, want, dtype, "FiLM")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("cfg", [dict(i=1, o=2048, towers=1, edge=None, divide=False),      # benchmark_convs.py:197-206 (MNIST, x is 1 wide)
                                 dict(i=16, o=64, towers=4, edge=None, divide=True),
                                 dict(i=8, o=32, towers=2, edge=3, divide=False)])
def test_pna_conv(conv, ora, dtype, cfg):
    torch.manual_seed(6)
    g = torch.Generator().manual_seed(10)
    n, e = 600, 4800
    ei = _graph(8, n, e)
    deg_hist = torch.bincount(torch.bincount(ei[1], minlength=n))
    aggr, scal = ["mean", "min", "max", "std"], ["identity", "amplification", "attenuation"]
    layer = conv.PNAConv(cfg["i"], cfg["o"], aggr, scal, deg_hist, edge_dim=cfg["edge"], towers=cfg["towers"],
                         divide_input=cfg["divide"]).to(dtype).cuda()
    x = _rand(g, n, cfg["i"], dtype=dtype)
    ea = _rand(g, e, cfg["edge"], dtype=dtype) if cfg["edge"] else None
    with torch.no_grad():
        got = layer(x.cuda(), ei.cuda(), None if ea is None else ea.cuda())
    P = _np_params(layer)
    T = cfg["towers"]
    want = ora.pna_conv(_f64(x), ei.numpy(), [(P[f"pre_nns.{t}.0.weight"], P[f"pre_nns.{t}.0.bias"]) for t in range(T)],
                        [(P[f"post_nns.{t}.0.weight"], P[f"post_nns.{t}.0.bias"]) for t in range(T)], (P["lin.weight"], P["lin.bias"]),
                        aggr, scal, layer.avg_deg, None if ea is None else _f64(ea),
                        (P["edge_encoder.weight"], P["edge_encoder.bias"]) if cfg["edge"] else None, towers=T, divide_input=cfg["divide"])
    _close(got, want, dtype, "PNA")


@pytest.mark.parametrize("pre,post,edge", [(2, 1, None), (1, 3, None), (3, 2, 3)])
def test_pna_conv_deeper_pre_and_post_networks(conv, ora, pre, post, edge):
    """torch_geometric PNAConv(pre_layers, post_layers): Linear then (ReLU, Linear) per extra layer (/root/reference
    ops.txt lists the layer; benchmark_convs.py:197-206 uses one of each). A deeper pre-MLP is not linear in [x_i, x_j]: its
    first layer is still split per node, the rest runs on per-edge rows — against the per-edge float64 restatement."""
    torch.manual_seed(7)
    g = torch.Generator().manual_seed(11)
    n, e, T = 400, 3000, 2
    ei = _graph(9, n, e)
    deg_hist = torch.bincount(torch.bincount(ei[1], minlength=n))
    aggr, scal = ["mean", "min", "max", "std"], ["identity", "amplification", "attenuation"]
    layer = conv.PNAConv(8, 24, aggr, scal, deg_hist, edge_dim=edge, towers=T, pre_layers=pre, post_layers=post).cuda()
    x = _rand(g, n, 8)
    ea = _rand(g, e, edge) if edge else None
    with torch.no_grad():
        got = layer(x.cuda(), ei.cuda(), None if ea is None else ea.cuda())
    P = _np_params(layer)
    pres = [[(P[f"pre_nns.{t}.{2 * li}.weight"], P[f"pre_nns.{t}.{2 * li}.bias"]) for li in range(pre)] for t in range(T)]
    posts = [[(P[f"post_nns.{t}.{2 * li}.weight"], P[f"post_nns.{t}.{2 * li}.bias"]) for li in range(post)] for t in range(T)]
    want = ora.pna_conv(_f64(x), ei.numpy(), pres, posts, (P["lin.weight"], P["lin.bias"]), aggr, scal, layer.avg_deg,
                        None if ea is None else _f64(ea), (P["edge_encoder.weight"], P["edge_encoder.bias"]) if edge else None, towers=T)
    assert got.shape == (n, 24)
    _close(got, want, torch.float32, f"PNA pre={pre} post={post}")


def test_layers_join_the_graph_outside_no_grad(conv):
    """groq_script.py:135-137 warms the model up OUTSIDE torch.no_grad(): every layer is differentiable
    (tests/test_conv_train_gpu.py), so such a call returns a result attached to the graph; under no_grad / with frozen
    parameters the fused inference pass runs and the result is detached."""
    layer = conv.CGConv(11, 0).half().cuda()
    x = torch.rand(29, 11, device="cuda").half()
    ei = torch.randint(0, 29, (2, 56), device="cuda")
    out = layer(x, ei)                     # grad mode on, parameters trainable
    assert out.shape == (29, 11) and out.requires_grad
    with torch.no_grad():
        assert not layer(x, ei).requires_grad
    pna = conv.PNAConv(4, 8, ["mean", "max"], ["identity"], torch.tensor([0, 3, 5, 2])).cuda()
    xp = torch.rand(29, 4, device="cuda")
    assert pna(xp, ei).requires_grad
    with torch.no_grad():
        fused = pna(xp, ei)
    assert not fused.requires_grad
    torch.testing.assert_close(pna(xp, ei).detach(), fused, rtol=1e-4, atol=1e-5)     # the train chain == the fused pass
    pna.requires_grad_(False)
    assert not pna(xp, ei).requires_grad   # frozen: the fused pass, grad mode on


def test_layers_under_inference_mode(conv, ora):
    """torch.inference_mode() is the normal GNN serving setting: tensors made inside it have no version counter, so neither the
    plan of edge_index nor the packed weights may be cached on them — and the result must still be right, call after call."""
    g = torch.Generator().manual_seed(12)
    ei = _graph(10, 60, 400)
    x = _rand(g, 60, 8)
    with torch.inference_mode():
        layer = conv.CGConv(8, 0).cuda()                 # parameters are inference tensors too
        xd, ed = x.cuda(), ei.cuda()
        outs = [layer(xd, ed) for _ in range(2)]
        layer.lin_f.weight.mul_(0.5)
        outs.append(layer(xd, ed))
    P = _np_params(layer)
    want = ora.cg_conv(_f64(x), ei.numpy(), P["lin_f.weight"], P["lin_f.bias"], P["lin_s.weight"], P["lin_s.bias"])
    _close(outs[2], want, torch.float32, "after an in-place update under inference_mode")
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])


def test_layer_weights_follow_parameter_updates(conv, ora):
    """The packed weight operand is cached: an in-place update, a load_state_dict and a dtype change must all be seen."""
    torch.manual_seed(9)
    layer = conv.CGConv(8, 0).cuda()
    g = torch.Generator().manual_seed(11)
    ei = _graph(9, 50, 300)
    x = _rand(g, 50, 8)

    def check(dtype):
        with torch.no_grad():
            got = layer(x.to(dtype).cuda(), ei.cuda())
        P = _np_params(layer)
        want = ora.cg_conv(_f64(x.to(dtype)), ei.numpy(), P["lin_f.weight"], P["lin_f.bias"], P["lin_s.weight"], P["lin_s.bias"])
        _close(got, want, dtype, "after update")

    check(torch.float32)
    with torch.no_grad():
        layer.lin_f.weight.mul_(-1.5)
    check(torch.float32)
    layer.load_state_dict({k: torch.randn_like(v) * 0.3 for k, v in layer.state_dict().items()})
    check(torch.float32)
    layer.half()
    check(torch.float16)


def test_cgconv_tuple_then_tensor_input_on_one_layer(conv, ora):
    """the fused [p | q] packing (4K columns, x_src is x_dst) and the destination-only packing (2K columns,
    bipartite call) must not share a cache — a (x_a, x_b) call followed by a plain-tensor call used to reuse the [c, 2K]
    weight and return an [N, 0] result without an error. Both orders, same layer."""
    torch.manual_seed(12)
    layer = conv.CGConv(16, 0).cuda()
    g = torch.Generator().manual_seed(13)
    n, e = 120, 900
    ei = _graph(10, n, e)
    xa, xb = _rand(g, n, 16), _rand(g, n, 16)
    P = _np_params(layer)

    def want_bip():
        src, dst = ei.numpy()
        z = np.concatenate([_f64(xb)[dst], _f64(xa)[src]], -1)
        m = ora.sigmoid(z @ P["lin_f.weight"].T + P["lin_f.bias"]) * ora.softplus(z @ P["lin_s.weight"].T + P["lin_s.bias"])
        return ora.scatter(m, dst, n, "sum") + _f64(xb)

    want_one = ora.cg_conv(_f64(xa), ei.numpy(), P["lin_f.weight"], P["lin_f.bias"], P["lin_s.weight"], P["lin_s.bias"])
    with torch.no_grad():
        for _ in range(2):      # tuple -> tensor -> tuple -> tensor
            got = layer((xa.cuda(), xb.cuda()), ei.cuda())
            assert got.shape == (n, 16)
            _close(got, want_bip(), torch.float32, "CGConv bipartite call")
            got = layer(xa.cuda(), ei.cuda())
            assert got.shape == (n, 16)
            _close(got, want_one, torch.float32, "CGConv plain call after a bipartite one")


def test_fused_layer_equals_the_unfused_chain_at_scale(conv):
    """E = 5M, N = 1M, D = 64 fp16 (too big for the float64 oracle): the fused CGConv equals the same layer computed the
    propagate way from this package's own index_select / addmm / scatter_add kernels, within fp16 rounding of the chain."""
    import gnnops

    torch.manual_seed(10)
    n, e, d = 1_000_000, 5_000_000, 64
    g = torch.Generator(device="cuda").manual_seed(12)
    ei = torch.randint(0, n, (2, e), generator=g, device="cuda")
    x = (torch.rand(n, d, generator=g, device="cuda") - 0.5).half()
    layer = conv.CGConv(d, 0).half().cuda()
    with torch.no_grad():
        fused = layer(x, ei)
        z = torch.cat([gnnops.index_select(x, 0, ei[1]), gnnops.index_select(x, 0, ei[0])], dim=1)
        wf = layer.lin_f.weight.t().contiguous()
        ws = layer.lin_s.weight.t().contiguous()
        m = torch.sigmoid(gnnops.addmm(layer.lin_f.bias, z, wf)).float() * torch.nn.functional.softplus(gnnops.addmm(layer.lin_s.bias, z, ws).float())
        chain = gnnops.scatter_add(m, ei[1], dim=0, dim_size=n) + x.float()
    err = (fused.float() - chain).abs().max().item() / chain.abs().max().item()
    assert err < 4e-3, err
