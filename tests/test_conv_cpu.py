"""CPU checks for the SURVEY.md §8(f) rank-4 widening (no GPU): the layer classes expose torch_geometric's parameter names and
shapes (so a state_dict moves between the two), the float64 oracle reproduces a hand-computed CGConv (the layer whose text the
reference holds, app_bm/groq_script.py:91-109) and its own algebra (per-edge Linear == split per-node products), and the
benchmark's synthetic batches have the datasets' shapes."""
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layer_parameter_names_and_shapes_follow_pyg():
    from gnnops import conv

    def shapes(m):
        return {k: tuple(v.shape) for k, v in m.state_dict().items()}

    # torch_geometric.nn.conv.CGConv (text: groq_script.py:58-79): lin_f / lin_s: Linear(sum(channels) + dim, channels[1])
    assert shapes(conv.CGConv(11, 0)) == {"lin_f.weight": (11, 22), "lin_f.bias": (11,), "lin_s.weight": (11, 22), "lin_s.bias": (11,)}
    assert shapes(conv.CGConv((24, 16), 3))["lin_s.weight"] == (16, 43)
    assert set(shapes(conv.CGConv(8, 0, batch_norm=True))) >= {"bn.weight", "bn.bias", "bn.running_mean", "bn.running_var"}
    # GINConv(nn): parameters live under nn.*; eps is a buffer unless train_eps
    assert shapes(conv.GINConv(torch.nn.Linear(11, 2048))) == {"eps": (1,), "nn.weight": (2048, 11), "nn.bias": (2048,)}
    # SAGEConv: lin_l (with bias) on the aggregate, lin_r (no bias) on the root
    assert shapes(conv.SAGEConv(89, 2048)) == {"lin_l.weight": (2048, 89), "lin_l.bias": (2048,), "lin_r.weight": (2048, 89)}
    # FiLMConv: lins.r (no bias), films.r: Linear(in, 2 out), lin_skip / film_skip (no bias)
    assert shapes(conv.FiLMConv(11, 2048)) == {"lins.0.weight": (2048, 11), "films.0.weight": (4096, 11), "films.0.bias": (4096,),
                                               "lin_skip.weight": (2048, 11), "film_skip.weight": (4096, 11)}
    # PNAConv: pre_nns.t.0 Linear(2 F, F), post_nns.t.0 Linear((A S + 1) F, out / towers), lin Linear(out, out)
    pna = conv.PNAConv(1, 2048, ["mean", "min", "max", "std"], ["identity", "amplification", "attenuation"], torch.tensor([0, 3, 5, 2]))
    assert shapes(pna) == {"pre_nns.0.0.weight": (1, 2), "pre_nns.0.0.bias": (1,), "post_nns.0.0.weight": (2048, 13),
                           "post_nns.0.0.bias": (2048,), "lin.weight": (2048, 2048), "lin.bias": (2048,)}
    assert pna.avg_deg["lin"] == pytest.approx((3 * 1 + 5 * 2 + 2 * 3) / 10) and pna.avg_deg["log"] == pytest.approx(
        (3 * np.log(2) + 5 * np.log(3) + 2 * np.log(4)) / 10)
    assert all(p.requires_grad for p in pna.parameters()) and all(p.requires_grad for p in conv.CGConv(4).parameters()), "every layer is trainable"
    deep = conv.PNAConv(4, 8, ["mean"], ["identity"], torch.tensor([1, 1]), pre_layers=2, post_layers=3)      # PyG's module layout
    assert set(shapes(deep)) >= {"pre_nns.0.0.weight", "pre_nns.0.2.weight", "post_nns.0.0.weight", "post_nns.0.2.weight", "post_nns.0.4.bias"}
    with pytest.raises(ValueError):
        conv.PNAConv(4, 8, ["mean"], ["identity"], torch.tensor([1, 1]), pre_layers=0)


def test_layers_refuse_cpu_tensors():
    """No CPU fallback: the product path fails loudly without the device."""
    from gnnops import conv

    layer = conv.CGConv(4)
    with pytest.raises((RuntimeError, ValueError, OSError)):
        layer(torch.rand(5, 4), torch.randint(0, 5, (2, 7)))


def test_oracle_cgconv_known_answer():
    """Two nodes, one edge 0 -> 1, one channel: x' = x + sigmoid(w_f . [x_1, x_0] + b_f) * softplus(w_s . [x_1, x_0] + b_s) at node 1
    (z = [x_i, x_j] with i the destination: groq_script.py:105-106), untouched at node 0."""
    from oracle import conv_oracle as o

    x = np.array([[2.0], [-1.0]])
    ei = np.array([[0], [1]])
    W_f, b_f, W_s, b_s = np.array([[0.5, -0.25]]), np.array([0.1]), np.array([[1.0, 2.0]]), np.array([-0.5])
    f = 0.5 * -1.0 + -0.25 * 2.0 + 0.1
    s = 1.0 * -1.0 + 2.0 * 2.0 - 0.5
    want1 = -1.0 + 1 / (1 + np.exp(-f)) * np.log1p(np.exp(s))
    got = o.cg_conv(x, ei, W_f, b_f, W_s, b_s)
    assert got[0, 0] == 2.0 and got[1, 0] == pytest.approx(want1, rel=1e-14)
    assert o.softplus(np.array([25.0]))[0] == 25.0                         # torch: above the threshold the input itself
    assert o.scatter(np.array([[1.0], [3.0]]), np.array([2, 2]), 4, "max").ravel().tolist() == [0, 0, 3, 0]   # empty groups -> 0


def test_oracle_per_edge_linear_equals_split_per_node_products():
    """The algebra the product relies on, checked inside the oracle's own arithmetic: Linear(cat[x_i, x_j, e]) per edge ==
    (x W_i^T + b)[dst] + (x W_j^T)[src] + e W_e^T."""
    from oracle import conv_oracle as o

    rng = np.random.default_rng(0)
    n, e, c, d = 40, 300, 6, 3
    x, ea = rng.normal(size=(n, c)), rng.normal(size=(e, d))
    ei = rng.integers(0, n, size=(2, e))
    W_f, b_f, W_s, b_s = rng.normal(size=(c, 2 * c + d)), rng.normal(size=c), rng.normal(size=(c, 2 * c + d)), rng.normal(size=c)
    want = o.cg_conv(x, ei, W_f, b_f, W_s, b_s, ea)
    src, dst = ei
    f = (x @ W_f[:, :c].T + b_f)[dst] + (x @ W_f[:, c:2 * c].T)[src] + ea @ W_f[:, 2 * c:].T
    s = (x @ W_s[:, :c].T + b_s)[dst] + (x @ W_s[:, c:2 * c].T)[src] + ea @ W_s[:, 2 * c:].T
    got = o.scatter(o.sigmoid(f) * o.softplus(s), dst, n, "sum") + x
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_synthetic_batches_have_the_datasets_shapes():
    spec = importlib.util.spec_from_file_location("benchmark_convs", os.path.join(ROOT, "gnn-ops-benchmark_amd", "app_bm", "benchmark_convs.py"))
    bc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bc)
    rng = np.random.default_rng(1)
    stats = {}
    for name, fn in bc._DATASETS.items():
        ns, es = [], []
        for _ in range(300 if name != "MNIST" else 5):
            n, ei, x = fn(rng)
            assert ei.shape[0] == 2 and ei.min() >= 0 and ei.max() < n and x.shape[0] == n
            ns.append(n), es.append(ei.shape[1])
        stats[name] = (np.mean(ns), np.mean(es), x.shape[1])
    assert 17 < stats["QM9"][0] < 19 and 34 < stats["QM9"][1] < 40 and stats["QM9"][2] == 11          # QM9: 18.0 nodes, 37.3 edges
    assert stats["MNIST"] == (75.0, 600.0, 1)                                                          # 75 superpixels, 8 neighbours
    assert 10 < stats["IMDB-MULTI"][0] < 16 and 100 < stats["IMDB-MULTI"][1] < 160 and stats["IMDB-MULTI"][2] == 89
