"""Single-layer forward benchmarks — counterpart of the reference's app_bm/benchmark_convs.py:146-246 (FiLMConv, GINConv,
CGConv on QM9; PNAConv on MNIST superpixels; SAGEConv on IMDB-MULTI with OneHotDegree(88)), hidden width 2048, fp16
(`.to(torch.float16)`, :147) or fp32, batch size 512 (apps_bm_data/model_data_fp16.txt:2) or 1 (`Config.batch_size`, :26,
apps_bm_data/model_data_fp16_no_batching.txt).

Protocol as the reference's inference() (:50-77): 10 untimed calls, then every call bracketed by two device events and a
synchronize, mean over the calls; EVERY call sees a different batch (the reference's loader shuffles), so the destination
plan of a batch is built inside the timed call, never reused.

The datasets cannot be fetched here (the reference downloads them: QM9(root="/tmp/QM9"), :129). Batches are synthetic
with the published shape statistics of each dataset:
    QM9          x [n, 11], n in 3..29 (mean 18), molecular graphs: a random tree plus ring-closing bonds, both directions
                 (mean 37 directed edges)
    MNIST        x [n, 1], n = 75 superpixels, 8 nearest neighbours of random 2-D positions (600 directed edges)
    IMDB-MULTI   n in 7..89 (mean 13), dense ego networks (mean 132 directed edges), x = one-hot in-degree, 89 wide
and the timings are of THIS box; the A100 numbers printed beside them are the reference's own files, for orientation only.

  python benchmark_convs.py                       # fp16, batch 512, fused layers (gnnops.conv)
  python benchmark_convs.py --impl chain          # the same layers in propagate order on stock torch ops (what PyG runs)
  python benchmark_convs.py --batch-size 1 --dtype fp32 --n 300 --out conv.json
  python benchmark_convs.py --big                 # one C2-sized graph (N = 10M, E = 50M): the edge pass against its roofline
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))

A100_REFERENCE_S = {   # apps_bm_data/model_data_fp16.txt / model_data_fp32.txt (batch 512, seconds per call, torch_geometric.profile)
    "fp16": {"FiLMConv": 0.022828, "GIN": 0.003335, "CGConv": 0.003968, "PNA": 0.010827, "GraphSAGE": 0.004714},
    "fp32": {"FiLMConv": 0.021335, "GIN": 0.003017, "CGConv": 0.004380, "PNA": 0.011511},
}
A100_REFERENCE_NO_BATCHING_MS = {"FiLMConv": 0.4376, "GIN": 0.1880, "CGConv": 0.2983, "PNA": 1.0495, "GraphSAGE": 0.2632}   # model_data_fp16_no_batching.txt


# ---------------------------------------------------------------------------------------------------------------------
# synthetic batches with the datasets' shapes
# ---------------------------------------------------------------------------------------------------------------------
def _qm9_graph(rng):
    n = int(np.clip(round(rng.normal(18.0, 3.0)), 3, 29))
    parent = np.array([rng.integers(0, i) for i in range(1, n)])            # random tree: n - 1 bonds
    a, b = np.arange(1, n), parent
    extra = rng.integers(0, n, size=(max(int(round(rng.normal(1.6, 1.0))), 0), 2))
    extra = extra[extra[:, 0] != extra[:, 1]]
    a, b = np.concatenate([a, extra[:, 0]]), np.concatenate([b, extra[:, 1]])
    return n, np.stack([np.concatenate([a, b]), np.concatenate([b, a])]), rng.random((n, 11), dtype=np.float32)


def _mnist_graph(rng):
    n = 75
    pos = rng.random((n, 2))
    d = ((pos[:, None] - pos[None]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    nbr = np.argsort(d, axis=1)[:, :8]                                      # source = neighbour, destination = the node
    return n, np.stack([nbr.reshape(-1), np.repeat(np.arange(n), 8)]), rng.random((n, 1), dtype=np.float32)


def _imdb_graph(rng):
    n = int(np.clip(round(np.exp(rng.normal(2.35, 0.45))), 7, 89))
    p = min(1.0, 132.0 / (13.0 * 12.0) * (13.0 / n) ** 0.5)
    upper = np.triu(rng.random((n, n)) < p, 1)
    a, b = np.nonzero(upper)
    ei = np.stack([np.concatenate([a, b]), np.concatenate([b, a])])
    deg = np.minimum(np.bincount(ei[1], minlength=n), 88)
    return n, ei, np.eye(89, dtype=np.float32)[deg]                         # OneHotDegree(88)


_DATASETS = {"QM9": _qm9_graph, "MNIST": _mnist_graph, "IMDB-MULTI": _imdb_graph}


def make_batches(name, batch_size, count, dtype, seed):
    """`count` collated batches (x, edge_index) on the device, PyG-style: node ids of graph g offset by the nodes before it."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        xs, eis, off = [], [], 0
        for _ in range(batch_size):
            n, ei, x = _DATASETS[name](rng)
            xs.append(x)
            eis.append(ei + off)
            off += n
        out.append((torch.from_numpy(np.concatenate(xs)).to(dtype).cuda(), torch.from_numpy(np.concatenate(eis, axis=1)).long().cuda()))
    return out


def degree_histogram(batches):
    hist = torch.zeros(1, dtype=torch.long)
    for x, ei in batches:
        d = torch.bincount(torch.bincount(ei[1].cpu(), minlength=x.size(0)))
        if d.numel() > hist.numel():
            hist = torch.cat([hist, hist.new_zeros(d.numel() - hist.numel())])
        hist[: d.numel()] += d
    return hist


# ---------------------------------------------------------------------------------------------------------------------
# the same layers in propagate order on stock torch ops (gather per edge -> cat -> Linear per edge -> message -> scatter):
# the work torch_geometric's MessagePassing does, for an A/B on this box. Parameters are shared with the fused layer.
# ---------------------------------------------------------------------------------------------------------------------
def _scatter(m, index, n, reduce):
    out = m.new_zeros((n,) + m.shape[1:])
    if reduce in ("sum", "mean"):
        out.index_add_(0, index, m)
        if reduce == "mean":
            cnt = torch.bincount(index, minlength=n).clamp_(min=1).to(m.dtype)
            out /= cnt.view(n, *([1] * (m.dim() - 1)))
        return out
    return out.scatter_reduce_(0, index.view(-1, *([1] * (m.dim() - 1))).expand_as(m), m, "amin" if reduce == "min" else "amax", include_self=False)


def chain_forward(name, layer, x, ei):
    F = torch.nn.functional
    src, dst = ei[0], ei[1]
    n = x.size(0)
    if name == "GIN":
        return layer.nn(_scatter(x[src], dst, n, "sum") + (1.0 + float(layer.eps)) * x)
    if name == "GraphSAGE":
        return layer.lin_l(_scatter(x[src], dst, n, "mean")) + layer.lin_r(x)
    if name == "CGConv":
        z = torch.cat([x[dst], x[src]], dim=-1)
        return _scatter(layer.lin_f(z).sigmoid() * F.softplus(layer.lin_s(z)), dst, n, "sum") + x
    if name == "FiLMConv":
        o = layer.out_channels
        beta, gamma = layer.film_skip(x).split(o, dim=-1)
        out = torch.relu(gamma * layer.lin_skip(x) + beta)
        beta, gamma = layer.films[0](x).split(o, dim=-1)
        return out + _scatter(torch.relu(gamma[dst] * layer.lins[0](x)[src] + beta[dst]), dst, n, "mean")
    if name == "PNA":
        h = layer.pre_nns[0](torch.cat([x[dst], x[src]], dim=-1))
        mean = _scatter(h, dst, n, "mean")
        outs = [mean, _scatter(h, dst, n, "min"), _scatter(h, dst, n, "max"),
                (torch.relu(_scatter(h * h, dst, n, "mean") - mean * mean) + 1e-5).sqrt()]
        out = torch.cat(outs, dim=-1)
        deg = torch.bincount(dst, minlength=n).clamp_(min=1).to(x.dtype).view(-1, 1)
        amp, att = torch.log(deg + 1) / layer.avg_deg["log"], layer.avg_deg["log"] / torch.log(deg + 1)
        out = torch.cat([x, out, out * amp, out * att], dim=-1)
        return layer.lin(layer.post_nns[0](out))
    raise KeyError(name)


# ---------------------------------------------------------------------------------------------------------------------
def inference(fn, batches, n_timed, warmup=10):
    """benchmark_convs.py:50-77: untimed calls, then one (event, call, event, synchronize) per timed call."""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stats = []
    with torch.no_grad():
        for i in range(warmup + n_timed):
            x, ei = batches[i % len(batches)]
            if i < warmup:
                fn(x, ei)
                continue
            start.record()
            fn(x, ei)
            end.record()
            torch.cuda.synchronize()
            stats.append(start.elapsed_time(end))
    return np.array(stats)


def build_models(dtype, deg_hist):
    from gnnops import conv

    torch.manual_seed(0)
    aggr, scal = ["mean", "min", "max", "std"], ["identity", "amplification", "attenuation"]
    return [   # (name, dataset, layer) in the reference's order (benchmark_convs.py:146, 163, 180, 197, 231)
        ("FiLMConv", "QM9", conv.FiLMConv(in_channels=11, out_channels=2048).to(dtype).cuda()),
        ("GIN", "QM9", conv.GINConv(torch.nn.Linear(11, 2048)).to(dtype).cuda()),
        ("CGConv", "QM9", conv.CGConv(11, 0).to(dtype).cuda()),
        ("PNA", "MNIST", conv.PNAConv(in_channels=1, out_channels=2048, aggregators=aggr, scalers=scal, deg=deg_hist).to(dtype).cuda()),
        ("GraphSAGE", "IMDB-MULTI", conv.SAGEConv(89, 2048).to(dtype).cuda()),
    ]


def run(args):
    import gnnops

    gnnops.load_library()
    dtype = {"fp16": torch.float16, "fp32": torch.float32, "bf16": torch.bfloat16}[args.dtype]
    data = {name: make_batches(name, args.batch_size, args.batches, dtype, seed=17 + i) for i, name in enumerate(_DATASETS)}
    models = build_models(dtype, degree_histogram(data["MNIST"]))
    results = []
    for name, ds, layer in models:
        x0, e0 = data[ds][0]
        row = {"model": name, "dataset": ds, "dtype": args.dtype, "batch_size": args.batch_size, "nodes": x0.size(0), "edges": e0.size(1)}
        impls = ["fused", "chain"] if args.impl == "both" else [args.impl]
        for impl in impls:
            fn = layer if impl == "fused" else (lambda x, ei, _n=name, _l=layer: chain_forward(_n, _l, x, ei))
            stats = inference(fn, data[ds], args.n)
            row[f"{impl}_ms_mean"] = float(stats.mean())
            row[f"{impl}_ms_std"] = float(stats.std())
            row[f"{impl}_ms_median"] = float(np.median(stats))
        if len(impls) == 2:
            with torch.no_grad():
                a, b = layer(x0, e0).float(), chain_forward(name, layer, x0, e0).float()
            row["fused_vs_chain_max_rel"] = float((a - b).abs().max() / b.abs().max())
        ref = A100_REFERENCE_S.get(args.dtype, {}).get(name) if args.batch_size == 512 else \
            (A100_REFERENCE_NO_BATCHING_MS.get(name) if args.batch_size == 1 and args.dtype == "fp16" else None)
        row["reference_a100_ms"] = None if ref is None else (ref * 1e3 if args.batch_size == 512 else ref)
        results.append(row)
        print(f"Statistics for model {name} and dataset {ds}")          # benchmark_convs.py:149-150
        print(f"\t{row[impls[0] + '_ms_mean']}")
        extra = "".join(f"  {k}={v:.4g}" for k, v in row.items() if isinstance(v, float) and k not in (impls[0] + "_ms_mean",))
        print(f"\t[{impls[0]} ms/call on this box; batch of {row['nodes']} nodes, {row['edges']} edges]{extra}")
        print()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)
    return results


def run_big(args):
    """One graph of BASELINE config 2's size (N = 10M nodes, E = 50M edges, uniform random endpoints): CGConv D = 128 and the
    PNA aggregation set D = 64, fp16 — the edge pass against the bytes it must move."""
    import gnnops
    from gnnops import conv

    gnnops.load_library()
    n, e = args.big_nodes, args.big_edges
    g = torch.Generator(device="cuda").manual_seed(5)
    ei = torch.randint(0, n, (2, e), generator=g, device="cuda")
    out = []

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        t.record()
        torch.cuda.synchronize()
        return s.elapsed_time(t) / reps

    with torch.no_grad():
        d = 128
        x = (torch.rand(n, d, generator=g, device="cuda") - 0.5).half()
        layer = conv.CGConv(d, 0).half().cuda()
        ms_layer = timed(lambda: layer(x, ei))
        pq = torch.empty(n, 4 * d, dtype=torch.float16, device="cuda").normal_()
        ms_edge = timed(lambda: conv.edge_reduce("cgconv", pq[:, 2 * d:], ei, n, p=pq[:, :2 * d], add=x))
        gb = (e * (2 * d * 2 + 8) + n * (2 * d * 2 + d * 2 + d * 2) + 4 * (n + 1)) / 1e9
        out.append({"what": f"CGConv({d}) fp16, N={n} E={e}: whole layer (plan cached)", "ms": ms_layer})
        out.append({"what": "  its edge pass (gnnops_edge_reduce cgconv)", "ms": ms_edge, "algorithmic_GB": gb, "GBps": gb / ms_edge * 1e3,
                    "pct_of_hbm_peak": gb / ms_edge * 1e3 / 8000 * 100})
        z = None
        try:   # the propagate-order chain on stock ops (needs ~ E * 4 D * 2 B * several)
            ms_chain = timed(lambda: chain_forward("CGConv", layer, x, ei), reps=2)
            out.append({"what": "  the same layer in propagate order on stock torch ops", "ms": ms_chain})
        except torch.OutOfMemoryError:
            out.append({"what": "  the same layer in propagate order on stock torch ops", "ms": None, "note": "out of memory"})
        del z, pq
        d = 64
        pq = torch.empty(n, 2 * d, dtype=torch.float16, device="cuda").normal_()
        aggr, scal = ("mean", "min", "max", "std"), ("identity", "amplification", "attenuation")
        ms_pna = timed(lambda: conv.edge_reduce("add", pq[:, d:], ei, n, p=pq[:, :d], aggr=aggr, scalers=scal, avg_deg={"log": 1.7, "lin": 5.0}))
        gb = (e * (d * 2 + 8) + n * (d * 2 + 12 * d * 2) + 4 * (n + 1)) / 1e9
        out.append({"what": f"PNA aggregation set (4 aggregators x 3 scalers) D={d} fp16, one pass", "ms": ms_pna, "algorithmic_GB": gb,
                    "GBps": gb / ms_pna * 1e3, "pct_of_hbm_peak": gb / ms_pna * 1e3 / 8000 * 100})
    for r in out:
        print(json.dumps(r))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32", "bf16"])
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--n", type=int, default=300, help="timed calls per layer (Config.n, benchmark_convs.py:25)")
    ap.add_argument("--batches", type=int, default=24, help="distinct synthetic batches cycled through")
    ap.add_argument("--impl", default="fused", choices=["fused", "chain", "both"])
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--big-nodes", type=int, default=10_000_000)
    ap.add_argument("--big-edges", type=int, default=50_000_000)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise Exception("Benchmarking only supported for CUDA")
    run_big(args) if args.big else run(args)


if __name__ == "__main__":
    main()
