"""CPU suite, part 2: the C-ABI library loads without a GPU and exports every symbol include/gnnops.h
declares; host-side logic (argument checks, shape decomposition, harness mirror) behaves like the
reference's call sites expect. No compute calls here."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gnnops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnnops_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import gnnops
    from gnnops import _lib

    names = _declared_symbols()
    assert len(names) >= 13
    lib = gnnops.load_library()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gnnops.h but not exported by libgnnops.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in gnnops/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.gnnops_version() == 2
    assert lib.gnnops_last_error() is not None


def test_workspace_queries_are_host_only():
    import gnnops

    lib = gnnops.load_library()
    e, n = 50_000_000, 10_000_000
    ws = lib.gnnops_plan_workspace_bytes(e, n)
    assert 3 * 4 * e <= ws <= 3 * 4 * e + 16 * (1 << 20)
    assert lib.gnnops_plan_workspace_bytes(0, 0) > 0
    assert lib.gnnops_scatter_elementwise_workspace_bytes(1, 100, 8, 0, 0) == 0      # f32 sum: in place
    assert lib.gnnops_scatter_elementwise_workspace_bytes(1, 100, 8, 1, 0) >= 3200   # f16 sum: fp32 scratch
    assert lib.gnnops_scatter_elementwise_workspace_bytes(1, 100, 8, 1, 1) >= 6400   # f16 mean: + counts
    assert lib.gnnops_fused_select_sum_workspace_bytes() > 0
    # 16-bit addmm (csrc/gemm.hip gemm_plan): nothing for whole K-tiles and 16-B rows in whole rounds of tiles; side copies
    # of the last K-tile otherwise; 256 KiB per CU when the last round of 256 x 256 tiles is cut along K; and a problem
    # taller than one grid is planned slab by slab (the largest need of the slab heights that occur)
    ws = lib.gnnops_addmm_workspace_bytes
    assert ws(4096, 4096, 4096) == 0
    assert 0 < ws(4096, 4096, 4100) < (8 << 20)
    assert ws(4352, 4352, 4096) >= 256 * (256 << 10)          # 289 tiles: 33 left over
    slab = 65280 * 128
    assert ws(slab + 777, 512, 264) >= max(ws(slab, 512, 264), ws(777, 512, 264))
    assert ws(2 * slab + 4352, 4352, 264) >= ws(4352, 4352, 264) >= 256 * (256 << 10)


def test_cpu_tensors_are_refused_not_emulated():
    """The product has no CPU path: CPU tensors raise instead of running an eager fallback."""
    import gnnops
    import torch_scatter

    src = torch.rand(8, 4)
    idx = torch.randint(0, 3, (8,))
    for fn in (gnnops.scatter_add, torch_scatter.scatter_min, torch_scatter.scatter_mean):
        with pytest.raises(RuntimeError, match="no CPU path"):
            fn(src, idx, 0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        gnnops.index_select(src, 0, idx)
    with pytest.raises(RuntimeError, match="no CPU path"):
        gnnops.Plan(idx, 3)


def test_shape_decomposition_and_index_layout_detection():
    from gnnops import ops

    assert ops._bek((6, 5, 4), 0) == (1, 6, 20)
    assert ops._bek((6, 5, 4), 1) == (6, 5, 4)
    assert ops._bek((6, 5, 4), 2) == (30, 4, 1)
    src = torch.zeros(6, 5)
    row = torch.arange(6)
    assert ops._row_index_of(row, src, 0) is row
    exp = row.view(-1, 1).expand(6, 5)
    assert torch.equal(ops._row_index_of(exp, src, 0), row)
    assert ops._row_index_of(torch.zeros(6, 5, dtype=torch.int64), src, 0) is None  # genuine full index
    col = torch.arange(5)
    assert torch.equal(ops._row_index_of(col.view(1, -1).expand(6, 5), src, 1), col)
    assert ops._broadcast_index(row, src, 0).shape == src.shape
    with pytest.raises(IndexError):
        ops._norm_dim(2, 2, "x")
    assert ops._norm_dim(-1, 2, "x") == 1


def test_torch_scatter_shim_surface():
    """Names and signatures the reference imports (benchmark_scatter_add.py:5-7 etc.)."""
    import inspect

    import torch_scatter

    for name in ("scatter", "scatter_add", "scatter_sum", "scatter_mean", "scatter_min", "scatter_max", "scatter_mul"):
        fn = getattr(torch_scatter, name)
        params = list(inspect.signature(fn).parameters)
        assert params[:5] == ["src", "index", "dim", "out", "dim_size"], (name, params)
    assert inspect.signature(torch_scatter.scatter).parameters["dim"].default == -1
    assert inspect.signature(torch_scatter.scatter).parameters["reduce"].default == "sum"


def test_harness_mirror(tmp_path, capsys):
    """util / DataWriter counterparts keep the reference's observable behaviour
    (graph_benchmark/benchmark/util.py:11-61, DataWriter.py:5-36; expected strings from SURVEY.md §8c)."""
    import pandas as pd

    from graph_benchmark.benchmark import util
    from graph_benchmark.benchmark.DataWriter import DataWriter

    util.setup_seed(42)
    a = torch.rand(3)
    util.setup_seed(42)
    assert torch.equal(a, torch.rand(3))
    assert util.combine_vals(1.5, 0.25) == "1.5 (0.25)"
    with pytest.raises(Exception, match="Benchmarking only supported for CUDA"):
        if not torch.cuda.is_available():
            util.setup_cuda()
        else:
            raise Exception("Benchmarking only supported for CUDA")
    util.print_input_dims((3, 3))
    util.print_sparsity_info(0.5, torch.tensor([0.0, 1.0]))
    out = capsys.readouterr().out
    assert "DEBUG: Current input has dims 2" in out and "Sparsity info: 0.5" in out
    for name in ("setup_seed", "print_util_info", "get_reserved_in_mb", "combine_vals", "setup_cuda", "empty_cache",
                 "print_sparsity_info", "print_bm_stats", "print_input_dims"):
        assert callable(getattr(util, name))

    dw = DataWriter("x", "p")
    dw.add_entry(["1", "0", "True"], (3, 3), 0, 1.5)
    dw.write_data(str(tmp_path))
    df = pd.read_csv(tmp_path / "x.csv", index_col=0)
    assert list(df.columns) == ["p", "Input size (>95% mem util)*", "Sparsity", "GPU clock time"]
    assert df.iloc[0].tolist() == ["1;0;True", "(3, 3)", 0, 1.5]


def test_product_never_touches_the_oracle_or_a_cpu_fallback():
    """The oracle is test infrastructure: no module of the shipped package may import or load it, and no product source
    may mention a CPU fallback path (static check over gnn-ops-benchmark_amd/)."""
    pkg = os.path.join(ROOT, "gnn-ops-benchmark_amd")
    offenders = []
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".hip", ".h")):
                continue
            text = open(os.path.join(dirpath, f), errors="replace").read()
            if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "liboracle" in text or "oracle/_build" in text:
                offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    # the only users outside tests/: smoke() and bench.py's cpu_baseline leg
    for name, needle in (("__graft_entry__.py", "def smoke"), ("bench.py", "def cpu_baseline_leg")):
        text = open(os.path.join(ROOT, name)).read()
        first_use = text.index("from oracle import oracle")
        assert first_use > text.index(needle), f"{name}: oracle imported outside {needle}"


def test_raw_ops_never_drop_a_gradient_silently():
    """The guard every raw entry point runs first (gnnops/ops.py _refuse_grad): an operand that requires grad raises while
    grad mode is on — outside a torch.autograd.Function nothing may hand back a tensor cut off from the graph — and passes
    under no_grad (where the Functions' forward / backward run)."""
    import torch

    from gnnops import ops

    t = torch.rand(4, 3, requires_grad=True)
    with pytest.raises(NotImplementedError):
        ops._refuse_grad("scatter", t)
    with torch.no_grad():
        ops._refuse_grad("scatter", t)
    ops._refuse_grad("scatter", t.detach(), None)
    # the package-level names are the differentiable front ends
    import gnnops
    from gnnops import autograd

    assert gnnops.scatter is autograd.scatter and gnnops.index_select is autograd.index_select
    assert gnnops.addmm is autograd.addmm


def test_runner_writes_the_reference_csv_schemas():
    """a18: every one of the reference's 17 scripts has a spec whose CSV path and header row equal what the reference's
    script (or its DataWriter) writes — fixture tests/golden/csv_headers.json, generated from the reference by
    tests/golden/make_csv_headers.py — and a same-named entry script."""
    import importlib.util
    import json

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    scripts = os.path.join(root, "gnn-ops-benchmark_amd", "op_bm_scripts")
    spec = importlib.util.spec_from_file_location("benchmark_ops", os.path.join(scripts, "benchmark_ops.py"))
    bo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bo)
    fixture = json.load(open(os.path.join(here, "golden", "csv_headers.json")))
    assert sorted(bo.SPECS) == sorted(fixture) and len(fixture) == 17
    for op, info in fixture.items():
        path, cols = bo.spec_header(op)
        assert path == info["csv"], op
        assert cols == info["columns"], op
        assert bo.SPECS[op].script == info["script"], op
        assert os.path.exists(os.path.join(scripts, info["script"])), op
    # sweeps: the loops of the reference scripts (counts from their range literals)
    counts = {op: sum(1 for _ in bo.SPECS[op].sweep(bo.SPECS[op].num)) for op in bo.SPECS}
    assert counts["scatter_add"] == 4 * 100 * 2 and counts["scatter_min"] == 800      # RF x lengths x dim
    assert counts["native_index_select"] == 10 * 2 * 4 and counts["fused_index_select_reduce"] == 100 * 2 * 4
    assert counts["native_index_add_"] == 10 and counts["native_gather"] == 20 and counts["fused_index_add_reduce"] == 200
    assert counts["native_sort"] == 4 * (1 + 2 + 3) * 2 and counts["scatter_multiply"] == 4 * 4 * 2
    assert counts["sparse_coalesce"] == 3 * 2 * 4 and counts["sparse_spmm"] == 10 and counts["sparse_transpose"] == 10
    assert counts["native_addmm"] == 100 and counts["native_matmul"] == 100
    first = next(iter(bo.SPECS["scatter_add"].sweep(100)))
    assert first["src_dims"] == (223, 223)                      # int(sqrt(50_000)): the reference's first length
    assert bo.lengths(7_500_000, 200_000_000, 10)[-1] == 14142   # benchmark_native_index_select.py:38-39
    # the DataWriter row format the sort spec relies on
    from graph_benchmark.benchmark.DataWriter import DataWriter
    import tempfile

    dw = DataWriter(op_name="native_sort", param_names=fixture["native_sort"]["columns"][0])
    dw.add_entry(params_lst=["1", "0", "True"], tshape=(3,), sparsity=0, bm_val=1.5)
    with tempfile.TemporaryDirectory() as tmp:
        dw.write_data(path=tmp)
        lines = open(os.path.join(tmp, "native_sort.csv")).read().splitlines()
    import csv

    assert next(csv.reader([lines[0]]))[1:] == fixture["native_sort"]["columns"]
    assert lines[1] == fixture["native_sort"]["datawriter_row"]
