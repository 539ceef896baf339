"""GPU suite: bench.py's own plumbing, as fresh subprocesses on a tiny workload.

  * the one-GPU line carries the contract's keys plus `roofline` and `cpu_baseline`;
  * `python bench.py --gpus 2` WITHOUT a launcher starts its ranks itself (torch.distributed.run children, before the
    parent touches a GPU) — rehearsed here with both ranks on cuda:0 over gloo (GNNOPS_BENCH_GLOO_ONE_GPU=1: a one-GPU
    box cannot hold two RCCL ranks); the line is labelled a rehearsal and carries the N>1 legs (same_work.c2_share,
    other_cuts, spmm_src_partitioned) and, from rank 0, `roofline` and `cpu_baseline` too.
Timings of these runs mean nothing; the JSON shape and the absence of errors are what is checked."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def _run(args, extra_env=None, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=timeout)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def _no_errors(obj, path=""):
    if isinstance(obj, dict):
        assert "error" not in obj, (path, obj["error"])
        for k, v in obj.items():
            _no_errors(v, path + "/" + k)


@pytest.mark.timeout(900)
def test_one_gpu_line_has_the_contract_keys_roofline_and_cpu_baseline():
    d = _run(["--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-extra-ops"])
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "GB/s" and d["dtype"] == "f32"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["stream_mix_ceiling"]["GBps"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    _no_errors(d)


@pytest.mark.timeout(1200)
def test_self_launched_two_rank_run_prints_one_line():
    d = _run(["--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1"], {"GNNOPS_BENCH_GLOO_ONE_GPU": "1"}, timeout=1100)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and "rehearsal" in d and d["value"] > 0 and d["scaling"] == "weak"
    assert "edge cut" in d["config"]["workload"]
    assert d["same_work"]["c2_share"]["value"] > 0
    assert d["other_cuts"]["cut_0"]["value"] > 0 and d["other_cuts"]["uniform_random_graph"]["dense_reduce_scatter"]["value"] > 0
    assert d["spmm_src_partitioned"]["ms_per_step"] > 0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0
    assert d["cpu_baseline"]["value"] > 0
    _no_errors(d)
