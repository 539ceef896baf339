"""Kernel trace workload for the single-layer forward passes (app_bm/benchmark_convs.py): a few calls of each fused layer on
fresh batches, so the per-call launch list (plan build included) shows up in rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), os.path.join(ROOT, "gnn-ops-benchmark_amd", "app_bm")]
import torch
import benchmark_convs as bc
import gnnops

which = sys.argv[1] if len(sys.argv) > 1 else "CGConv"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 512
gnnops.load_library()
data = {name: bc.make_batches(name, bs, 12, torch.float16, seed=3) for name in bc._DATASETS}
models = bc.build_models(torch.float16, bc.degree_histogram(data["MNIST"]))
for name, ds, layer in models:
    if name != which:
        continue
    with torch.no_grad():
        for i in range(12):
            layer(*data[ds][i])
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    with torch.no_grad():
        for i in range(120):
            layer(*data[ds][i % 12])
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / 120 * 1e3:.4f} ms per call, back to back (host + device)")
