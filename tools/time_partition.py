"""Stage 1 of the one-shot scatter at config 2 (N=10M, E=50M): gnnops_bucket_partition alone, then the plan build and the
1-D sort that share the radix engine. usage (GPU box): python tools/time_partition.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch

import gnnops
from gnnops import _lib
from gnnops.ops import _stream


def ev(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    lib = gnnops.load_library()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(42)
    for N, E in ((10_000_000, 50_000_000), (10_000_000, 100_000_000)):
        index = torch.randint(0, N, (E,), generator=g, device=dev)
        ws_bytes = lib.gnnops_bucket_workspace_bytes(E, N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        part = ev(lambda: _lib.check(lib.gnnops_bucket_partition(index.data_ptr(), E, N, ws.data_ptr(), ws_bytes, _stream()), "p"))
        plan = ev(lambda: gnnops.Plan(index, N), 5)
        print(f"N={N} E={E}: bucket_partition {part * 1e3:.0f} us   plan build {plan * 1e3:.0f} us", flush=True)
        del ws, index
    x = torch.rand(50_000_000, generator=g, device=dev)
    print(f"sort 50M fp32 1-D: {ev(lambda: gnnops.sort(x), 5) * 1e3:.0f} us", flush=True)
    m = torch.rand(7071, 7071, generator=g, device=dev)
    print(f"sort (7071)^2 dim 1: {ev(lambda: gnnops.sort(m, 1), 5) * 1e3:.0f} us   dim 0: {ev(lambda: gnnops.sort(m, 0), 5) * 1e3:.0f} us", flush=True)


if __name__ == "__main__":
    main()
