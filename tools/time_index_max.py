"""index_max (the dim_size = index.max() + 1 of a scatter called without dim_size) on the reference's (6708)^2 int64 index."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch

import gnnops
from gnnops import _lib
from gnnops.ops import _stream

L = _lib.load()
for n in (6708 * 6708, 50_000_000, 1_000_000):
    idx = torch.randint(0, 6708, (n,), device="cuda")
    out = torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(5):
        L.gnnops_index_max(idx.data_ptr(), n, out.data_ptr(), _stream())
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        L.gnnops_index_max(idx.data_ptr(), n, out.data_ptr(), _stream())
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 50
    assert int(out.item()) == int(idx.max().item())
    print(f"index_max n={n:9d}  {ms * 1e3:8.1f} us  {n * 8 / ms / 1e6:7.1f} GB/s", flush=True)
