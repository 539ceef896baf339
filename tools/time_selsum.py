import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops
E, D = 100_000_000, 128
table = torch.empty(E, D, device="cuda", dtype=torch.float16).uniform_(0, 1)
index = torch.randint(0, E, (E,), device="cuda")
for rif in ("4",):
    gnnops.index_select_sum(table, 0, index); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): gnnops.index_select_sum(table, 0, index)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 3
    print(f"rif={rif} {ms:.3f} ms {(E*8+E*D*2)/ms/1e6:.0f} GB/s", flush=True)
