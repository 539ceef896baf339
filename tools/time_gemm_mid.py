"""addmm at mid sizes that take the 128 x 128 LDS-DMA kernel (aligned: interior tiles only), bf16."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch

import gnnops

g = torch.Generator(device="cuda").manual_seed(1)
for L in [int(a) for a in sys.argv[1:]] or [1024, 1536, 2048, 2560, 2816]:
    a, b, c = [(torch.rand(L, L, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16) for _ in range(3)]
    for _ in range(40):
        gnnops.addmm(c, a, b)
    best = 1e9
    for rnd in range(4):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            out = gnnops.addmm(c, a, b)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 50)
    err = (out.float() - torch.addmm(c, a, b).float()).abs().max().item()
    print(f"bf16 L={L:5d} {best:.4f} ms {2 * L ** 3 / best / 1e9:7.1f} TFLOP/s maxdiff {err:.3f}", flush=True)
