"""Which block tile wins below one 256 x 256 tile per CU? A/B via GNNOPS_GEMM_MIN256 (1 = always the big tile)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch

import gnnops

g = torch.Generator(device="cuda").manual_seed(1)
for L in [int(a) for a in sys.argv[1:]] or [1536, 2048, 2560, 3072, 3584, 4000, 4096, 5000, 6144]:
    a, b, c = [(torch.rand(L, L, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16) for _ in range(3)]
    for _ in range(40):
        gnnops.addmm(c, a, b)
    line = f"L={L:5d} tiles256={((L + 255) // 256) ** 2:4d}"
    for mn in ("1", "100000"):
        os.environ["GNNOPS_GEMM_MIN256"] = mn
        best = 1e9
        for rnd in range(3):
            for _ in range(5):
                gnnops.addmm(c, a, b)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                gnnops.addmm(c, a, b)
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 20)
        line += f"   {'256x256' if mn == '1' else '128x128'} {best:.4f} ms {2 * L ** 3 / best / 1e9:7.1f} TF"
    print(line, flush=True)
