"""fp32 addmm (v_mfma_f32_16x16x4_f32: exact fp32 products) at square sizes up to the reference's (48000)^2."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch

import gnnops

g = torch.Generator(device="cuda").manual_seed(1)
for L in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 48000]:
    a, b, c = [torch.rand(L, L, generator=g, device="cuda") * 2 - 1 for _ in range(3)]
    for _ in range(2):
        out = gnnops.addmm(c, a, b)
    torch.cuda.synchronize()
    n = 10 if L <= 8192 else 2
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        out = gnnops.addmm(c, a, b)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    s.record()
    ref = torch.addmm(c, a, b)
    e.record()
    torch.cuda.synchronize()
    print(f"fp32 L={L:6d} {ms:10.3f} ms {2 * L ** 3 / ms / 1e9:7.1f} TFLOP/s   torch.addmm {s.elapsed_time(e):10.3f} ms   maxdiff {(out - ref).abs().max().item():.5f}", flush=True)
    del a, b, c, out, ref
