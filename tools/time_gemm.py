"""Time gnnops.addmm at a few square sizes: 16-bit operands with the LDS-DMA fast path on and off (A/B in one process,
interleaved), then fp32. Prints ms, TFLOP/s and the largest difference from torch.addmm."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch

import gnnops


def timed(a, b, c, iters):
    for _ in range(3):
        out = gnnops.addmm(c, a, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        out = gnnops.addmm(c, a, b)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters, (out.float() - torch.addmm(c, a, b).float()).abs().max().item()


def operands(L, dt):
    g = torch.Generator(device="cuda").manual_seed(1)
    return [(torch.rand(L, L, generator=g, device="cuda") * 2 - 1).to(dt) for _ in range(3)]


sizes = [int(a) for a in sys.argv[1:]] or [1581, 4096, 8164, 8192]
for dt in (torch.bfloat16, torch.float16):
    for L in sizes:
        a, b, c = operands(L, dt)
        for no_dma in ("0", "3", "1", "0"):
            os.environ["GNNOPS_GEMM_NO_DMA"] = no_dma
            ms, err = timed(a, b, c, 10)
            print(f"{str(dt):16s} no_dma={no_dma} L={L:6d} {ms:8.3f} ms  {2 * L ** 3 / ms / 1e9:8.1f} TFLOP/s  maxdiff_vs_torch={err:.4f}", flush=True)
os.environ["GNNOPS_GEMM_NO_DMA"] = "0"
for L in (4096, 8192):
    a, b, c = operands(L, torch.float32)
    ms, err = timed(a, b, c, 5)
    print(f"float32          L={L:6d} {ms:8.3f} ms  {2 * L ** 3 / ms / 1e9:8.1f} TFLOP/s  maxdiff_vs_torch={err:.5f}", flush=True)
