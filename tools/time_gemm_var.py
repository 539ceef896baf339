"""A/B of gemm_dma256_kernel variants (GNNOPS_GEMM_VAR) in one process, interleaved, after the matrix clock has settled."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch

import gnnops

variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"]


def operands(M, N, K, dt):
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda r, c: (torch.rand(r, c, generator=g, device="cuda") * 2 - 1).to(dt)
    return mk(M, K), mk(K, N), mk(M, N)


for dt, (M, N, K) in ((torch.bfloat16, (8192, 8192, 8192)), (torch.float16, (8164, 8164, 8164)), (torch.bfloat16, (4096, 4096, 4096)),
                      (torch.bfloat16, (16384, 8192, 4096))):
    a, b, c = operands(M, N, K, dt)
    ref = torch.addmm(c, a, b).float()
    for _ in range(40):
        gnnops.addmm(c, a, b)
    res = {v: [] for v in variants}
    for rnd in range(4):
        for v in variants:
            os.environ["GNNOPS_GEMM_VAR"] = v
            for _ in range(5):
                out = gnnops.addmm(c, a, b)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                out = gnnops.addmm(c, a, b)
            e.record()
            torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) / 20)
            err = (out.float() - ref).abs().max().item()
            assert err < 0.6, (v, err)
    for v in variants:
        ms = min(res[v])
        print(f"{str(dt):15s} {M}x{N}x{K} var={v}  best {ms:.4f} ms  {2 * M * N * K / ms / 1e9:7.1f} TFLOP/s   all {['%.4f' % x for x in res[v]]}", flush=True)
