"""Which part of a K-step does the fp32 256 x 256 kernel wait for? Timing-only builds (GNNOPS_GEMM_F32_DBG), 8192^3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops

L = 8192
g = torch.Generator(device="cuda").manual_seed(1)
a, b, c = [torch.rand(L, L, generator=g, device="cuda") * 2 - 1 for _ in range(3)]
for name, env in (("ping-pong", {}), ("lockstep", {"GNNOPS_GEMM_F32_BIG": "1"}), ("lockstep, no DMA in the loop", {"GNNOPS_GEMM_F32_DBG": "1"}),
                  ("lockstep, no fragment reads in the loop", {"GNNOPS_GEMM_F32_DBG": "2"}), ("lockstep, no barrier / wait", {"GNNOPS_GEMM_F32_DBG": "3"}),
                  ("lockstep, fragments software-pipelined", {"GNNOPS_GEMM_F32_DBG": "4"}),
                  ("lockstep, operands staged through registers", {"GNNOPS_GEMM_F32_DBG": "5"}),
                  ("four waves of 128 x 128, one per SIMD", {"GNNOPS_GEMM_F32_BIG": "4"}),
                  ("128 x 128 register-staged", {"GNNOPS_GEMM_F32_BIG": "0"})):
    for k in ("GNNOPS_GEMM_F32_BIG", "GNNOPS_GEMM_F32_DBG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(3):
        gnnops.addmm(c, a, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        gnnops.addmm(c, a, b)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"{name:42s} {ms:8.3f} ms  {2 * L ** 3 / ms / 1e9:7.1f} TFLOP/s", flush=True)
s.record()
for _ in range(10):
    torch.addmm(c, a, b)
e.record()
torch.cuda.synchronize()
print(f"{'torch.addmm (hipBLASLt)':42s} {s.elapsed_time(e) / 10:8.3f} ms  {2 * L ** 3 / (s.elapsed_time(e) / 10) / 1e9:7.1f} TFLOP/s")
