"""Workload for tools/pmc_gemm.sh: the three 16-bit GEMM kernels at 8192^3 bf16 (GNNOPS_GEMM_NO_DMA picks the kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops

L = 8192
g = torch.Generator(device="cuda").manual_seed(1)
a, b, c = [(torch.rand(L, L, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16) for _ in range(3)]
for sw in ("0", "3", "1"):  # 0 = default ladder (256 x 256 ping-pong at this size)
    os.environ["GNNOPS_GEMM_NO_DMA"] = sw
    for _ in range(3):
        out = gnnops.addmm(c, a, b)
os.environ["GNNOPS_GEMM_NO_DMA"] = "0"
a32, b32, c32 = [torch.rand(L, L, generator=g, device="cuda") * 2 - 1 for _ in range(3)]
for _ in range(3):
    out = gnnops.addmm(c32, a32, b32)   # gemm_f32_kernel
    ref = torch.addmm(c32, a32, b32)    # the library's fp32 kernel, for the same counters
torch.cuda.synchronize()
print("done")
