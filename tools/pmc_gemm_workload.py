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
# round 3: 17 x 17 tiles (a last round of 33: the persistent split-K kernel) beside the plain grid on the same operands, and the
# reference's largest fp16 length (operands in place / A copied)
for L2, dt2 in ((4352, torch.bfloat16), (8164, torch.float16)):
    a2, b2, c2 = [(torch.rand(L2, L2, generator=g, device="cuda") * 2 - 1).to(dt2) for _ in range(3)]
    for sk in ("", "0"):
        if sk: os.environ["GNNOPS_GEMM_SK"] = sk
        else: os.environ.pop("GNNOPS_GEMM_SK", None)
        for _ in range(3):
            out = gnnops.addmm(c2, a2, b2)
    os.environ.pop("GNNOPS_GEMM_SK", None)
    del a2, b2, c2
a32, b32, c32 = [torch.rand(L, L, generator=g, device="cuda") * 2 - 1 for _ in range(3)]
for _ in range(3):
    out = gnnops.addmm(c32, a32, b32)   # gemm_f32_kernel
    ref = torch.addmm(c32, a32, b32)    # the library's fp32 kernel, for the same counters
torch.cuda.synchronize()
print("done")
