"""Workload for tools/pmc_gemm_f32.sh: the fp32 GEMM kernels at 8192^3 and the library's kernel, same operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator(device="cuda").manual_seed(1)
a32, b32, c32 = [torch.rand(L, L, generator=g, device="cuda") * 2 - 1 for _ in range(3)]
for sw in ("", "1", "0"):   # default (ping-pong 256 x 256), lockstep 256 x 256, 128 x 128 register-staged
    if sw:
        os.environ["GNNOPS_GEMM_F32_BIG"] = sw
    else:
        os.environ.pop("GNNOPS_GEMM_F32_BIG", None)
    for _ in range(4):
        out = gnnops.addmm(c32, a32, b32)
os.environ.pop("GNNOPS_GEMM_F32_BIG", None)
for _ in range(4):
    ref = torch.addmm(c32, a32, b32)
torch.cuda.synchronize()
print("done")
