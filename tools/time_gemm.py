"""Time gnnops.addmm (bf16/fp16) at a few square sizes; prints ms and TFLOP/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8164, 8192]
for dt in (torch.bfloat16, torch.float16):
  for L in sizes:
   for no_dma in ("0", "1", "0", "1"):
    os.environ["GNNOPS_GEMM_NO_DMA"] = no_dma
    if True:
        g = torch.Generator(device="cuda").manual_seed(1)
        a = (torch.rand(L, L, generator=g, device="cuda") * 2 - 1).to(dt)
        b = (torch.rand(L, L, generator=g, device="cuda") * 2 - 1).to(dt)
        c = (torch.rand(L, L, generator=g, device="cuda") * 2 - 1).to(dt)
        for _ in range(3):
            out = gnnops.addmm(c, a, b)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        n = 10
        for _ in range(n):
            out = gnnops.addmm(c, a, b)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        ref = torch.addmm(c, a, b)
        err = (out.float() - ref.float()).abs().max().item()
        print(f"{str(dt):16s} no_dma={no_dma} L={L:6d} {ms:8.3f} ms  {2*L**3/ms/1e9:8.1f} TFLOP/s  maxdiff_vs_torch={err:.4f}", flush=True)
os.environ["GNNOPS_GEMM_NO_DMA"] = "0"
for L in (4096, 8192):
    a = torch.rand(L, L, device="cuda") * 2 - 1; b = torch.rand(L, L, device="cuda") * 2 - 1; c = torch.rand(L, L, device="cuda")
    for _ in range(2): out = gnnops.addmm(c, a, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): out = gnnops.addmm(c, a, b)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"float32 L={L} {ms:8.3f} ms {2*L**3/ms/1e9:8.1f} TFLOP/s maxdiff_vs_torch={(out-torch.addmm(c,a,b)).abs().max().item():.5f}", flush=True)
