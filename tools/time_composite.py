"""Time the SURVEY 8(f) widening ops at the config-2 shape (N=10M, E=50M, D=128 fp32), plan reused."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops, torch_scatter
N, E, D = 10_000_000, 50_000_000, 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(42)
src = torch.rand(E, D, generator=g, device=dev)
idx = torch.randint(0, N, (E,), generator=g, device=dev)
plan = gnnops.Plan(idx, N)
def t(name, fn, bytes_, iters=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:34s} {ms:8.3f} ms  {bytes_/ms/1e6:8.1f} GB/s (compulsory bytes)", flush=True)
srcb, outb = E * D * 4, N * D * 4
t("scatter_softmax (plan)", lambda: torch_scatter.scatter_softmax(src, plan, 0), 2 * srcb + E * 4)
t("scatter_log_softmax (plan)", lambda: torch_scatter.scatter_log_softmax(src, plan, 0), 2 * srcb + E * 4)
t("scatter_logsumexp (plan)", lambda: torch_scatter.scatter_logsumexp(src, plan, 0), srcb + outb + E * 4)
t("scatter_std (plan)", lambda: torch_scatter.scatter_std(src, plan, 0), srcb + outb + E * 4)
sidx = idx.sort().values
indptr = gnnops.rowptr_from_sorted(sidx, N)
t("segment_csr sum", lambda: torch_scatter.segment_csr(src, indptr), srcb + outb)
t("segment_coo sum (sorted index)", lambda: torch_scatter.segment_coo(src, sidx, dim_size=N), srcb + outb + E * 8)
