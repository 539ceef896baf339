"""Is the split-K hand-off of gemm_sk256_kernel safe when the SAME workspace addresses are reused with DIFFERENT operands?
Eager calls back to back and graph replays, fresh A every time, each result against the plain-grid kernel's (differences
beyond one rounding = a partner's partial tile from an earlier launch was read). Round 3: with the flags cleared by
hipMemsetAsync 3 of 8 graph replays failed (the memset NODE was not reliably seen by the next node's pollers; eager calls
never failed); cleared by a kernel of the library's own, none of 24."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops
M = N = 4352; K = 1024
gen = torch.Generator(device="cuda").manual_seed(1)
b = (torch.rand(K, N, generator=gen, device="cuda") - 0.5).half()
c = (torch.rand(M, N, generator=gen, device="cuda") - 0.5).half()
def fresh(): return (torch.rand(M, K, generator=gen, device="cuda") - 0.5).half()
def plain(a):
    os.environ["GNNOPS_GEMM_SK"] = "0"
    try: return gnnops.addmm(c, a, b)
    finally: del os.environ["GNNOPS_GEMM_SK"]
for order in sys.argv[1:] or ["3", "3", "3"]:
    os.environ["GNNOPS_GEMM_SK_ORDER"] = order
    eager = []
    for t in range(6):
        a = fresh()
        eager.append(round((gnnops.addmm(c, a, b).float() - plain(a).float()).abs().max().item(), 4))
    a = fresh()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): gnnops.addmm(c, a, b)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph): out = gnnops.addmm(c, a, b)
    rep = []
    for t in range(8):
        a.copy_(fresh())
        graph.replay(); torch.cuda.synchronize()
        rep.append(round((out.float() - plain(a).float()).abs().max().item(), 4))
    print(f"order {order:>3s}: eager max|sk - plain| {eager}   graph replays {rep}", flush=True)
    del graph, out
