"""Where the split-K tail's time goes (timing only — orders 7 and 11 leave the tail tiles wrong): plain grid, the
persistent kernel, the same without the hand-off (order 7), the same without the tail at all (order 11)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops
os.environ["GNNOPS_GEMM_SK_TIMING_ONLY"] = "1"   # orders 7 and 11 are honoured only with this set
def timed(a, b, iters=10):
    for _ in range(2): out = gnnops.matmul(a, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): out = gnnops.matmul(a, b)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for L in [int(x) for x in sys.argv[1:]] or [4352, 5888, 7168]:   # multiples of 256: no pad copies; 17^2, 23^2, 28^2 tiles
    g = torch.Generator(device="cuda").manual_seed(1)
    a, b = [(torch.rand(L, L, generator=g, device="cuda") * 2 - 1).half() for _ in range(2)]
    line = f"L={L} tiles={(L // 256) ** 2}"
    for rep in range(2):
        for sk, od, name in (("0", "3", "plain"), ("1", "3", "split"), ("1", "7", "no-handoff"), ("1", "11", "no-tail")):
            os.environ["GNNOPS_GEMM_SK"] = sk; os.environ["GNNOPS_GEMM_SK_ORDER"] = od
            line += f" | {name} {timed(a, b):7.1f}"
    print(line, flush=True)
