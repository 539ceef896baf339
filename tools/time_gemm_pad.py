"""16-bit addmm: whole padded copies of both operands (GNNOPS_GEMM_PAD=full), of A only (a), of B only (b), or both
read in place with side copies of the last K-tile (none), alternating in one process; us per call. usage: time_gemm_pad.py [mode,mode,...] [L ...]   (modes: full a b none default)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops
def timed(a, b, c, iters=8):
    for _ in range(2): out = gnnops.addmm(c, a, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): out = gnnops.addmm(c, a, b)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3, out
for L in [int(x) for x in sys.argv[1:] if x.isdigit()] or [1581, 2527, 3205, 3763, 4249, 4684, 5082, 5797, 6433, 7011, 7546, 8045, 8164, 8168, 8192]:
    g = torch.Generator(device="cuda").manual_seed(1)
    a, b, c = [(torch.rand(L, L, generator=g, device="cuda") * 2 - 1).half() for _ in range(3)]
    line = f"L={L:5d}"
    outs = {}
    for mode in (sys.argv[1].split(",") * 2 if len(sys.argv) > 1 and not sys.argv[1].isdigit() else ["full", "a", "b", "none"] * 2):
        if mode == "default": os.environ.pop("GNNOPS_GEMM_PAD", None)
        else: os.environ["GNNOPS_GEMM_PAD"] = mode
        us, out = timed(a, b, c)
        outs[mode] = out
        line += f" | {mode:4s} {us:7.1f}"
    print(line + f" | equal={all(torch.equal(outs['full'], o) for o in outs.values())}", flush=True)
