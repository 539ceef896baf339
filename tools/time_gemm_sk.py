"""Split-K-tail A/B for 16-bit addmm (csrc/gemm.hip gemm_sk256_kernel): per length, GNNOPS_GEMM_SK = 0 (plain grid) and
default (persistent workgroups, last round cut along K), interleaved in one process; prints us per call, TFLOP/s and the
largest difference from the plain-grid result and from a float64 product on 64 sampled rows.
usage: time_gemm_sk.py [L ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops

def timed(a, b, c, iters=8):
    for _ in range(2):
        out = gnnops.addmm(c, a, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        out = gnnops.addmm(c, a, b)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3, out

sizes = [int(x) for x in sys.argv[1:]] or [3205, 3763, 4249, 4684, 5082, 5797, 6433, 7011, 7546, 8045, 8164, 8192]
dts = (torch.float16,) if os.environ.get("SK_DT", "") == "" else (torch.float16, torch.bfloat16)
for dt in dts:
    for L in sizes:
        g = torch.Generator(device="cuda").manual_seed(1)
        a, b, c = [(torch.rand(L, L, generator=g, device="cuda") * 2 - 1).to(dt) for _ in range(3)]
        rows = torch.randint(0, L, (64,), device="cuda")
        ref = (c[rows].double() + a[rows].double() @ b.double())
        base = None
        line = f"{str(dt)[6:]:9s} L={L:5d} tiles={(-(-L // 256)) ** 2:5d}"
        for sk in (os.environ.get("SK_MODES", "0,,0,,0,").split(",")):
            if sk: os.environ["GNNOPS_GEMM_SK"] = sk
            else: os.environ.pop("GNNOPS_GEMM_SK", None)
            us, out = timed(a, b, c)
            if base is None: base = out
            d0 = (out.float() - base.float()).abs().max().item()
            d64 = (out[rows].double() - ref).abs().max().item()
            line += f" | sk={sk or 'd'} {us:7.1f}us {2 * L ** 3 / us / 1e6:6.0f}TF d0={d0:.3g} d64={d64:.3g}"
        print(line, flush=True)
        del a, b, c, out, base
os.environ.pop("GNNOPS_GEMM_SK", None)
