"""rocprofv3 target: fp16 addmm at a spread of the reference's sweep lengths (benchmark_native_addmm.py:23-27), NCALL calls
per length with a marker kernel (a 1-element fill of a size-L tensor) between lengths so the trace can be cut per length.
usage: prof_gemm_sweep.py [L ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops
NCALL = 6
sizes = [int(a) for a in sys.argv[1:]] or [1581, 2527, 3205, 3763, 4249, 4684, 5082, 5797, 6433, 7011, 7546, 8045, 8164]
for L in sizes:
    g = torch.Generator(device="cuda").manual_seed(1)
    a, b, c = [(torch.rand(L, L, generator=g, device="cuda") * 2 - 1).half() for _ in range(3)]
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        o = gnnops.addmm(c, a, b)
    torch.cuda.synchronize()
    s.record()
    for _ in range(NCALL):
        o = gnnops.addmm(c, a, b)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / NCALL
    print(f"L={L:5d} {ms * 1e3:8.1f} us/call {2 * L ** 3 / ms / 1e9:7.1f} TFLOP/s", flush=True)
    del a, b, c, o
