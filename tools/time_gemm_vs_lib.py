"""bf16 / fp16 addmm: ours (gemm.hip) against the vendor library behind torch.addmm on the same box, warmed, 20 calls each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd")]
import torch
import gnnops

gnnops.load_library()


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


g = torch.Generator(device="cuda").manual_seed(1)
for dt in (torch.bfloat16, torch.float16):
    for M, N, K in ((8192, 8192, 8192), (8164, 8164, 8164), (4096, 4096, 4096), (16384, 16384, 4096), (2_000_000, 256, 256), (5000, 5000, 5000)):
        a = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(dt)
        b = (torch.rand(K, N, generator=g, device="cuda") * 2 - 1).to(dt)
        c = (torch.rand(M, N, generator=g, device="cuda") * 2 - 1).to(dt)
        ours = timed(lambda: gnnops.addmm(c, a, b))
        lib = timed(lambda: torch.addmm(c, a, b))
        fl = 2.0 * M * N * K
        print(f"{str(dt):15s} {M:8d} x {N:6d} x {K:6d}   ours {ours:8.3f} ms {fl / ours / 1e9:8.1f} TF   library {lib:8.3f} ms {fl / lib / 1e9:8.1f} TF   ours/library time {ours / lib:5.2f}", flush=True)
        del a, b, c
