"""One-launch plan (csrc/plan.hip plan_small_kernel) against the radix build, per call, host + device (back to back on one
stream, synchronised once): where does the single workgroup stop paying?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd")]
import torch
import gnnops

gnnops.load_library()
gnnops.set_plan_cache(False)
for E, N in [(40, 21), (1000, 500), (4096, 2048), (8192, 4096), (18744, 9134), (32768, 16384), (49152, 20000), (65536, 6177), (65536, 40000)]:
    g = torch.Generator(device="cuda").manual_seed(E)
    idx = torch.randint(0, N, (E,), generator=g, device="cuda")
    comp = torch.randint(0, N, (E,), generator=g, device="cuda")
    row = f"E={E:6d} N={N:6d}"
    for name, env in (("one launch", "1"), ("radix", "0")):
        os.environ["GNNOPS_PLAN_SMALL"] = env
        for _ in range(5):
            gnnops.Plan(idx, N, comp)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(100):
            gnnops.Plan(idx, N, comp)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / 100 * 1e6
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(100):
            gnnops.Plan(idx, N, comp)
        e.record()
        torch.cuda.synchronize()
        row += f"   {name}: {wall:7.1f} us wall, {s.elapsed_time(e) * 10:7.1f} us stream"
    print(row, flush=True)
