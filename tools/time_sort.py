import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops
for shape, dim in [((7071, 7071), 1), ((7071, 7071), 0), ((20000, 20000), 1), ((20000, 20000), 0), ((800, 800, 800), 2), ((800, 800, 800), 0)]:
    x = torch.rand(shape, device="cuda")
    for _ in range(2):
        v, i = gnnops.sort(x, dim, stable=True); del v, i
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        v, i = gnnops.sort(x, dim, stable=True); del v, i
    e.record(); torch.cuda.synchronize()
    print(shape, dim, f"{s.elapsed_time(e)/3:.2f} ms", flush=True)
    del x
