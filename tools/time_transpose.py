"""Dense transposes (gnnops.transpose_contiguous) at the reference's fp16 sweep lengths (benchmark_sparse_transpose.py) and the
big fp32 / int shapes of the dim-0 routes: us per call and GB/s of bytes read + written."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops
def timed(x, iters=10):
    for _ in range(2): y = gnnops.transpose_contiguous(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): y = gnnops.transpose_contiguous(x)
    e.record(); torch.cuda.synchronize()
    assert torch.equal(y[:64], x[:, :64].t())
    return s.elapsed_time(e) / iters * 1e3
for dt, shapes in ((torch.float16, [(1414, 1414), (3000, 3000), (5000, 5000), (7071, 7071), (8192, 8192), (20000, 20000)]),
                   (torch.float32, [(7071, 7071), (8192, 8192), (28200, 28200), (38000, 38000)])):
    for R, C in shapes:
        x = torch.rand(R, C, device="cuda").to(dt)
        line = f"{str(dt)[6:]:8s} ({R},{C})"
        for mode in (("0", "1", "2", "3", "0", "3") if dt == torch.float32 else ("",)):
            if mode: os.environ["GNNOPS_T32"] = mode
            us = timed(x, 10 if R * C < 2e8 else 3)
            line += f" | {('T32=' + mode) if mode else ''} {us:9.1f} us {2 * R * C * x.element_size() / us / 1e3:7.1f} GB/s"
        os.environ.pop("GNNOPS_T32", None)
        print(line, flush=True)
        del x
