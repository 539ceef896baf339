"""Row sorts (on-chip, sort_rows.hip) and the tile transpose at the reference's shapes: timing after/before load-phase changes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd")]
import torch
import gnnops

gnnops.load_library()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


g = torch.Generator(device="cuda").manual_seed(0)
for shape, dim in (((20000, 20000), 1), ((20000, 20000), 0), ((7071, 7071), 1), ((800, 800, 800), 2), ((28200, 28200), 1)):
    x = torch.rand(*shape, generator=g, device="cuda")
    ms = timed(lambda: gnnops.sort(x, dim, False, True), 3)
    st = timed(lambda: torch.sort(x, dim=dim, stable=True), 2)
    print(f"sort {str(shape):20s} dim {dim}: {ms:9.3f} ms   (stock torch {st:9.3f} ms)", flush=True)
    del x
for L, dt in ((7071, torch.float16), (38000, torch.float32), (38000, torch.float16), (16384, torch.float32)):
    x = torch.rand(L, L, generator=g, device="cuda").to(dt)
    ms = timed(lambda: gnnops.transpose_contiguous(x), 5)
    st = timed(lambda: x.t().contiguous(), 5)
    gb = 2 * L * L * x.element_size() / 1e9
    print(f"transpose ({L})^2 {str(dt):14s}: {ms:8.3f} ms  {gb / ms * 1e3:7.1f} GB/s   (stock torch {st:8.3f} ms)", flush=True)
    del x
