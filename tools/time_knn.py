"""torch_cluster.knn, D = 3, fp32: the one-pass kernel (k <= 64) against the k-round kernel (GNNOPS_KNN_ROUNDS=1) with a batch
vector, and — one cloud, no batch vector — the grid form against the one-pass kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch, gnnops
from torch_cluster import knn
def timed(x, y, k, bx, by, iters=3):
    for _ in range(1): out = knn(x, y, k, bx, by)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): out = knn(x, y, k, bx, by)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters, out
from gnnops import spatial
for n, nb in ((32768, 32), (100000, 1), (400000, 1)):     # 32 clouds of 1024 points (PointNet++-style batches); one large cloud
    x = torch.rand(n, 3, device="cuda")
    b = (torch.arange(n, device="cuda") * nb // n)
    for k in (6, 16, 40):
        os.environ.pop("GNNOPS_KNN_ROUNDS", None)
        spatial._KNN_GRID_MIN_POINTS = 1 << 40            # the exhaustive kernels, also for one large cloud under a batch vector
        t1, o1 = timed(x, x, k, b, b)
        tg = None
        if nb == 1:
            spatial._KNN_GRID_MIN_POINTS = 8192
            tg, og = timed(x, x, k, None, None, iters=10)
            spatial._KNN_GRID_MIN_POINTS = 1 << 40
            assert torch.equal(og, o1)
        os.environ["GNNOPS_KNN_ROUNDS"] = "1"
        t0, o0 = timed(x, x, k, b, b)
        os.environ.pop("GNNOPS_KNN_ROUNDS", None)
        spatial._KNN_GRID_MIN_POINTS = 8192
        print(f"n={n:7d} clouds={nb:3d} k={k:3d}: k rounds {t0:9.3f} ms   one pass {t1:9.3f} ms   ({t0 / t1:5.1f}x)  equal={torch.equal(o0, o1)}" + (f"   grid {tg:8.3f} ms ({t1 / tg:6.1f}x the one pass)" if tg else ""), flush=True)
for n in (1_000_000, 4_000_000):
    x = torch.rand(n, 3, device="cuda")
    tg, og = timed(x, x, 16, None, None, iters=3)
    print(f"n={n:8d} one cloud k= 16: grid {tg:9.3f} ms", flush=True)
