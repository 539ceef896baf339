"""gnnops_edge_reduce on one graph of BASELINE config 2's size (N = 10M, E = 50M, uniform endpoints): lane width sweep
(GNNOPS_EDGE_VEC = elements per lane; fewer = more lanes per row, down to one row per wave)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd")]
import torch
import gnnops
from gnnops import conv

gnnops.load_library()
n, e = 10_000_000, 50_000_000
g = torch.Generator(device="cuda").manual_seed(5)
ei = torch.randint(0, n, (2, e), generator=g, device="cuda")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / reps


cases = []
for dt, name in ((torch.float16, "fp16"), (torch.float32, "fp32")):
    d = 128
    pq = torch.empty(n, 4 * d, dtype=dt, device="cuda").normal_()
    x = torch.empty(n, d, dtype=dt, device="cuda").normal_()
    es = pq.element_size()
    gb = (e * (2 * d * es + 8) + n * (2 * d * es + 2 * d * es) + 4 * (n + 1)) / 1e9
    cases.append((f"cgconv D={d} {name}", lambda pq=pq, x=x, d=d: conv.edge_reduce("cgconv", pq[:, 2 * d:], ei, n, p=pq[:, :2 * d], add=x), gb))
    if dt == torch.float16:
        cases.append((f"film   D={d} {name}", lambda pq=pq, x=x, d=d: conv.edge_reduce("film", pq[:, 2 * d:3 * d], ei, n, p=pq[:, :2 * d], add=x, aggr=("mean",)),
                      (e * (d * es + 8) + n * (2 * d * es + 2 * d * es) + 4 * (n + 1)) / 1e9))
        cases.append((f"copy   D={d} {name} (sum; = spmm without values)", lambda x=x: conv.edge_reduce("copy", x, ei, n, add=x),
                      (e * (d * es + 8) + n * (2 * d * es) + 4 * (n + 1)) / 1e9))
        d2 = 64
        aggr, scal = ("mean", "min", "max", "std"), ("identity", "amplification", "attenuation")
        cases.append((f"add    D={d2} {name} x (4 aggregators, 3 scalers)",
                      lambda pq=pq, d2=d2: conv.edge_reduce("add", pq[:, d2:2 * d2], ei, n, p=pq[:, :d2], aggr=aggr, scalers=scal, avg_deg={"log": 1.7, "lin": 5.0}),
                      (e * (d2 * es + 8) + n * (d2 * es + 12 * d2 * es) + 4 * (n + 1)) / 1e9))
    for label, fn, gb in cases:
        for v in ("", "8", "4", "2", "1"):
            os.environ.pop("GNNOPS_EDGE_VEC", None)
            if v:
                os.environ["GNNOPS_EDGE_VEC"] = v
            ms = timed(fn)
            print(f"{label:52s} lane width {v or 'policy':>6s}: {ms:8.3f} ms  {gb / ms * 1e3:7.1f} GB/s alg  ({gb / ms * 1e3 / 80:4.1f} % of 8 TB/s)", flush=True)
    cases = []
    del pq, x
