"""Does setting hubs aside cost anything when there are none? Plain vs hub-aware entries, config-2 shape, sum and min."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops
from gnnops import _lib
from gnnops.ops import _stream

L = gnnops.load_library()
N, E, D = 10_000_000, 50_000_000, 128
g = torch.Generator(device="cuda").manual_seed(42)
src = torch.rand(E, D, generator=g, device="cuda")
idx = torch.randint(0, N, (E,), generator=g, device="cuda")
plan = gnnops.Plan(idx, N)
out = torch.empty(N, D, device="cuda")
arg = torch.empty(N, D, dtype=torch.int64, device="cuda")
ws_bytes = L.gnnops_bucket_workspace_bytes(E, N)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
_lib.check(L.gnnops_bucket_partition(idx.data_ptr(), E, N, ws.data_ptr(), ws_bytes, _stream()), "partition")


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for name, rc in (("sum", _lib.SUM), ("min", _lib.MIN)):
    a = arg.data_ptr() if rc == _lib.MIN else None
    hb = L.gnnops_hub_workspace_bytes(E, D, rc)
    hw = torch.empty(hb, dtype=torch.uint8, device="cuda")
    for rep in range(2):
        t1 = timed(lambda: L.gnnops_segment_reduce(src.data_ptr(), plan.rowptr.data_ptr(), plan.perm.data_ptr(), out.data_ptr(), a, 1, E, D, N, 0, rc, 0, _stream()))
        t2 = timed(lambda: L.gnnops_segment_reduce_hubs(src.data_ptr(), plan.rowptr.data_ptr(), plan.perm.data_ptr(), out.data_ptr(), a, 1, E, D, N, 0, rc, 0, hw.data_ptr(), hb, _stream()))
        t3 = timed(lambda: L.gnnops_bucket_reduce(src.data_ptr(), ws.data_ptr(), out.data_ptr(), a, E, D, N, 0, rc, 0, _stream()))
        t4 = timed(lambda: L.gnnops_bucket_reduce_hubs(src.data_ptr(), ws.data_ptr(), out.data_ptr(), a, E, D, N, 0, rc, 0, hw.data_ptr(), hb, _stream()))
        print(f"{name}: segment plain {t1:.3f}  hubs {t2:.3f} | bucket plain {t3:.3f}  hubs {t4:.3f} ms", flush=True)
