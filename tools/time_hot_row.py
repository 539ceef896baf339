"""index_select with one hot row (a table row selected by a large share of the outputs): push form vs pull form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops
from gnnops import ops as _ops

N, E, D = 2_200_000, 7_000_000, 128
g = torch.Generator(device="cuda").manual_seed(0)
table = torch.rand(N, D, generator=g, device="cuda")


def timed(fn, iters=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for hot in (0, 100_000, 1_000_000):
    idx = torch.randint(0, N, (E,), generator=g, device="cuda")
    if hot:
        idx[torch.randperm(E, device="cuda")[:hot]] = 7
    res = []
    for cache in (False, True):
        gnnops.set_plan_cache(cache)
        res.append(timed(lambda: gnnops.index_select(table, 0, idx)))
    saved = _ops._PUSH_MIN_TABLE_BYTES
    _ops._PUSH_MIN_TABLE_BYTES = 1 << 62
    pull = timed(lambda: gnnops.index_select(table, 0, idx))
    _ops._PUSH_MIN_TABLE_BYTES = saved
    ok = torch.equal(gnnops.index_select(table, 0, idx), table[idx])
    print(f"hot-row selections {hot:>8d}: push one-shot {res[0]:7.3f} ms, push planned (cached) {res[1]:7.3f} ms, pull {pull:7.3f} ms  equal={ok}", flush=True)
