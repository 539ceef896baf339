"""Same-box A/B of the two item -> workgroup maps of the config-2 kernels whose OUTPUT rows are consecutive
(round 3; tools/micro/store_sweep.hip says contiguous-chunk stores beat grid-strided ones in a plain copy / fill / mix):
  seg_rows_kernel   (plan-path segment reduce)     GNNOPS_SEG_MAP  = g (grid-strided destinations) | b (256 consecutive rows per workgroup)
  select_rows_kernel (pull index_select)           GNNOPS_PULL_MAP = g | b
Boxes differ by several per cent, so the two maps alternate inside one process. usage: python tools/ab_store_map.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops
from gnnops import _lib, ops

N, E, D = 10_000_000, 50_000_000, 128
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(42)
src = torch.rand(E, D, generator=g, device=dev)
idx = torch.randint(0, N, (E,), generator=g, device=dev)
lib = gnnops.load_library()
gnnops.set_plan_cache(False)
plan = gnnops.Plan(idx, N)
out = torch.empty(N, D, device=dev)
table = torch.rand(N, D, generator=g, device=dev)


def ev(fn, it=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def seg():
    _lib.check(lib.gnnops_segment_reduce(src.data_ptr(), plan.rowptr.data_ptr(), plan.perm.data_ptr(), out.data_ptr(), None, 1, E, D, N,
                                         _lib.F32, _lib.SUM, 0, ops._stream()), "seg")


def seg_min():
    arg = torch.empty(N, D, dtype=torch.int64, device=dev)
    _lib.check(lib.gnnops_segment_reduce(src.data_ptr(), plan.rowptr.data_ptr(), plan.perm.data_ptr(), out.data_ptr(), arg.data_ptr(), 1, E, D, N,
                                         _lib.F32, _lib.MIN, 0, ops._stream()), "seg")


ops._PUSH_MIN_TABLE_BYTES = 1 << 62
sel_out = [None]


def pull():
    sel_out[0] = None
    sel_out[0] = gnnops.index_select(table, 0, idx)


alg = E * D * 4 + E * 8 + N * D * 4
for rnd in range(4):
    for name, var, fn, b in (("seg_rows sum", "GNNOPS_SEG_MAP", seg, alg), ("seg_rows min+arg", "GNNOPS_SEG_MAP", seg_min, alg + N * D * 8),
                            ("pull index_select", "GNNOPS_PULL_MAP", pull, alg)):
        res = {}
        for m in ("g", "b"):
            os.environ[var] = m
            res[m] = ev(fn)
        os.environ.pop(var)
        print(f"round {rnd}  {name:20s} grid-strided {res['g']:7.3f} ms ({b / res['g'] / 1e6:6.0f} GB/s)   contiguous {res['b']:7.3f} ms "
              f"({b / res['b'] / 1e6:6.0f} GB/s)   contiguous/strided time {res['b'] / res['g']:.3f}", flush=True)
