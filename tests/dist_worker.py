"""Worker for test_dist_cpu.py: one gloo rank of the destination-partitioned scatter (gnnops.dist).
The HIP op cannot run on CPU, so the local reduction is injected: the oracle (test infrastructure)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def oracle_local_scatter(src, index, dim, out, dim_size, reduce):
    from oracle import oracle

    res = oracle.scatter(src.numpy(), index.numpy(), dim=dim, dim_size=dim_size, reduce=reduce)
    if isinstance(res, tuple):
        return torch.from_numpy(res[0]), torch.from_numpy(res[1])
    return torch.from_numpy(res)


def make_inputs(rank, world, n_total, e_local, d):
    g = torch.Generator().manual_seed(100 + rank)
    src = torch.rand(e_local, d, generator=g) * 2 - 1
    idx = torch.randint(0, n_total, (e_local,), generator=g)
    idx[idx == 5] = 6  # global destination 5 receives nothing from anyone
    return src, idx


def run(rank, world, init_file, n_total, e_local, d, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        from gnnops.dist import owned_rows, sharded_scatter

        src, idx = make_inputs(rank, world, n_total, e_local, d)
        res = {}
        for r in ("sum", "min", "max", "mean"):
            res[r] = sharded_scatter(src, idx, n_total, r, local_scatter=oracle_local_scatter).numpy()
        lo, hi = owned_rows(n_total, rank, world)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, **res)
    finally:
        dist.destroy_process_group()
