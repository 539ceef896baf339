"""Per-rank LOCAL work of gnnops.dist.sharded_scatter(exchange="sparse") at config 5's per-GPU share, on one GPU: this GPU
plays rank 1 of G (destinations below and above its range), the all-to-all is replaced by a stand-in of the same size
(what the peers would send back: as many (id, row) pairs as this rank sends). Prints the time of each piece."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops
from gnnops.dist import HipLocal

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cut = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
Nloc, E, D = 10_000_000, 50_000_000, 128
Ntot = Nloc * G
rank = 1
lo, hi = rank * Nloc, (rank + 1) * Nloc
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(43)
src = torch.rand(E, D, generator=gen, device=dev)
own = torch.randint(lo, hi, (E,), generator=gen, device=dev)
other = torch.randint(0, Ntot - Nloc, (E,), generator=gen, device=dev)
other += (other >= lo).to(torch.int64) * Nloc
index = torch.where(torch.rand(E, generator=gen, device=dev) < cut, other, own)
del own, other
gnnops.set_plan_cache(False)
local = HipLocal()


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


for rep in range(3):
    slab = torch.empty(Nloc, D, device=dev)
    torch.cuda.synchronize()
    t0 = ev()
    own_fn, ids, rows = local.split(src, index, Ntot, lo, hi, "sum", True)
    t1 = ev()
    own_fn(slab)
    t2 = ev()
    recv_ids = torch.randint(lo, hi, (ids.numel(),), generator=gen, device=dev)   # stand-in for the peers' shares
    recv_rows = rows
    torch.cuda.synchronize()
    t3 = ev()
    local.accumulate(slab, recv_rows, recv_ids - lo, "sum")
    t4 = ev()
    torch.cuda.synchronize()
    print(f"G={G} cut={cut}: remote rows {ids.numel()/1e6:.2f}M ({ids.numel()*(D*4+8)/1e9:.2f} GB on the wire) | "
          f"split {t0.elapsed_time(t1):.2f} ms, own slab {t1.elapsed_time(t2):.2f} ms, accumulate {t3.elapsed_time(t4):.2f} ms, "
          f"local total {t0.elapsed_time(t2) + t3.elapsed_time(t4):.2f} ms", flush=True)
    del slab, ids, rows, recv_ids, recv_rows, own_fn
