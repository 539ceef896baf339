"""Time gnnops_plan_build at the config-2 shape (and check it against torch's stable sort)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops
N, E = (int(a) for a in sys.argv[1:3]) if len(sys.argv) >= 3 else (10_000_000, 50_000_000)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(42)
idx = torch.randint(0, N, (E,), generator=g, device=dev)
for _ in range(2):
    p = gnnops.Plan(idx, N)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    p = gnnops.Plan(idx, N)
e.record(); torch.cuda.synchronize()
ok = torch.equal(p.perm.long(), torch.sort(idx, stable=True).indices)
print(f"plan_build N={N} E={E}: {s.elapsed_time(e)/10:.4f} ms  matches torch stable argsort: {ok}")
