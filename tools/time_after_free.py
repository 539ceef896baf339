"""Does a random-row gather slow down on memory the driver hands out AFTER a large free? bench.py's `layers` leg measured
inside the full run (after the config-2 legs freed ~35 GB with torch.cuda.empty_cache()) was 1.5x slower than in a fresh
process. Variants: fresh process | after allocating and freeing 35 GB through empty_cache() | the same without empty_cache()
(the caching allocator keeps the blocks and carves the new tensors out of them)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops
import bench

gnnops.load_library()
mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
if mode != "fresh":
    a = torch.empty(25_600_000_000 // 4, device="cuda")
    b = torch.empty(5_120_000_000 // 4, device="cuda")
    c = torch.empty(5_120_000_000 // 4, device="cuda")
    a.fill_(1.0); b.fill_(1.0); c.fill_(1.0)
    torch.cuda.synchronize()
    del a, b, c
    if mode == "empty_cache":
        torch.cuda.empty_cache()
r = bench.layers_leg(torch, gnnops)
print(mode, "gather-sum", r["gather_sum128_fp16_edge_pass"]["ms"], "ms   cgconv edge pass", r["cgconv128_fp16_edge_pass"]["ms"], "ms", flush=True)
