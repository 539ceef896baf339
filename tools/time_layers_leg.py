"""bench.py's `layers` leg in a fresh process (nothing allocated or freed before it) — against the same leg inside a full
bench.py run, where it follows the config 2 / 3 / 4 legs."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT]
import torch
import gnnops
import bench

gnnops.load_library()
print(json.dumps(bench.layers_leg(torch, gnnops)))
