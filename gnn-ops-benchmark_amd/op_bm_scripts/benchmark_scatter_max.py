#!/usr/bin/env python3
"""`python op_bm_scripts/benchmark_scatter_max.py` — the reference script of the same name (its sweep, its timer protocol, its CSV
`mem_prof_data/scatter_max_small.csv`), run by the spec of benchmark_ops.py on the gfx950 kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import benchmark_ops

if __name__ == "__main__":
    benchmark_ops.run_script("scatter_max")
