"""Harness helpers the op scripts star-import (`from graph_benchmark.benchmark.util import *`,
reference: graph_benchmark/benchmark/util.py:11-61). Same names and call signatures so an unchanged
script finds them; written for ROCm (memory fractions are of the device's real capacity rather than a
hard-coded 40 GB board).
"""
import random

import numpy as np
import torch


def setup_seed(seed):
    """Seed python, numpy and torch (host and every visible device)."""
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True


def _device_capacity():
    return torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory


def print_util_info():
    cap = _device_capacity()
    print("GPU INFO:")
    print("\t Memory allocated: ", torch.cuda.memory_allocated() / cap)
    print("\t Memory reserved: ", torch.cuda.memory_reserved() / cap)


def get_reserved_in_mb():
    return torch.cuda.memory_reserved() / 1e6


def combine_vals(bm_val, bm_val_native):
    return f"{bm_val} ({bm_val_native})"


def setup_cuda():
    if not torch.cuda.is_available():
        raise Exception("Benchmarking only supported for CUDA")
    return "cuda"


def empty_cache():
    torch.cuda.empty_cache()


def print_sparsity_info(sparsity, input, verbose=True):
    if not verbose:
        return
    frac = torch.count_nonzero(input) / input.numel()
    print(f"Sparsity info: {sparsity}, percent non-zero is: {frac}")


def print_bm_stats(m0, verbose=True):
    if not verbose:
        return
    print(f"Benchmark blocked autorange stats: median is {m0.median}, iqr is {m0.iqr}, count is {len(m0.times)}")


def print_input_dims(tshape):
    print(f"DEBUG: Current input has dims {len(tshape)}")
