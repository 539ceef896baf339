"""Harness helpers the op scripts star-import (`from graph_benchmark.benchmark.util import *`).

Counterpart of the reference's graph_benchmark/benchmark/util.py:11-61: every public name and call signature it
offers is offered here, so an unchanged script resolves them. Written for ROCm: memory figures are fractions of
the device's real capacity (the reference divides by a hard-coded 40 GB board), and the device helpers are
no-ops rather than errors when torch was built without a device runtime.
"""
import random as _random

import numpy as _np
import torch as _torch

__all__ = [
    "setup_seed", "print_util_info", "get_reserved_in_mb", "combine_vals", "setup_cuda", "empty_cache",
    "print_sparsity_info", "print_bm_stats", "print_input_dims",
]

_MB = 1e6


def _have_device():
    return _torch.cuda.is_available()


def _capacity_bytes():
    props = _torch.cuda.get_device_properties(_torch.cuda.current_device())
    return float(props.total_memory)


# ---- reproducibility -------------------------------------------------------------------------------------
def setup_seed(seed):
    """One seed for python's, numpy's and torch's generators (host and all visible devices)."""
    _random.seed(seed)
    _np.random.seed(seed)
    _torch.manual_seed(seed)
    if _have_device():
        _torch.cuda.manual_seed_all(seed)
    _torch.backends.cudnn.deterministic = True


# ---- device bookkeeping ----------------------------------------------------------------------------------
def setup_cuda():
    """Return the device string the scripts pass to tensor factories; benchmarking needs a device."""
    if _have_device():
        return "cuda"
    raise Exception("Benchmarking only supported for CUDA")


def empty_cache():
    if _have_device():
        _torch.cuda.empty_cache()


def get_reserved_in_mb():
    return _torch.cuda.memory_reserved() / _MB


def print_util_info():
    total = _capacity_bytes()
    lines = (
        "GPU INFO:",
        f"\t Memory allocated:  {_torch.cuda.memory_allocated() / total}",
        f"\t Memory reserved:  {_torch.cuda.memory_reserved() / total}",
    )
    print("\n".join(lines))


# ---- reporting -------------------------------------------------------------------------------------------
def combine_vals(bm_val, bm_val_native):
    """'ours (native)' cell of the CSVs."""
    return "{} ({})".format(bm_val, bm_val_native)


def print_sparsity_info(sparsity, input, verbose=True):
    if verbose:
        nonzero_fraction = _torch.count_nonzero(input) / input.numel()
        print(f"Sparsity info: {sparsity}, percent non-zero is: {nonzero_fraction}")


def print_bm_stats(m0, verbose=True):
    if verbose:
        summary = f"median is {m0.median}, iqr is {m0.iqr}, count is {len(m0.times)}"
        print("Benchmark blocked autorange stats: " + summary)


def print_input_dims(tshape):
    print("DEBUG: Current input has dims {}".format(len(tshape)))
