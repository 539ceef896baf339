"""Shared helpers for the tests: torch <-> numpy views the oracle understands, golden loading."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TORCH_DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
NAME_OF = {v: k for k, v in TORCH_DT.items()}


def to_np(t):
    """CPU/GPU tensor -> numpy array; bf16 travels as uint16 bit patterns."""
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def from_np(a, dname):
    """numpy array (uint16 bits for bf16) -> CPU tensor of the named dtype."""
    a = np.ascontiguousarray(a)
    if dname == "bf16":
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
    return torch.from_numpy(a.copy())


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def assert_bits_equal(got, exp, what=""):
    got = np.ascontiguousarray(got)
    exp = np.ascontiguousarray(exp)
    assert got.shape == exp.shape, f"{what}: shape {got.shape} != {exp.shape}"
    assert got.dtype.itemsize == exp.dtype.itemsize, f"{what}: dtype {got.dtype} vs {exp.dtype}"
    gb = got.view(f"u{got.dtype.itemsize}") if got.dtype.kind != "i" else got
    eb = exp.view(f"u{exp.dtype.itemsize}") if exp.dtype.kind != "i" else exp
    if got.dtype.kind == "f":  # +0 / -0 compare equal as values; NaN bits must match positions
        same = (got == exp) | (np.isnan(got) & np.isnan(exp))
    else:
        same = gb == eb
    if not same.all():
        bad = np.argwhere(~same)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {got.size} elements differ; first at {i}: got {got[i]!r} expected {exp[i]!r}")


def f32_of(a, dname):
    """Widen a numpy array of the named dtype to float32."""
    if dname == "bf16":
        return (a.astype(np.uint32) << 16).view(np.float32)
    return a.astype(np.float32)
