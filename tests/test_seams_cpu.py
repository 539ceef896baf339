"""CPU suite: the two seams that are not ATen kernels (gnnops/jit.py, gnnops/aten.py _patch_contiguous).

The reference's "fused" scripts hand `@torch.jit.script` functions to the Timer (op_bm_scripts/
benchmark_fused_index_select_reduce.py:12-15, benchmark_fused_index_add_reduce.py:12-15); their bodies are restated here
in a real source file (TorchScript needs one) exactly as the reference writes them. The rewrite itself is host logic and
runs without a GPU (the custom operators have a CPU kernel that is the literal chain)."""
import pytest
import torch

from gnnops import jit


def ref_fused_index_select_reduce(input, dim: int, index):
    out = torch.index_select(input, dim, index).sum()
    return out


def ref_fused_index_add_reduce(input, dim: int, index, other):
    out = torch.index_add(input, dim, index, other)
    return torch.index_select(out, dim, index).sum(dim)


def near_miss_alpha(input, dim: int, index, other):
    out = torch.index_add(input, dim, index, other, alpha=2)
    return torch.index_select(out, dim, index).sum(dim)


def near_miss_other_dim(input, dim: int, index, other):
    out = torch.index_add(input, dim, index, other)
    return torch.index_select(out, dim, index).sum(1)


def near_miss_dtype(input, dim: int, index):
    return torch.index_select(input, dim, index).sum(dtype=torch.float32)


def near_miss_other_index(input, dim: int, index, other):
    out = torch.index_add(input, dim, index, other)
    return torch.index_select(out, dim, index + 0).sum(dim)


def near_miss_intermediate_used(input, dim: int, index):
    x = torch.index_select(input, dim, index)
    return x.sum() + x[0, 0]


def test_script_hook_rewrites_the_reference_bodies_only():
    jit.install_script_hook()
    try:
        a = torch.jit.script(ref_fused_index_select_reduce)
        b = torch.jit.script(ref_fused_index_add_reduce)
        assert jit.is_fused(a) and "aten::index_select" not in str(a.graph)
        assert jit.is_fused(b) and "aten::index_add" not in str(b.graph)
        for fn in (near_miss_alpha, near_miss_other_dim, near_miss_dtype, near_miss_other_index, near_miss_intermediate_used):
            assert not jit.is_fused(torch.jit.script(fn)), fn.__name__
        g = torch.Generator().manual_seed(0)
        x = torch.rand(9, 7, generator=g)
        idx = torch.randint(0, 9, (9,), generator=g)
        assert torch.equal(a(x, 0, idx), ref_fused_index_select_reduce(x, 0, idx))
        assert torch.equal(b(x, 0, idx, x.clone()), ref_fused_index_add_reduce(x, 0, idx, x.clone()))
        idx1 = torch.randint(0, 7, (7,), generator=g)
        assert torch.equal(b(x, 1, idx1, x.clone()), ref_fused_index_add_reduce(x, 1, idx1, x.clone()))
    finally:
        jit.uninstall_script_hook()
    assert not hasattr(torch.jit.script, "__wrapped__")        # after uninstall: stock scripting again


def ref_fused_index_select_reduce_again(input, dim: int, index):
    out = torch.index_select(input, dim, index).sum()
    return out


def test_fuse_applies_to_an_already_scripted_function():
    fn = torch.jit.script(ref_fused_index_select_reduce_again)   # torch caches the ScriptFunction per python function
    assert not jit.is_fused(fn)
    assert jit.is_fused(jit.fuse(fn))


def test_contiguous_patch_is_reversible_and_leaves_cpu_tensors_alone():
    from gnnops import aten

    orig = torch.Tensor.contiguous
    aten._patch_contiguous()
    try:
        assert torch.Tensor.contiguous is not orig
        x = torch.arange(12.).view(3, 4)
        y = torch.transpose(x, 0, 1).contiguous()          # a CPU tensor: the original method
        assert y.is_contiguous() and torch.equal(y, x.t().clone())
        assert x.contiguous() is x
        z = torch.arange(24.).view(2, 3, 4).permute(2, 0, 1).contiguous(memory_format=torch.contiguous_format)
        assert z.shape == (4, 2, 3) and z.is_contiguous()
    finally:
        aten._unpatch_contiguous()
    assert torch.Tensor.contiguous is orig
