"""CPU suite: the TorchScript seam (gnnops/jit.py) and the fall-through mechanism of the ATen routes (gnnops/aten.py).

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


def _workload():
    """Calls of every routed operator with operands the HIP kernels do not take (here: CPU tensors; on the GPU box
    tests/test_fallthrough_gpu.py does the same with device tensors of unsupported dtype / layout / arguments)."""
    g = torch.Generator().manual_seed(5)
    a, b = torch.rand(6, 4, generator=g, dtype=torch.float64), torch.rand(4, 5, generator=g, dtype=torch.float64)
    out = [a @ b, torch.addmm(torch.ones(5, dtype=torch.float64), a, b, beta=0.5, alpha=2)]
    x = torch.zeros(5, 3)
    x.index_add_(0, torch.tensor([1, 1, 2]), torch.ones(3, 3), alpha=2)
    out.append(x)
    out.append(torch.index_add(torch.zeros(5, 3), 0, torch.tensor([4, 0, 0]), torch.ones(3, 3)))
    m = torch.randint(0, 50, (7, 9), generator=g)
    out += list(torch.sort(m, dim=1)) + list(torch.sort(m, dim=0, stable=True, descending=True))
    out.append(torch.zeros(3, 4, dtype=torch.int64).scatter_add_(1, torch.tensor([[0, 0, 1]]), torch.tensor([[1, 2, 3]])))
    out.append(torch.ones(3, 4).scatter_(1, torch.tensor([[0, 0, 1]]), torch.tensor([[1., 2, 3]]), reduce="multiply"))
    out.append(torch.index_select(a, 1, torch.tensor([3, 0])))
    out.append(torch.gather(a, 0, torch.randint(0, 6, (2, 4), generator=g)))
    out.append(torch.transpose(a, 0, 1).contiguous())
    s = torch.sparse_coo_tensor(torch.tensor([[0, 0, 1], [1, 1, 2]]), torch.tensor([1., 2, 3]), (3, 3))
    out += [s.coalesce().to_dense(), torch.sparse.mm(s, s).to_dense(), torch.sparse.mm(s, torch.ones(3, 2))]
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(), torch.nn.Linear(8, 2))
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    loss = net(torch.rand(16, 4, generator=g)).square().mean()
    loss.backward()
    opt.step()
    out += [loss.detach()] + [p.detach().clone() for p in net.parameters()]
    return out


def test_routes_hand_everything_they_do_not_accept_to_the_stock_kernel():
    """The mechanism of gnnops.install(), exercised on the CPU dispatch keys: every route captures the stock kernel with
    torch.library.get_kernel before it registers, and a call its predicate does not accept (CPU tensors never are) runs
    that kernel — results bit-identical to an un-routed process, a train step included; uninstall() restores it."""
    from gnnops import aten

    want = _workload()
    assert not aten.installed()
    aten.install(key="CPU", sparse_key="SparseCPU")
    try:
        assert aten.installed()
        for name in ("mm", "addmm", "index_add_", "index_add", "sort", "sort.stable", "scatter_add_", "scatter_.reduce",
                     "index_select", "gather", "clone"):
            assert name in aten.routed_ops, name
        aten.reset_stats()
        got = _workload()
        hip = {k: v[0] for k, v in aten.stats.items()}
        stock = {k: v[1] for k, v in aten.stats.items()}
        assert not any(hip.values()), hip                        # nothing here is a device tensor
        for name in ("mm", "addmm", "index_add_", "index_add", "sort.stable", "scatter_add_", "scatter_.reduce",
                     "index_select", "gather", "clone"):
            assert stock[name] >= 1, (name, stock)
    finally:
        aten.uninstall()
    assert not aten.installed() and not aten.routed_ops
    assert len(got) == len(want)
    for g_, w_ in zip(got, want):
        assert g_.dtype == w_.dtype and torch.equal(g_, w_)
    aten.reset_stats()
    _workload()
    assert not any(v[0] or v[1] for v in aten.stats.values())   # uninstalled: the routes are gone


def test_route_predicates_are_pure_and_reject_what_the_kernels_do_not_take():
    from gnnops import aten

    f32 = torch.rand(4, 4)
    idx = torch.tensor([0, 1])
    assert not aten.accepts_mm(f32, f32) and not aten.accepts_index_select(f32, 0, idx)       # CPU tensors
    assert not aten.accepts_clone(f32.t(), memory_format=torch.contiguous_format)
    meta = torch.empty(4, 4, device="meta")
    assert not aten.accepts_mm(meta, meta) and not aten.accepts_sort(meta, 1)
    assert aten._is_one(1) and aten._is_one(1.0) and not aten._is_one(2) and not aten._is_one(True)
