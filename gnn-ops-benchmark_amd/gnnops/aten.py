"""Opt-in ATen routes: the native ops the reference scripts call by name reach the HIP kernels — and every operand
the kernels do not take reaches the STOCK kernel it always reached.

Reference call sites: torch.index_select (op_bm_scripts/benchmark_native_index_select.py:14),
Tensor.index_add_ (benchmark_native_index_add_.py:15), torch.gather (benchmark_native_gather.py:16),
Tensor.scatter_add_ (benchmark_scatter_add.py:24), torch.index_add
(benchmark_fused_index_add_reduce.py:13), torch.sort (benchmark_native_sort.py:29), torch.addmm / torch.matmul
(benchmark_native_addmm.py:15, benchmark_native_matmul.py:15), Tensor.scatter_(reduce="multiply")
(benchmark_scatter_multiply.py:44), `torch.transpose(matA, 0, 1).contiguous()` (benchmark_sparse_transpose.py:13-16:
`contiguous` of a strided view is `clone(memory_format=contiguous_format)`), and — on the SparseCUDA key —
torch.sparse.mm with a dense or a sparse right operand (benchmark_sparse_spmm.py:13, benchmark_sparse_spspmm.py:13) and
Tensor.coalesce() (benchmark_sparse_coalesce.py:41). The same process also runs the reference's OTHER callers of these
ATen ops (graph_benchmark/profile/OpProfiler.py:259-322 train + eval loops over graph_benchmark/models/ptg_models.py:
nn.Linear in fp32 / fp64, integer `degree` scatters, sorts of int64 matrices, ...), so the seam has ONE mechanism:

    stock  = torch.library.get_kernel("aten::<op>", key)        # the kernel PyTorch itself would run, captured first
    kernel = lambda keyset, *a, **kw: hip(*a, **kw) if accepts(*a, **kw) else stock.call_boxed(keyset, *a, **kw)
    Library("aten", "IMPL").impl("<op>", kernel, key, with_keyset=True)

`accepts` is a pure predicate over (device, dtype, layout, rank, contiguity, scalar arguments) — one per route, below —
that says exactly what the gfx950 kernel behind the route takes; anything else is handed, untouched, to the captured
stock kernel (bit-for-bit PyTorch's own result, errors included). A route's HIP side may still meet a case only the
kernel can judge (`NotImplementedError` from an argument guard or an EUNSUPPORTED status, both raised before anything is
written): that also falls through. ROCm builds of PyTorch register device kernels under the "CUDA" dispatch key, so
routing that key is what makes ``device="cuda"`` script text reach our kernels. ``install()`` is reversible:
``uninstall()`` drops the Library object and the stock kernels are the registered ones again. Routes sit BELOW autograd
(PyTorch has already recorded the op's own derivative formula), so a train step through routed ops is differentiated
by PyTorch as always.

The one seam that is not an ATen kernel is the `@torch.jit.script` text of the two "fused" scripts: a graph rewrite
(gnnops/jit.py) — no single operator sees the index_select -> sum chain.
"""
import torch

from . import ops

_library = None
routed_ops = set()   # names routed by the last install()
stats = {}           # route name -> [calls that ran the HIP kernel, calls handed to the stock kernel]

_FLOATS = (torch.float32, torch.float16, torch.bfloat16)
_SORT_DTYPES = (torch.float32, torch.float16, torch.bfloat16, torch.int32, torch.int64, torch.float64)
_SORT_1D_ONLY = (torch.int64, torch.float64)


# ---- what the kernels take (pure predicates; no device work, no allocation) ---------------------------------------
# These run on EVERY call of a routed op, and at the reference's smallest shapes a call is host-bound (tools/host_overhead.py:
# the first version of accepts_scatter cost 10 us of a 24-us call): plain attribute tests in cheapest-first order, no
# generators. What the dispatcher already guarantees is not re-tested: a kernel on the CUDA key only sees strided, non-quantized
# tensors with the conjugate / negative bits resolved (sparse, quantized and lazy-bit tensors carry other keys above it).
# `get_device()` is -1 for a CPU tensor, so one comparison covers "on the GPU" and "on the SAME GPU".
def _dev(t):
    return t.get_device() if isinstance(t, torch.Tensor) else -2


def _copyable(t):
    return t.element_size() in (1, 2, 4, 8) and not t.is_complex()


def _dim_ok(dim, ndim):
    return isinstance(dim, int) and ndim >= 1 and -ndim <= dim < ndim


def _spans(index, self, d):
    """index.size(k) == self.size(k) for every k != d"""
    a, b = index.shape, self.shape
    for k in range(len(b)):
        if k != d and a[k] != b[k]:
            return False
    return True


def accepts_index_select(self, dim, index):
    dv = _dev(self)
    return (dv >= 0 and _dev(index) == dv and index.dim() == 1 and index.dtype in (torch.int64, torch.int32)
            and _dim_ok(dim, self.dim()) and _copyable(self) and self.is_contiguous() and self.numel() > 0 and index.numel() > 0)


def accepts_gather(self, dim, index, sparse_grad=False):
    dv = _dev(self)
    nd = self.dim()
    if not (dv >= 0 and _dev(index) == dv and index.dtype == torch.int64 and index.dim() == nd and _dim_ok(dim, nd)
            and _copyable(self) and self.is_contiguous() and index.is_contiguous() and self.numel() > 0 and index.numel() > 0):
        return False
    return _spans(index, self, dim % nd)


def accepts_index_add(self, dim, index, source, alpha=1):
    dv = _dev(self)
    nd = self.dim()
    if not (dv >= 0 and self.dtype in _FLOATS and _is_one(alpha) and _dev(index) == dv and _dev(source) == dv
            and source.dtype == self.dtype and index.dtype in (torch.int64, torch.int32) and index.dim() == 1
            and source.dim() == nd and _dim_ok(dim, nd) and self.is_contiguous() and source.is_contiguous()
            and self.numel() > 0 and source.numel() > 0):
        return False
    d = dim % nd
    return index.numel() == source.size(d) and _spans(source, self, d)


def accepts_scatter(self, dim, index, src):
    dv = _dev(self)
    nd = self.dim()
    if not (dv >= 0 and self.dtype in _FLOATS and _dev(index) == dv and _dev(src) == dv and src.dtype == self.dtype
            and index.dtype == torch.int64 and index.dim() == nd and src.dim() == nd and _dim_ok(dim, nd)
            and self.is_contiguous() and self.numel() > 0 and index.numel() > 0):
        return False
    if not _spans(index, self, dim % nd):
        return False
    a, b = index.shape, src.shape
    for k in range(nd):
        if a[k] > b[k]:
            return False
    return True


def accepts_scatter_reduce(self, dim, index, src, *, reduce):
    return reduce in ("add", "multiply") and accepts_scatter(self, dim, index, src)


def accepts_sort(self, dim=-1, descending=False, *, stable=None):
    if not (_dev(self) >= 0 and self.dtype in _SORT_DTYPES and _dim_ok(dim, self.dim()) and self.is_contiguous()
            and self.numel() > 0):
        return False
    return self.dtype not in _SORT_1D_ONLY or self.numel() == self.size(dim)


def _is_one(x):
    return x == 1 and isinstance(x, (int, float)) and not isinstance(x, bool)


def accepts_mm(self, mat2):
    dv = _dev(self)
    return (dv >= 0 and self.dtype in _FLOATS and _dev(mat2) == dv and mat2.dtype == self.dtype and self.dim() == 2
            and mat2.dim() == 2 and self.size(1) == mat2.size(0) and self.is_contiguous() and mat2.is_contiguous()
            and self.numel() > 0 and mat2.numel() > 0)


def accepts_addmm(self, mat1, mat2, *, beta=1, alpha=1):
    if not (_is_one(beta) and _is_one(alpha) and accepts_mm(mat1, mat2) and _dev(self) == _dev(mat1)
            and self.dtype == mat1.dtype and self.dim() in (1, 2) and self.is_contiguous()):
        return False
    M, N = mat1.size(0), mat2.size(1)
    shape = (1,) * (2 - self.dim()) + tuple(self.shape)
    return shape[0] in (1, M) and shape[1] in (1, N)


def accepts_clone(self, *, memory_format=None):
    """`m.transpose(0, 1).contiguous()` of a dense row-major 2-D matrix: the LDS tile transpose (csrc/sparse.hip)."""
    return (memory_format is torch.contiguous_format and self.dim() == 2 and self.stride(0) == 1 and self.size(0) > 1
            and self.size(1) > 1 and self.stride(1) == self.size(0) and _dev(self) >= 0 and _copyable(self))


def _coo2(t, floats=True):
    return (isinstance(t, torch.Tensor) and t.is_cuda and t.layout == torch.sparse_coo and t.dim() == 2
            and t.sparse_dim() == 2 and t.dense_dim() == 0 and (t.dtype in _FLOATS or not floats) and t._nnz() > 0)


def accepts_sparse_dense(sparse, dense):
    return (_coo2(sparse) and isinstance(dense, torch.Tensor) and dense.layout == torch.strided and dense.dim() == 2
            and dense.dtype == sparse.dtype and dense.is_contiguous() and sparse.size(1) == dense.size(0) and dense.numel() > 0
            and dense.get_device() == sparse.get_device())


def accepts_sparse_addmm(self, mat1, mat2, *, beta=1, alpha=1):
    # torch.sparse.mm(S, D) lowers to addmm(zeros, S, D, beta=0, alpha=1) on the sparse key; nothing else is ours
    return (isinstance(beta, (int, float)) and beta == 0 and _is_one(alpha) and isinstance(self, torch.Tensor)
            and self.layout == torch.strided and accepts_sparse_dense(mat1, mat2))


def accepts_sparse_sparse(self, other):
    return (_coo2(self) and _coo2(other) and self.dtype == other.dtype and self.size(1) == other.size(0)
            and self.get_device() == other.get_device())


def accepts_coalesce(self):
    return (isinstance(self, torch.Tensor) and self.is_cuda and self.layout == torch.sparse_coo and self.sparse_dim() == 2
            and self.dim() == 2 and self.dtype in _FLOATS and self._nnz() > 0 and not self.is_coalesced())


# ---- the HIP side of each route --------------------------------------------------------------------------------
def _hip_index_select(self, dim, index):
    return ops.index_select(self, dim, index)


def _hip_gather(self, dim, index, sparse_grad=False):
    return ops.gather(self, dim, index)


def _hip_index_add_(self, dim, index, source, alpha=1):
    return ops.index_add_(self, dim, index, source)


def _hip_index_add(self, dim, index, source, alpha=1):
    return ops.index_add_(_fresh_copy(self), dim, index, source)


def _hip_scatter_add_(self, dim, index, src):
    return ops.scatter_add_(self, dim, index, src)


def _hip_scatter_add(self, dim, index, src):
    return ops.scatter_add_(_fresh_copy(self), dim, index, src)


def _hip_scatter_reduce_(self, dim, index, src, *, reduce):
    # Tensor.scatter_(dim, index, src, reduce="add" | "multiply") (benchmark_scatter_multiply.py:44)
    if reduce == "add":
        return ops.scatter_add_(self, dim, index, src)
    return ops.scatter_reduce_mul_(self, dim, index, src)


def _hip_sort_stable(self, *, stable, dim=-1, descending=False):
    from . import sparse

    return sparse.sort(self, dim=dim, descending=descending, stable=True)     # the radix engine is always stable


def _hip_sort(self, dim=-1, descending=False):
    from . import sparse

    return sparse.sort(self, dim=dim, descending=descending)


def _hip_addmm(self, mat1, mat2, *, beta=1, alpha=1):
    return ops.addmm(self, mat1, mat2)


def _hip_mm(self, mat2):
    return ops.matmul(self, mat2)


def _hip_clone(self, *, memory_format=None):
    from . import sparse

    return sparse.transpose_contiguous(self.t())     # self.t() is the dense row-major [C, R] matrix


def _hip_sparse_mm(sparse, dense):
    from . import sparse as sp

    return sp.sparse_mm(sparse, dense)


def _hip_sparse_addmm(self, mat1, mat2, *, beta=1, alpha=1):
    from . import sparse as sp

    return sp.sparse_mm(mat1, mat2)


def _hip_coalesce(self):
    from . import sparse as sp

    return sp.coalesce_sparse_tensor(self)


def _fresh_copy(t):
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    out.copy_(t)
    return out


def _accepts_sort_stable(self, *, stable, dim=-1, descending=False):
    return accepts_sort(self, dim, descending, stable=stable)


# (operator, dispatch key, accepts, hip). The sparse routes are optional: a build that refuses them keeps its kernels.
ROUTES = (
    ("index_select", "CUDA", accepts_index_select, _hip_index_select),
    ("gather", "CUDA", accepts_gather, _hip_gather),
    ("index_add_", "CUDA", accepts_index_add, _hip_index_add_),
    ("index_add", "CUDA", accepts_index_add, _hip_index_add),
    ("scatter_add_", "CUDA", accepts_scatter, _hip_scatter_add_),
    ("scatter_add", "CUDA", accepts_scatter, _hip_scatter_add),
    ("scatter_.reduce", "CUDA", accepts_scatter_reduce, _hip_scatter_reduce_),
    ("sort.stable", "CUDA", _accepts_sort_stable, _hip_sort_stable),
    ("sort", "CUDA", accepts_sort, _hip_sort),
    ("addmm", "CUDA", accepts_addmm, _hip_addmm),
    ("mm", "CUDA", accepts_mm, _hip_mm),
    ("clone", "CUDA", accepts_clone, _hip_clone),
    ("_sparse_mm", "SparseCUDA", accepts_sparse_dense, _hip_sparse_mm),
    ("addmm", "SparseCUDA", accepts_sparse_addmm, _hip_sparse_addmm),
    ("_sparse_sparse_matmul", "SparseCUDA", accepts_sparse_sparse, _hip_sparse_mm),
    ("_coalesce", "SparseCUDA", accepts_coalesce, _hip_coalesce),
)


def _route_name(op, key):
    return op if key == "CUDA" else f"{op}@{key}"


def _make_kernel(name, accepts, hip, stock):
    count = stats.setdefault(name, [0, 0])

    def kernel(keyset, *args, **kwargs):
        if accepts(*args, **kwargs):
            try:
                # below PyTorch's Autograd key: the op's own derivative is already recorded, the raw kernel is what runs
                with torch.no_grad():
                    out = hip(*args, **kwargs)
                count[0] += 1
                return out
            except NotImplementedError:
                pass            # an argument guard / EUNSUPPORTED status: raised before anything was written
        count[1] += 1
        return stock.call_boxed(keyset, *args, **kwargs)

    kernel.__name__ = "gnnops_route_" + name.replace(".", "_").replace("@", "_")
    return kernel


def installed():
    return _library is not None


def install(key="CUDA", sparse_key="SparseCUDA"):
    """Route the ops of ROUTES. `key` / `sparse_key` exist for the CPU suite, which exercises the fall-through
    mechanism itself on the CPU keys (where no operand is ever accepted: the predicates require device tensors)."""
    global _library, routed_ops
    if _library is not None:
        return
    import warnings

    lib = torch.library.Library("aten", "IMPL")
    routed = set()
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", message="(?s).*Overriding a previously registered kernel.*")   # that is the point
        for op, route_key, accepts, hip in ROUTES:
            real_key = key if route_key == "CUDA" else sparse_key
            name = _route_name(op, route_key)
            try:
                stock = torch.library.get_kernel("aten::" + op, real_key)     # captured BEFORE the route is registered
                lib.impl(op, _make_kernel(name, accepts, hip, stock), real_key, with_keyset=True)
                routed.add(name)
            except Exception:  # pragma: no cover - depends on the torch build
                if route_key == "CUDA":
                    lib._destroy()
                    raise
    _library = lib
    from . import jit

    jit.install_script_hook()
    routed_ops = routed | {"torch.jit.script (index_select -> sum, index_add -> index_select -> sum(dim) rewritten)"}


def uninstall():
    global _library, routed_ops
    if _library is not None:
        _library._destroy()
        _library = None
    routed_ops = set()
    from . import jit

    jit.uninstall_script_hook()


def reset_stats():
    for v in stats.values():
        v[0] = v[1] = 0
