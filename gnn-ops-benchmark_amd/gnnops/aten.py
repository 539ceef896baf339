"""Opt-in ATen overrides: route the native ops the reference scripts call by name to the HIP kernels.

Reference call sites: torch.index_select (op_bm_scripts/benchmark_native_index_select.py:14),
Tensor.index_add_ (benchmark_native_index_add_.py:15), torch.gather (benchmark_native_gather.py:16),
Tensor.scatter_add_ (benchmark_scatter_add.py:24), torch.index_add
(benchmark_fused_index_add_reduce.py:13), torch.sort (benchmark_native_sort.py:29), torch.addmm / torch.matmul
(benchmark_native_addmm.py:15, benchmark_native_matmul.py:15), Tensor.scatter_(reduce="multiply")
(benchmark_scatter_multiply.py:44), and — on the SparseCUDA key — torch.sparse.mm with a dense or a sparse right operand
(benchmark_sparse_spmm.py:13, benchmark_sparse_spspmm.py:13) and Tensor.coalesce() (benchmark_sparse_coalesce.py:41).

Two seams are not ATen kernels: `Tensor.contiguous()` of a 2-D transposed view (benchmark_sparse_transpose.py:13-16; see
_patch_contiguous) and the `@torch.jit.script` text of the two "fused" scripts (gnnops/jit.py).

ROCm builds of PyTorch register device kernels under the "CUDA" dispatch key, so overriding that key is
what makes ``device="cuda"`` script text reach our kernels. ``install()`` is reversible: ``uninstall()``
drops the Library object and the stock kernels come back. Inputs the kernels do not cover raise
NotImplementedError rather than silently running something else.
"""
import torch

from . import ops

_library = None
_orig_contiguous = None
routed_ops = set()   # names overridden by the last install()


def _patch_contiguous():
    """`torch.transpose(matA, 0, 1).contiguous()` (benchmark_sparse_transpose.py:13-16): `contiguous` is a composite op
    (clone -> empty_like + copy_), and a Python kernel on the CUDA key cannot hand the cases it does not want back to
    the stock `copy_` it replaced. So the seam is the METHOD: `Tensor.contiguous` gets a wrapper that sends a 2-D
    transposed view of a dense row-major device matrix through the LDS tile transpose (csrc/sparse.hip) and leaves
    every other call to the original method. Reversible (uninstall())."""
    global _orig_contiguous
    if _orig_contiguous is not None:
        return
    _orig_contiguous = torch.Tensor.contiguous

    def contiguous(self, *args, **kwargs):
        if (not args and not kwargs and self.dim() == 2 and self.is_cuda and not self.is_contiguous()
                and self.stride(0) == 1 and self.stride(1) == self.size(0) and self.size(0) > 1 and self.size(1) > 1
                and self.element_size() in (1, 2, 4, 8) and not self.is_complex()
                and not (torch.is_grad_enabled() and self.requires_grad)
                and not self.is_sparse and self.layout == torch.strided):
            from . import sparse

            return sparse.transpose_contiguous(self.t())     # self.t() is the dense row-major [C, R] matrix
        return _orig_contiguous(self, *args, **kwargs)

    torch.Tensor.contiguous = contiguous


def _unpatch_contiguous():
    global _orig_contiguous
    if _orig_contiguous is not None:
        torch.Tensor.contiguous = _orig_contiguous
        _orig_contiguous = None


def installed():
    return _library is not None


def install():
    global _library
    if _library is not None:
        return
    lib = torch.library.Library("aten", "IMPL")
    routed = set()

    def below_autograd(fn):
        """A kernel registered on the CUDA / SparseCUDA key runs BELOW PyTorch's Autograd key: the stock derivative
        formulas of the ATen op have already been recorded, so the raw kernels are what must run here (their own
        "operand requires grad" refusal does not apply)."""
        def kernel(*args, **kwargs):
            with torch.no_grad():
                return fn(*args, **kwargs)
        kernel.__name__ = fn.__name__
        return kernel

    _impl = lib.impl

    def impl(name, fn, key):
        _impl(name, below_autograd(fn), key)

    def index_select(self, dim, index):
        return ops.index_select(self, dim, index)

    def gather(self, dim, index, sparse_grad=False):
        return ops.gather(self, dim, index)

    def index_add_(self, dim, index, source, alpha=1):
        return ops.index_add_(self, dim, index, source, alpha)

    def index_add(self, dim, index, source, alpha=1):
        return ops.index_add_(self.clone(), dim, index, source, alpha)

    def scatter_add_(self, dim, index, src):
        return ops.scatter_add_(self, dim, index, src)

    def scatter_add(self, dim, index, src):
        return ops.scatter_add_(self.clone(), dim, index, src)

    def sort_stable(self, *, stable, dim=-1, descending=False):
        from . import sparse

        return sparse.sort(self, dim=dim, descending=descending, stable=bool(stable))

    def sort_default(self, dim=-1, descending=False):
        from . import sparse

        return sparse.sort(self, dim=dim, descending=descending)

    def addmm(self, mat1, mat2, *, beta=1, alpha=1):
        return ops.addmm(self, mat1, mat2, beta=beta, alpha=alpha)

    def mm(self, mat2):
        return ops.matmul(self, mat2)

    def scatter_reduce_(self, dim, index, src, *, reduce):
        # Tensor.scatter_(dim, index, src, reduce="add" | "multiply") (benchmark_scatter_multiply.py:44)
        if reduce == "add":
            return ops.scatter_add_(self, dim, index, src)
        if reduce == "multiply":
            return ops.scatter_reduce_mul_(self, dim, index, src)
        raise NotImplementedError(f"gnnops: scatter_(reduce={reduce!r}) is not supported")

    def sparse_mm(sparse, dense):
        from . import sparse as sp

        return sp.sparse_mm(sparse, dense)

    def sparse_coalesce(self):
        from . import sparse as sp

        return sp.coalesce_sparse_tensor(self)

    impl("scatter_.reduce", scatter_reduce_, "CUDA")
    routed.update({"scatter_.reduce"})
    # sparse COO operands dispatch on the SparseCUDA key (torch.sparse.mm, Tensor.coalesce()); optional: a build
    # that refuses these registrations keeps its stock kernels and gnnops.sparse_mm / coalesce_sparse_tensor stay
    # available by name
    def sparse_addmm(self, mat1, mat2, *, beta=1, alpha=1):
        # torch.sparse.mm(S, D) lowers to addmm(zeros, S, D, beta=0, alpha=1) on the sparse key
        from . import sparse as sp

        prod = sp.sparse_mm(mat1, mat2)
        if alpha != 1:
            prod = prod * alpha
        return prod if beta == 0 else prod + beta * self

    for name, fn in (("_sparse_mm", sparse_mm), ("addmm", sparse_addmm), ("_sparse_sparse_matmul", sparse_mm),
                     ("_coalesce", sparse_coalesce)):
        try:
            impl(name, fn, "SparseCUDA")
            routed.add(name + "@SparseCUDA")
        except Exception:  # pragma: no cover - depends on the torch build
            pass
    impl("sort.stable", sort_stable, "CUDA")
    impl("sort", sort_default, "CUDA")
    impl("addmm", addmm, "CUDA")
    impl("mm", mm, "CUDA")
    impl("index_select", index_select, "CUDA")
    impl("gather", gather, "CUDA")
    impl("index_add_", index_add_, "CUDA")
    impl("index_add", index_add, "CUDA")
    impl("scatter_add_", scatter_add_, "CUDA")
    impl("scatter_add", scatter_add, "CUDA")
    _library = lib
    _patch_contiguous()
    from . import jit

    jit.install_script_hook()
    global routed_ops
    routed_ops = routed | {"index_select", "gather", "index_add_", "index_add", "scatter_add_", "scatter_add", "sort",
                           "sort.stable", "addmm", "mm", "Tensor.contiguous (2-D transposed view)",
                           "torch.jit.script (index_select -> sum, index_add -> index_select -> sum(dim) rewritten)"}


def uninstall():
    global _library
    if _library is not None:
        _library._destroy()
        _library = None
    _unpatch_contiguous()
    from . import jit

    jit.uninstall_script_hook()
_orig_contiguous = None
routed_ops = set()   # names overridden by the last install()


def _patch_contiguous():
    """`torch.transpose(matA, 0, 1).contiguous()` (benchmark_sparse_transpose.py:13-16): `contiguous` is a composite op
    (clone -> empty_like + copy_), and a Python kernel on the CUDA key cannot hand the cases it does not want back to
    the stock `copy_` it replaced. So the seam is the METHOD: `Tensor.contiguous` gets a wrapper that sends a 2-D
    transposed view of a dense row-major device matrix through the LDS tile transpose (csrc/sparse.hip) and leaves
    every other call to the original method. Reversible (uninstall())."""
    global _orig_contiguous
    if _orig_contiguous is not None:
        return
    _orig_contiguous = torch.Tensor.contiguous

    def contiguous(self, *args, **kwargs):
        if (not args and not kwargs and self.dim() == 2 and self.is_cuda and not self.is_contiguous()
                and self.stride(0) == 1 and self.stride(1) == self.size(0) and self.size(0) > 1 and self.size(1) > 1
                and self.element_size() in (1, 2, 4, 8) and not self.is_complex()
                and not (torch.is_grad_enabled() and self.requires_grad)
                and not self.is_sparse and self.layout == torch.strided):
            from . import sparse

            return sparse.transpose_contiguous(self.t())     # self.t() is the dense row-major [C, R] matrix
        return _orig_contiguous(self, *args, **kwargs)

    torch.Tensor.contiguous = contiguous


def _unpatch_contiguous():
    global _orig_contiguous
    if _orig_contiguous is not None:
        torch.Tensor.contiguous = _orig_contiguous
        _orig_contiguous = None
