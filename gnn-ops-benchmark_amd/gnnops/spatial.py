"""The remaining ops of the reference's list (ops.txt:17-19, 29-41) behind the packages' own
signatures: torch_spline_conv.{spline_basis, spline_weighting, spline_conv} and torch_cluster.{grid_cluster, fps, knn,
knn_graph, radius, radius_graph, nearest, random_walk}. The import seams are the sibling packages ``torch_spline_conv`` and
``torch_cluster`` in this directory. Kernels: csrc/spline.hip, csrc/cluster.hip. Parity unpinned (neither package nor any
output of it is in the reference tree): oracle/spatial_oracle.py restates the published definitions.

Forward only. Device tensors only — CPU tensors are refused, not emulated."""
import weakref

import torch

from . import _lib, ops
from ._lib import check
from .ops import _dtype_code, _on, _require_gpu, _stream, get_plan
from .sparse import _csr_arrays


# ---------------------------------------------------------------------------------------------------------------------
# torch_spline_conv
# ---------------------------------------------------------------------------------------------------------------------
_host_cache = {}   # id(tensor) -> (weakref to it, version, host copy): kernel_size / is_open_spline are tiny and constant per layer


def _host(t, dtype):
    """kernel_size / is_open_spline as host arrays (the C entry points read them while launching). A device tensor is read
    back once per tensor OBJECT and version — one synchronisation when a layer is first used, none afterwards. The weak
    reference is what makes the entry safe: an id (and a data pointer) can be handed to a new tensor once the old one died."""
    if t.device.type == "cpu":
        return t.to(dtype).contiguous()
    hit = _host_cache.get(id(t))
    if hit is not None and hit[0]() is t and hit[1] == t._version:
        return hit[2]
    host = t.detach().to("cpu", dtype).contiguous()
    if not t.is_inference():
        key = id(t)
        _host_cache[key] = (weakref.ref(t, lambda _r, key=key: _host_cache.pop(key, None)), t._version, host)
    return host


def spline_basis(pseudo, kernel_size, is_open_spline, degree):
    """torch_spline_conv.spline_basis (torch.ops.torch_spline_conv.spline_basis, ops.txt:19): (basis [E, S], weight_index [E, S])."""
    _require_gpu(pseudo)
    ops._refuse_grad("spline_basis", pseudo)
    if pseudo.dim() == 1:
        pseudo = pseudo.unsqueeze(-1)
    pseudo = pseudo.contiguous()
    E, D = pseudo.shape
    dt = _dtype_code(pseudo, "spline_basis")
    ks, op = _host(kernel_size, torch.int64), _host(is_open_spline, torch.uint8)
    if ks.numel() != D or op.numel() != D:
        raise RuntimeError("spline_basis: kernel_size and is_open_spline need one entry per pseudo-coordinate")
    S = (degree + 1) ** D
    basis = torch.empty((E, S), dtype=pseudo.dtype, device=pseudo.device)
    wi = torch.empty((E, S), dtype=torch.int64, device=pseudo.device)
    with _on(pseudo.device):
        check(_lib.load().gnnops_spline_basis(pseudo.data_ptr(), ks.data_ptr(), op.data_ptr(), E, D, int(degree), basis.data_ptr(),
                                              wi.data_ptr(), dt, _stream()), "spline_basis")
    return basis, wi


def spline_weighting(x, weight, basis, weight_index):
    """torch_spline_conv.spline_weighting (ops.txt:18): x [E, Min], weight [K, Min, Mout] -> [E, Mout]."""
    _require_gpu(x, weight, basis, weight_index)
    ops._refuse_grad("spline_weighting", x, weight, basis)
    if x.dim() != 2 or weight.dim() != 3 or weight.size(1) != x.size(1) or basis.shape != weight_index.shape or basis.size(0) != x.size(0):
        raise RuntimeError("spline_weighting: x [E, Min], weight [K, Min, Mout], basis / weight_index [E, S]")
    if not (x.dtype == weight.dtype == basis.dtype) or weight_index.dtype != torch.int64:
        raise RuntimeError("spline_weighting: x, weight and basis share a dtype; weight_index is int64")
    dt = _dtype_code(x, "spline_weighting")
    x, weight, basis, weight_index = x.contiguous(), weight.contiguous(), basis.contiguous(), weight_index.contiguous()
    E, Min = x.shape
    Mout = weight.size(2)
    out = torch.empty((E, Mout), dtype=x.dtype, device=x.device)
    with _on(x.device):
        check(_lib.load().gnnops_spline_weighting(x.data_ptr(), weight.data_ptr(), basis.data_ptr(), weight_index.data_ptr(),
                                                  out.data_ptr(), E, Min, Mout, basis.size(1), dt, _stream()), "spline_weighting")
    return out


def spline_conv(x, edge_index, pseudo, weight, kernel_size, is_open_spline, degree=1, norm=True, root_weight=None, bias=None):
    """torch_spline_conv.spline_conv (ops.txt:31): messages x[edge_index[1]] blended through the B-spline kernel, summed at
    edge_index[0] (divided by that row's degree when ``norm``), + x @ root_weight + bias. One pass over the plan of
    edge_index[0]; neither the basis tensors nor the [E, Mout] messages are materialised."""
    _require_gpu(x, edge_index, pseudo, weight, root_weight, bias)
    ops._refuse_grad("spline_conv", x, pseudo, weight, root_weight, bias)
    if x.dim() == 1:
        x = x.unsqueeze(-1)
    if pseudo.dim() == 1:
        pseudo = pseudo.unsqueeze(-1)
    if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.int64:
        raise ValueError("spline_conv: edge_index must be int64 [2, E]")
    x, pseudo, weight = x.contiguous(), pseudo.contiguous(), weight.contiguous()
    edge_index = edge_index.contiguous()
    N, Min = x.shape
    E, D = pseudo.shape
    if edge_index.size(1) != E or weight.dim() != 3 or weight.size(1) != Min:
        raise RuntimeError("spline_conv: pseudo has one row per edge; weight is [K, Min, Mout]")
    Mout = weight.size(2)
    dt = _dtype_code(x, "spline_conv")
    for t in (pseudo, weight, root_weight, bias):
        if t is not None and t.dtype != x.dtype:
            raise RuntimeError("spline_conv: operands must have the same dtype")
    ks, op = _host(kernel_size, torch.int64), _host(is_open_spline, torch.uint8)
    if ks.numel() != D or op.numel() != D:
        raise RuntimeError("spline_conv: kernel_size and is_open_spline need one entry per pseudo-coordinate")
    row, col = edge_index[0], edge_index[1]
    plan = get_plan(row, N, owner=edge_index, tag=0, companion=col)
    if plan.col is not None or E == 0:
        src = plan.col if E else col
    else:
        src, _ = _csr_arrays(plan, col, None, owner=edge_index, tag=1)
    out = torch.empty((N, Mout), dtype=x.dtype, device=x.device)
    ptr = lambda t: t.contiguous().data_ptr() if t is not None else None   # noqa: E731
    root_c = root_weight.contiguous() if root_weight is not None else None
    bias_c = bias.contiguous() if bias is not None else None
    with _on(x.device):
        check(_lib.load().gnnops_spline_conv(x.data_ptr(), plan.rowptr.data_ptr(), plan.perm.data_ptr(), src.data_ptr(), pseudo.data_ptr(),
                                             weight.data_ptr(), ks.data_ptr(), op.data_ptr(), D, int(degree), ptr(root_c), ptr(bias_c),
                                             out.data_ptr(), N, E, Min, Mout, 1 if norm else 0, dt, _stream()), "spline_conv")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# torch_cluster
# ---------------------------------------------------------------------------------------------------------------------
def _ptr_of(batch, n, device, batch_size=None):
    """CSR pointer over points sorted by batch (what the package builds from `batch`); None = one batch of everything."""
    if batch is None:
        return torch.tensor([0, n], dtype=torch.int64, device=device)
    if batch.numel() != n:
        raise RuntimeError("batch must have one entry per point")
    B = int(batch.max()) + 1 if batch_size is None and batch.numel() else (batch_size or 1)   # the package reads batch.max() back too
    ptr = torch.zeros(B + 1, dtype=torch.int64, device=device)
    ptr[1:] = torch.bincount(batch, minlength=B).cumsum(0)
    return ptr


def _points(t, what):
    _require_gpu(t)
    t = t.view(-1, 1) if t.dim() == 1 else t
    if t.dim() != 2:
        raise RuntimeError(f"{what}: points must be [N, D]")
    return t.contiguous()


def grid_cluster(pos, size, start=None, end=None):
    """torch_cluster.grid_cluster(pos, size, start=None, end=None) -> voxel id per point (ops.txt:36)."""
    pos = _points(pos, "grid_cluster")
    dt = _dtype_code(pos, "grid_cluster")
    N, D = pos.shape
    dev = pos.device
    as_dev = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev).reshape(-1).contiguous()   # noqa: E731
    size = as_dev(size)
    start = as_dev(start) if start is not None else pos.min(dim=0).values.to(torch.float64)
    end = as_dev(end) if end is not None else pos.max(dim=0).values.to(torch.float64)
    if not (size.numel() == start.numel() == end.numel() == D):
        raise RuntimeError("grid_cluster: size, start and end need one entry per coordinate")
    out = torch.empty(N, dtype=torch.int64, device=dev)
    with _on(dev):
        check(_lib.load().gnnops_grid_cluster(pos.data_ptr(), N, D, size.data_ptr(), start.contiguous().data_ptr(), end.contiguous().data_ptr(),
                                              out.data_ptr(), dt, _stream()), "grid_cluster")
    return out


def fps(x, batch=None, ratio=0.5, random_start=True):
    """torch_cluster.fps(x, batch=None, ratio=0.5, random_start=True) -> indices of the sampled points, batch after batch:
    ceil(ratio * n_b) per batch, each the point farthest from those already chosen (ties: the smaller index)."""
    x = _points(x, "fps")
    if not 0.0 < float(ratio) <= 1.0:
        raise ValueError("fps: ratio must be in (0, 1]")
    dt = _dtype_code(x, "fps")
    N, D = x.shape
    dev = x.device
    ptr = _ptr_of(batch, N, dev)
    deg = ptr[1:] - ptr[:-1]
    k = torch.ceil(deg.to(torch.float64) * float(ratio)).to(torch.int64)
    out_ptr = torch.zeros_like(ptr)
    out_ptr[1:] = k.cumsum(0)
    total = int(out_ptr[-1])                                   # the package sizes its output the same way (one read-back)
    if random_start:
        start = ptr[:-1] + (torch.rand(deg.numel(), device=dev) * deg.to(torch.float32)).to(torch.int64).clamp_(max=(deg - 1).clamp_(min=0))
    else:
        start = ptr[:-1].clone()
    out = torch.empty(total, dtype=torch.int64, device=dev)
    dist = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().gnnops_fps(x.data_ptr(), ptr.data_ptr(), out_ptr.data_ptr(), start.data_ptr(), ptr.numel() - 1, D, dist.data_ptr(),
                                     out.data_ptr(), dt, _stream()), "fps")
    return out


def _pair_ptrs(x, y, batch_x, batch_y):
    if (batch_x is None) != (batch_y is None):
        raise RuntimeError("give both batch_x and batch_y or neither")
    if batch_x is None:
        return _ptr_of(None, x.size(0), x.device), _ptr_of(None, y.size(0), x.device)
    B = int(max(batch_x.max(), batch_y.max())) + 1 if batch_x.numel() and batch_y.numel() else 1
    return _ptr_of(batch_x, x.size(0), x.device, B), _ptr_of(batch_y, y.size(0), x.device, B)


def _pairs(col, k):
    """[Ny, k] neighbour table with -1 for 'none' -> the package's [2, M] (row = query index, col = neighbour index)."""
    ny = col.size(0)
    row = torch.arange(ny, device=col.device).view(-1, 1).expand(ny, k).reshape(-1)
    col = col.reshape(-1)
    mask = col >= 0
    return torch.stack([row[mask], col[mask]], dim=0)


_KNN_GRID_MIN_POINTS = 8192     # one cloud of at least this many points is searched through a uniform grid (csrc/cluster.hip)
_KNN_GRID_PER_CELL = 8          # points per cell aimed at
_KNN_GRID_MAX_BATCHES = 64      # with a batch vector: at most this many clouds are taken one by one so that the large ones get a grid


def _knn_grid(x, y, k, radius=None):
    """One cloud, Euclidean, D <= 3, fp32, k <= 64: bin x into G^D cells of its bounding box, order the points by cell with
    the plan builder (cell = destination: rowptr = first point of each cell, perm = points in cell order) and let every
    query walk the cells around its own. Everything stays on the stream; the pairs are the exhaustive kernel's.
    radius = r: torch_cluster.radius instead — the k smallest indices inside the ball."""
    L = _lib.load()
    nx, D = x.shape
    G = int(round((nx / _KNN_GRID_PER_CELL) ** (1.0 / D)))
    G = max(1, min(G, 1023 if D == 3 else 2048 if D == 2 else 4096))
    cells = G ** D
    dev = x.device
    box = torch.empty(6, dtype=torch.int32, device=dev)
    cell = torch.empty(nx, dtype=torch.int64, device=dev)
    rowptr = torch.empty(cells + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(nx, dtype=torch.int32, device=dev)
    ws_bytes = L.gnnops_plan_workspace_bytes(nx, cells)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    col = torch.empty((y.size(0), k), dtype=torch.int64, device=dev)
    with _on(dev):
        check(L.gnnops_knn_grid_cells(x.data_ptr(), nx, D, G, box.data_ptr(), cell.data_ptr(), _stream()), "knn_grid_cells")
        check(L.gnnops_plan_build(cell.data_ptr(), nx, cells, rowptr.data_ptr(), perm.data_ptr(), ws.data_ptr(), ws_bytes, _stream()), "plan_build")
        if radius is None:
            check(L.gnnops_knn_grid_query(x.data_ptr(), y.data_ptr(), y.size(0), D, k, G, box.data_ptr(), rowptr.data_ptr(), perm.data_ptr(),
                                          col.data_ptr(), _stream()), "knn_grid_query")
        else:
            check(L.gnnops_radius_grid_query(x.data_ptr(), y.data_ptr(), y.size(0), D, float(radius), k, G, box.data_ptr(), rowptr.data_ptr(),
                                             perm.data_ptr(), col.data_ptr(), _stream()), "radius_grid_query")
    return col


def _by_cloud(x, y, k, ptr_x, ptr_y, radius, exhaustive):
    """A batch vector over a FEW clouds of which at least one is large (>= _KNN_GRID_MIN_POINTS points): the clouds are taken one by
    one — the large ones walk a grid of their own, the others go through `exhaustive(x_slice, y_slice)` — and the neighbour
    indices are shifted back by the cloud's first point. None when that does not apply (the caller makes its one exhaustive call)."""
    B = ptr_x.numel() - 1
    if B < 1 or B > _KNN_GRID_MAX_BATCHES or x.size(0) < _KNN_GRID_MIN_POINTS:
        return None
    if B > 1 and x.size(0) < B * (_KNN_GRID_MIN_POINTS // 4):
        return None      # many small clouds (32 x 1024 points): not worth reading the pointers back to find out that none is large
    px, py = ptr_x.tolist(), ptr_y.tolist()
    if not any(px[b + 1] - px[b] >= _KNN_GRID_MIN_POINTS and py[b + 1] > py[b] for b in range(B)):
        return None
    col = torch.full((y.size(0), k), -1, dtype=torch.int64, device=x.device)
    for b in range(B):
        if py[b + 1] == py[b] or px[b + 1] == px[b]:
            continue
        xs, ys = x[px[b]:px[b + 1]], y[py[b]:py[b + 1]]
        part = _knn_grid(xs, ys, k, radius=radius) if xs.size(0) >= _KNN_GRID_MIN_POINTS else exhaustive(xs, ys)
        col[py[b]:py[b + 1]] = torch.where(part >= 0, part + px[b], part)
    return col


def knn(x, y, k, batch_x=None, batch_y=None, cosine=False, num_workers=1):
    """torch_cluster.knn(x, y, k, batch_x, batch_y, cosine): for every y its k nearest x of the same batch, nearest first:
    int64 [2, M] = (index into y, index into x)."""
    x, y = _points(x, "knn"), _points(y, "knn")
    if x.size(1) != y.size(1) or x.dtype != y.dtype:
        raise RuntimeError("knn: x and y need the same width and dtype")
    dt = _dtype_code(x, "knn")
    if (batch_x is None and batch_y is None and not cosine and x.dtype == torch.float32 and x.size(1) <= 3 and 1 <= int(k) <= 64
            and x.size(0) >= _KNN_GRID_MIN_POINTS and x.size(0) < 2 ** 31):
        return _pairs(_knn_grid(x, y, int(k)), k)
    ptr_x, ptr_y = _pair_ptrs(x, y, batch_x, batch_y)

    def exhaustive(xs, ys, px=None, py=None):
        px = _ptr_of(None, xs.size(0), xs.device) if px is None else px
        py = _ptr_of(None, ys.size(0), xs.device) if py is None else py
        out = torch.empty((ys.size(0), k), dtype=torch.int64, device=xs.device)
        with _on(xs.device):
            check(_lib.load().gnnops_knn(xs.data_ptr(), ys.data_ptr(), px.data_ptr(), py.data_ptr(), px.numel() - 1, ys.size(0), xs.size(1),
                                         int(k), 1 if cosine else 0, out.data_ptr(), dt, _stream()), "knn")
        return out

    col = None
    if not cosine and x.dtype == torch.float32 and x.size(1) <= 3 and 1 <= int(k) <= 64 and x.size(0) < 2 ** 31:
        col = _by_cloud(x, y, int(k), ptr_x, ptr_y, None, exhaustive)
    if col is None:
        col = exhaustive(x, y, ptr_x, ptr_y)
    return _pairs(col, k)


def knn_graph(x, k, batch=None, loop=False, flow="source_to_target", cosine=False, num_workers=1):
    """torch_cluster.knn_graph (ops.txt:38): edges from the k nearest neighbours to each point."""
    if flow not in ("source_to_target", "target_to_source"):
        raise ValueError(flow)
    ei = knn(x, x, k if loop else k + 1, batch, batch, cosine)
    row, col = (ei[1], ei[0]) if flow == "source_to_target" else (ei[0], ei[1])
    if not loop:
        mask = row != col
        row, col = row[mask], col[mask]
    return torch.stack([row, col], dim=0)


def radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32, num_workers=1):
    """torch_cluster.radius: for every y the x of the same batch within distance r (at most max_num_neighbors, by index)."""
    x, y = _points(x, "radius"), _points(y, "radius")
    if x.size(1) != y.size(1) or x.dtype != y.dtype:
        raise RuntimeError("radius: x and y need the same width and dtype")
    dt = _dtype_code(x, "radius")
    if (batch_x is None and batch_y is None and x.dtype == torch.float32 and x.size(1) <= 3 and 1 <= int(max_num_neighbors) <= 64
            and x.size(0) >= _KNN_GRID_MIN_POINTS and x.size(0) < 2 ** 31 and float(r) >= 0):
        return _pairs(_knn_grid(x, y, int(max_num_neighbors), radius=r), max_num_neighbors)
    ptr_x, ptr_y = _pair_ptrs(x, y, batch_x, batch_y)

    def exhaustive(xs, ys, px=None, py=None):
        px = _ptr_of(None, xs.size(0), xs.device) if px is None else px
        py = _ptr_of(None, ys.size(0), xs.device) if py is None else py
        out = torch.empty((ys.size(0), max_num_neighbors), dtype=torch.int64, device=xs.device)
        with _on(xs.device):
            check(_lib.load().gnnops_radius(xs.data_ptr(), ys.data_ptr(), px.data_ptr(), py.data_ptr(), px.numel() - 1, ys.size(0), xs.size(1),
                                            float(r), int(max_num_neighbors), out.data_ptr(), dt, _stream()), "radius")
        return out

    col = None
    if x.dtype == torch.float32 and x.size(1) <= 3 and 1 <= int(max_num_neighbors) <= 64 and x.size(0) < 2 ** 31 and float(r) >= 0:
        col = _by_cloud(x, y, int(max_num_neighbors), ptr_x, ptr_y, r, exhaustive)
    if col is None:
        col = exhaustive(x, y, ptr_x, ptr_y)
    return _pairs(col, max_num_neighbors)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target", num_workers=1):
    """torch_cluster.radius_graph (ops.txt:39)."""
    if flow not in ("source_to_target", "target_to_source"):
        raise ValueError(flow)
    ei = radius(x, x, r, batch, batch, max_num_neighbors if loop else max_num_neighbors + 1)
    row, col = (ei[1], ei[0]) if flow == "source_to_target" else (ei[0], ei[1])
    if not loop:
        mask = row != col
        row, col = row[mask], col[mask]
    return torch.stack([row, col], dim=0)


def nearest(x, y, batch_x=None, batch_y=None):
    """torch_cluster.nearest(x, y, batch_x, batch_y) (ops.txt:40): for every x the index of its nearest y of the same batch."""
    x, y = _points(x, "nearest"), _points(y, "nearest")
    dt = _dtype_code(x, "nearest")
    ptr_x, ptr_y = _pair_ptrs(x, y, batch_x, batch_y)
    col = torch.empty((x.size(0), 1), dtype=torch.int64, device=x.device)
    with _on(x.device):   # the k = 1 search with the roles swapped: queries are x, candidates y
        check(_lib.load().gnnops_knn(y.data_ptr(), x.data_ptr(), ptr_y.data_ptr(), ptr_x.data_ptr(), ptr_x.numel() - 1, x.size(0), x.size(1),
                                     1, 0, col.data_ptr(), dt, _stream()), "nearest")
    return col.view(-1)


def random_walk(row, col, start, walk_length, p=1.0, q=1.0, coalesced=True, num_nodes=None, seed=None):
    """torch_cluster.random_walk(row, col, start, walk_length, p=1, q=1, coalesced, num_nodes) (ops.txt:41): [len(start),
    walk_length + 1] node ids; each step moves to a drawn neighbour (a node without neighbours stays): uniformly for
    p = q = 1, with node2vec's return / in-out bias otherwise (rejection sampling as the package: a candidate is accepted with
    probability (1/p, 1, 1/q) / max(...) when it is the previous node / a neighbour of it / neither). ``seed`` (extra): fixes
    the draws; default from torch's RNG."""
    _require_gpu(row, col, start)
    if p <= 0 or q <= 0:
        raise ValueError("random_walk: p and q must be positive")
    biased = p != 1.0 or q != 1.0
    if num_nodes is None:
        num_nodes = int(max(row.max(), col.max())) + 1 if row.numel() else int(start.max()) + 1
    # CSR adjacency from the plan of `row` (stable: a row's neighbours keep the caller's order — all a uniform draw needs). The
    # biased walk tests "is x adjacent to t" by binary search, so there the entries are first ordered by (row, col).
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=row.device)
    if row.numel():
        row, col = row.contiguous(), col.contiguous()
        if biased:
            from .sparse import sort as _sort

            _, order = _sort(row * num_nodes + col, 0, stable=True)
            row, col = ops.index_select(row, 0, order), ops.index_select(col, 0, order)
        plan = ops.Plan(row, num_nodes, col)
        rowptr = plan.rowptr.to(torch.int64)
        col = plan.col if plan.col is not None else col[plan.perm[: row.numel()].long()]
    start = start.contiguous()
    out = torch.empty((start.numel(), walk_length + 1), dtype=torch.int64, device=row.device)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    colc = col.contiguous() if col.numel() else torch.zeros(1, dtype=torch.int64, device=row.device)
    with _on(row.device):
        if biased:
            check(_lib.load().gnnops_random_walk_node2vec(rowptr.data_ptr(), colc.data_ptr(), start.data_ptr(), start.numel(), int(walk_length),
                                                          float(p), float(q), int(seed), out.data_ptr(), _stream()), "random_walk_node2vec")
        else:
            check(_lib.load().gnnops_random_walk(rowptr.data_ptr(), colc.data_ptr(), start.data_ptr(), start.numel(), int(walk_length), int(seed),
                                                 out.data_ptr(), _stream()), "random_walk")
    return out


def graclus_cluster(row, col, weight=None, num_nodes=None, seed=None):
    """torch_cluster.graclus_cluster(row, col, weight=None, num_nodes=None) (ops.txt:35): pairs every node with one unmatched
    neighbour (the heaviest edge when weights are given); cluster[n] = min(n, partner), or n for a node left alone. The
    package's result depends on a random node order; here on ``seed`` (extra argument; default from torch's RNG) through a
    hash that breaks ties between equally heavy edges. Rounds run eight at a time between read-backs of one counter."""
    _require_gpu(row, col, weight)
    if num_nodes is None:
        num_nodes = int(max(row.max(), col.max())) + 1 if row.numel() else 0
    dev = row.device
    cluster = torch.full((num_nodes,), -1, dtype=torch.int64, device=dev)
    if num_nodes == 0:
        return cluster
    dt = _dtype_code(weight, "graclus_cluster") if weight is not None else 0
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=dev)
    w_csr = None
    if row.numel():
        plan = ops.Plan(row.contiguous(), num_nodes, col.contiguous())
        rowptr = plan.rowptr.to(torch.int64)
        order = plan.perm[: row.numel()].long()
        col = plan.col if plan.col is not None else col[order]
        if weight is not None:
            w_csr = weight.contiguous()[order].contiguous()
    colc = col.contiguous() if col.numel() else torch.zeros(1, dtype=torch.int64, device=dev)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    proposal = torch.empty(num_nodes, dtype=torch.int64, device=dev)
    active = torch.zeros(1, dtype=torch.int32, device=dev)
    L = _lib.load()
    while True:
        with _on(dev):
            check(L.gnnops_graclus_rounds(rowptr.data_ptr(), colc.data_ptr(), w_csr.data_ptr() if w_csr is not None else None, num_nodes,
                                          int(seed), 8, cluster.data_ptr(), proposal.data_ptr(), active.data_ptr(), 0, dt, _stream()),
                  "graclus_cluster")
        if int(active) == 0:
            break
    with _on(dev):
        check(L.gnnops_graclus_rounds(rowptr.data_ptr(), colc.data_ptr(), None, num_nodes, int(seed), 0, cluster.data_ptr(),
                                      proposal.data_ptr(), active.data_ptr(), 1, dt, _stream()), "graclus_cluster")
    return cluster
