"""GPU parity for the remaining ops of the reference's list (ops.txt:17-19, 29-41): torch_spline_conv's
spline_basis / spline_weighting / spline_conv and torch_cluster's grid_cluster / fps / knn(_graph) / radius(_graph) / nearest
/ random_walk, through the import seams (`from torch_spline_conv import ...`, `from torch_cluster import ...`).

PARITY UNPINNED (oracle/spatial_oracle.py header): neither package nor any output of it is in the reference tree; the oracle
restates the published definitions. Bars: integer outputs (weight_index, cluster ids, neighbour lists, sampled indices) exact
— inputs are drawn so that no two candidate distances tie within fp32 rounding, except where a test builds exact ties on
purpose to pin the smaller-index rule; floating outputs within 1e-5 (fp32) / 4e-3 (fp16) of the value scale.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ora():
    from oracle import spatial_oracle

    return spatial_oracle


def _f64(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _close(got, want, tol, what):
    got = _f64(got)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-6)
    assert err <= tol, f"{what}: {err:.3e} > {tol:.1e}"


# ---- torch_spline_conv ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("degree", [1, 2, 3])
@pytest.mark.parametrize("D,kernel_size,is_open", [(1, [5], [1]), (2, [5, 4], [1, 0]), (3, [3, 4, 5], [0, 1, 0])])
def test_spline_basis_and_weighting(ora, degree, D, kernel_size, is_open):
    from torch_spline_conv import spline_basis, spline_weighting

    if degree == 3 and D == 3:
        kernel_size = [4, 4, 5]                 # an open cubic spline needs kernel_size > degree
    g = torch.Generator().manual_seed(degree * 10 + D)
    E, Min, Mout = 257, 7, 70
    pseudo = torch.rand(E, D, generator=g)
    pseudo[0] = 0.0
    pseudo[1] = 1.0                              # the closed end wraps around: (floor(v) + k) mod kernel_size
    ks, op = torch.tensor(kernel_size), torch.tensor(is_open, dtype=torch.uint8)
    basis, wi = spline_basis(pseudo.cuda(), ks.cuda(), op.cuda(), degree)
    want_b, want_wi = ora.spline_basis(pseudo.numpy(), kernel_size, is_open, degree)
    assert basis.shape == (E, (degree + 1) ** D)
    assert np.array_equal(wi.cpu().numpy(), want_wi)
    _close(basis, want_b, 1e-5, "basis")
    assert torch.allclose(basis[2:].sum(1).cpu(), torch.ones(E - 2), atol=1e-5), "a B-spline basis is a partition of unity"
    K = int(np.prod(kernel_size))
    x = torch.rand(E, Min, generator=g) - 0.5
    weight = torch.rand(K, Min, Mout, generator=g) - 0.5
    for dtype, tol in ((torch.float32, 1e-5), (torch.float16, 4e-3)):
        out = spline_weighting(x.to(dtype).cuda(), weight.to(dtype).cuda(), basis.to(dtype), wi)
        want = ora.spline_weighting(_f64(x.to(dtype)), _f64(weight.to(dtype)), _f64(basis.to(dtype)), want_wi)
        _close(out, want, tol, f"weighting {dtype}")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 4e-3)])
@pytest.mark.parametrize("norm,root,bias", [(True, True, True), (False, False, False)])
def test_spline_conv(ora, dtype, tol, norm, root, bias):
    from torch_spline_conv import spline_conv

    g = torch.Generator().manual_seed(3)
    N, E, Min, Mout, D = 120, 900, 9, 33, 2
    ks, op = torch.tensor([5, 5]), torch.tensor([1, 0], dtype=torch.uint8)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[0][ei[0] == 7] = 8                        # row 7 receives nothing
    x = (torch.rand(N, Min, generator=g) - 0.5).to(dtype)
    pseudo = torch.rand(E, D, generator=g).to(dtype)
    weight = (torch.rand(25, Min, Mout, generator=g) - 0.5).to(dtype)
    rw = (torch.rand(Min, Mout, generator=g) - 0.5).to(dtype) if root else None
    b = (torch.rand(Mout, generator=g) - 0.5).to(dtype) if bias else None
    out = spline_conv(x.cuda(), ei.cuda(), pseudo.cuda(), weight.cuda(), ks, op, 1, norm, None if rw is None else rw.cuda(),
                      None if b is None else b.cuda())
    want = ora.spline_conv(_f64(x), ei.numpy(), _f64(pseudo), _f64(weight), [5, 5], [1, 0], 1, norm,
                           None if rw is None else _f64(rw), None if b is None else _f64(b))
    _close(out, want, tol, "spline_conv")
    if not (root or bias):
        assert bool((out[7] == 0).all())


# ---- torch_cluster ----------------------------------------------------------------------------------------------------
def _cloud(seed, n, d, batches):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, d, generator=g)
    batch = torch.sort(torch.randint(0, batches, (n,), generator=g)).values if batches > 1 else None
    return x, batch


def test_grid_cluster(ora):
    from torch_cluster import grid_cluster

    x, _ = _cloud(1, 1000, 3, 1)
    size = [0.21, 0.37, 0.5]
    got = grid_cluster(x.cuda(), torch.tensor(size).cuda())
    want = ora.grid_cluster(x.numpy(), size, x.min(0).values.numpy(), x.max(0).values.numpy())
    assert np.array_equal(got.cpu().numpy(), want)
    got = grid_cluster(x.cuda(), torch.tensor(size), torch.tensor([-0.5, 0.0, 0.0]), torch.tensor([1.5, 1.0, 2.0]))
    assert np.array_equal(got.cpu().numpy(), ora.grid_cluster(x.numpy(), size, [-0.5, 0.0, 0.0], [1.5, 1.0, 2.0]))
    # known answer: 2-D, unit voxels over [0, 3) x [0, 2): id = ix + 3 * iy
    pos = torch.tensor([[0.5, 0.5], [2.5, 0.5], [1.5, 1.5]])
    assert grid_cluster(pos.cuda(), torch.tensor([1.0, 1.0]), torch.tensor([0.0, 0.0]), torch.tensor([2.9, 1.9])).tolist() == [0, 2, 4]


@pytest.mark.parametrize("batches", [1, 5])
def test_fps(ora, batches):
    from torch_cluster import fps

    x, batch = _cloud(2, 700, 3, batches)
    got = fps(x.cuda(), None if batch is None else batch.cuda(), ratio=0.25, random_start=False)
    ptr = np.concatenate([[0], np.cumsum(np.bincount(batch.numpy(), minlength=batches))]) if batch is not None else np.array([0, 700])
    want = ora.fps(x.numpy(), None if batch is None else batch.numpy(), 0.25, ptr[:-1])
    assert np.array_equal(got.cpu().numpy(), want)
    rnd = fps(x.cuda(), None if batch is None else batch.cuda(), ratio=0.25, random_start=True)
    assert rnd.numel() == got.numel() and rnd.unique().numel() == rnd.numel()
    if batch is not None:                        # every batch keeps its share and its own points
        assert torch.equal(torch.bincount(batch[rnd.cpu()], minlength=batches), torch.ceil(torch.bincount(batch, minlength=batches) * 0.25).long())


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("D", [1, 2, 3, 5])
def test_fps_clouds_kept_in_registers(dt, D, monkeypatch):
    """Clouds of at most 8192 points in <= 3 dimensions keep their points and running distances in registers (csrc/cluster.hip
    fps_kernel); GNNOPS_FPS_REGISTERS=0 forces the general loop. The same samples in the same order: ragged clouds (1, 2, 63,
    1025, 8192 and 8193 points — the last takes the general loop either way), duplicated points (ties to the smaller
    index), D = 5 (general loop)."""
    from torch_cluster import fps

    g = torch.Generator().manual_seed(17 + D)
    sizes = [1, 2, 63, 1025, 8192, 8193, 700]
    x = torch.cat([torch.rand(n, D, generator=g) for n in sizes]).to(dt)
    x[70:100] = x[66:67]                         # a run of equal points inside the 1025-point cloud
    batch = torch.cat([torch.full((n,), b) for b, n in enumerate(sizes)])
    got = fps(x.cuda(), batch.cuda(), ratio=0.3, random_start=False)
    monkeypatch.setenv("GNNOPS_FPS_REGISTERS", "0")
    ref = fps(x.cuda(), batch.cuda(), ratio=0.3, random_start=False)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("batches,cosine", [(1, False), (4, False), (3, True)])
def test_knn_and_knn_graph(ora, batches, cosine):
    from torch_cluster import knn, knn_graph

    x, bx = _cloud(3, 300, 3, batches)
    y, by = _cloud(4, 90, 3, batches)
    k = 7
    got = knn(x.cuda(), y.cuda(), k, None if bx is None else bx.cuda(), None if by is None else by.cuda(), cosine=cosine)
    want = ora.knn(x.numpy(), y.numpy(), k, None if bx is None else bx.numpy(), None if by is None else by.numpy(), cosine)
    assert np.array_equal(got.cpu().numpy(), want)
    if not cosine:
        g = knn_graph(x.cuda(), 4, None if bx is None else bx.cuda())
        w = ora.knn(x.numpy(), x.numpy(), 5, None if bx is None else bx.numpy(), None if bx is None else bx.numpy())
        w = np.stack([w[1], w[0]])[:, w[0] != w[1]]     # flow source_to_target: (neighbour, point), self loops dropped
        assert np.array_equal(g.cpu().numpy(), w)
        assert torch.equal(torch.bincount(g[1].cpu(), minlength=300), torch.full((300,), 4))


def test_knn_fewer_candidates_than_k_and_exact_ties():
    from torch_cluster import knn

    x = torch.tensor([[0.0, 0.0], [2.0, 0.0], [0.0, 2.0], [5.0, 5.0]])
    bx = torch.tensor([0, 0, 0, 1])
    y = torch.tensor([[1.0, 1.0], [5.0, 4.0]])
    by = torch.tensor([0, 1])
    got = knn(x.cuda(), y.cuda(), 2, bx.cuda(), by.cuda())
    # y0 is equidistant from x0, x1, x2: the two smaller indices; y1's batch holds one point only
    assert got.cpu().tolist() == [[0, 0, 1], [0, 1, 3]]


@pytest.mark.parametrize("k", [1, 16, 64, 70])
@pytest.mark.parametrize("cosine", [False, True])
def test_knn_one_pass_equals_k_rounds(k, cosine, monkeypatch):
    """k <= 64 takes the one-pass kernel (the wave keeps its best 64 keys sorted across lanes; csrc/cluster.hip knn_topk_kernel),
    larger k the k-round kernel; GNNOPS_KNN_ROUNDS=1 forces the latter. Same pairs in the same order on a cloud with many
    exact ties (points on a coarse lattice, duplicated points), ragged batches, a batch with fewer than k points, and a point
    whose coordinates are NaN (never a neighbour)."""
    from torch_cluster import knn

    g = torch.Generator().manual_seed(40 + k)
    x = torch.randint(0, 6, (5000, 3), generator=g).float() / 4 + 0.25      # lattice: lots of equal distances
    x[100:140] = x[60:100]
    x[777] = float("nan")
    bx = torch.sort(torch.randint(0, 3, (5000,), generator=g)).values
    bx[-9:] = 3                                                              # batch 3: nine points
    y = torch.randint(0, 6, (700, 3), generator=g).float() / 4 + 0.3
    by = torch.sort(torch.randint(0, 4, (700,), generator=g)).values
    got = knn(x.cuda(), y.cuda(), k, bx.cuda(), by.cuda(), cosine=cosine)
    monkeypatch.setenv("GNNOPS_KNN_ROUNDS", "1")
    ref = knn(x.cuda(), y.cuda(), k, bx.cuda(), by.cuda(), cosine=cosine)
    assert torch.equal(got, ref)
    assert not bool((got[1] == 777).any())
    per_query = torch.bincount(got[0].cpu(), minlength=700)
    assert int(per_query[by == 3].max()) == min(k, 9)


@pytest.mark.parametrize("D", [1, 2, 3])
@pytest.mark.parametrize("k", [1, 9, 64])
def test_knn_through_the_grid_equals_the_exhaustive_kernel(D, k, monkeypatch):
    """One cloud of >= 8192 fp32 points in <= 3 dimensions is searched through a uniform grid (gnnops_knn_grid_cells -> plan ->
    gnnops_knn_grid_query): the same pairs in the same order as the exhaustive kernel, on clouds made to hurt — a dense clump
    beside empty space (many empty cells, long walks), a coarse lattice (exact ties across cell faces), duplicated points,
    a flat axis, NaN and infinite coordinates among x, queries outside the box / NaN, fewer finite candidates than k never
    (that needs a tiny cloud: covered by the exhaustive tests)."""
    from torch_cluster import knn
    from gnnops import spatial

    g = torch.Generator().manual_seed(50 + 10 * D + k)
    n = 20000
    x = torch.rand(n, D, generator=g)
    x[:6000] = x[:6000] * 0.02 + 0.4                       # clump
    x[6000:9000] = (torch.randint(0, 12, (3000, D), generator=g).float() / 12)   # lattice
    x[9000:9100] = x[8900:9000]                             # duplicates
    if D == 3:
        x[9100:12000, 2] = 0.5                              # a slab
    x[123] = float("nan")
    x[456, 0] = float("inf")
    y = torch.rand(900, D, generator=g) * 1.6 - 0.3         # a third of the queries outside the box
    y[:200] = x[6000:6200]                                  # queries ON lattice points
    y[5] = float("nan")
    calls = []
    real = spatial._knn_grid
    monkeypatch.setattr(spatial, "_knn_grid", lambda *a: (calls.append(1), real(*a))[1])
    got = knn(x.cuda(), y.cuda(), k)
    assert calls == [1], "the grid form did not take the call"
    monkeypatch.setattr(spatial, "_KNN_GRID_MIN_POINTS", 1 << 40)
    ref = knn(x.cuda(), y.cuda(), k)
    assert calls == [1]
    assert torch.equal(got, ref)
    assert not bool((got[1] == 123).any())
    # the same cloud far from the origin (coordinates around 3000: an ulp is a sizeable part of a cell) and queries far outside
    xo, yo = x + 3000.0, torch.cat([y[:300] + 3000.0, y[300:600] * 500.0 - 7000.0])
    monkeypatch.setattr(spatial, "_KNN_GRID_MIN_POINTS", 8192)
    got_o = knn(xo.cuda(), yo.cuda(), k)
    monkeypatch.setattr(spatial, "_KNN_GRID_MIN_POINTS", 1 << 40)
    assert torch.equal(got_o, knn(xo.cuda(), yo.cuda(), k))
    # a flat cloud (every point the same): one cell, every query walks it
    flat = torch.full((9000, D), 0.25)
    assert torch.equal(knn(flat.cuda(), y[:50].cuda(), k), knn(flat[:9000].cuda(), y[:50].cuda(), k))
    monkeypatch.setattr(spatial, "_KNN_GRID_MIN_POINTS", 8192)
    got_flat = knn(flat.cuda(), y[10:50].cuda(), k)
    assert got_flat[1].view(40, k).cpu().tolist() == [list(range(k))] * 40


def test_knn_and_radius_grid_per_cloud_under_a_batch_vector(monkeypatch):
    """A batch vector over a few clouds of which some are large: the large ones walk a grid of their own, the small ones (and an
    empty one) the exhaustive kernel, cloud by cloud; the pairs are those of the single exhaustive call over all of them."""
    from torch_cluster import knn, radius
    from gnnops import spatial

    g = torch.Generator().manual_seed(91)
    sizes_x = [9000, 300, 0, 12000, 50]
    sizes_y = [400, 77, 5, 300, 0]
    x = torch.cat([torch.rand(n, 3, generator=g) * (1 + b) for b, n in enumerate(sizes_x)])
    y = torch.cat([torch.rand(n, 3, generator=g) * (1 + b) for b, n in enumerate(sizes_y)])
    bx = torch.cat([torch.full((n,), b) for b, n in enumerate(sizes_x)])
    by = torch.cat([torch.full((n,), b) for b, n in enumerate(sizes_y)])
    calls = []
    real = spatial._knn_grid
    monkeypatch.setattr(spatial, "_knn_grid", lambda *a, **kw: (calls.append(a[0].size(0)), real(*a, **kw))[1])
    got_k = knn(x.cuda(), y.cuda(), 12, bx.cuda(), by.cuda())
    got_r = radius(x.cuda(), y.cuda(), 0.15, bx.cuda(), by.cuda(), max_num_neighbors=20)
    assert calls == [9000, 12000, 9000, 12000]
    monkeypatch.setattr(spatial, "_KNN_GRID_MIN_POINTS", 1 << 40)
    assert torch.equal(got_k, knn(x.cuda(), y.cuda(), 12, bx.cuda(), by.cuda()))
    assert torch.equal(got_r, radius(x.cuda(), y.cuda(), 0.15, bx.cuda(), by.cuda(), max_num_neighbors=20))
    assert len(calls) == 4


@pytest.mark.parametrize("D", [1, 2, 3])
@pytest.mark.parametrize("r,cap", [(0.03, 32), (0.11, 64), (0.5, 7), (0.0, 4)])
def test_radius_through_the_grid_equals_the_exhaustive_kernel(D, r, cap, monkeypatch):
    """torch_cluster.radius on one cloud of >= 8192 fp32 points walks the same grid: the `cap` smallest indices inside the ball,
    ascending — the exhaustive kernel's answer — for balls smaller than a cell, spanning many cells, covering half the cloud,
    and empty (r = 0: the comparison is strict)."""
    from torch_cluster import radius
    from gnnops import spatial

    g = torch.Generator().manual_seed(70 + D)
    n = 15000
    x = torch.rand(n, D, generator=g)
    x[:4000] = x[:4000] * 0.05 + 0.2
    x[4000:6000] = torch.randint(0, 10, (2000, D), generator=g).float() / 10
    x[77] = float("nan")
    y = torch.rand(500, D, generator=g) * 1.4 - 0.2
    y[:100] = x[4000:4100]
    calls = []
    real = spatial._knn_grid
    monkeypatch.setattr(spatial, "_knn_grid", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    got = radius(x.cuda(), y.cuda(), r, max_num_neighbors=cap)
    assert calls == [1]
    monkeypatch.setattr(spatial, "_KNN_GRID_MIN_POINTS", 1 << 40)
    ref = radius(x.cuda(), y.cuda(), r, max_num_neighbors=cap)
    assert calls == [1] and torch.equal(got, ref)
    if r == 0.0:
        assert got.numel() == 0


@pytest.mark.parametrize("batches", [1, 4])
def test_radius_and_radius_graph(ora, batches):
    from torch_cluster import radius, radius_graph

    x, bx = _cloud(5, 400, 3, batches)
    y, by = _cloud(6, 60, 3, batches)
    for r, cap in ((0.25, 32), (0.6, 5)):
        got = radius(x.cuda(), y.cuda(), r, None if bx is None else bx.cuda(), None if by is None else by.cuda(), max_num_neighbors=cap)
        want = ora.radius(x.numpy(), y.numpy(), r, None if bx is None else bx.numpy(), None if by is None else by.numpy(), cap)
        assert np.array_equal(got.cpu().numpy(), want), (r, cap)
    g = radius_graph(x.cuda(), 0.2, None if bx is None else bx.cuda(), max_num_neighbors=64)
    w = ora.radius(x.numpy(), x.numpy(), 0.2, None if bx is None else bx.numpy(), None if bx is None else bx.numpy(), 65)
    w = np.stack([w[1], w[0]])[:, w[0] != w[1]]
    assert np.array_equal(g.cpu().numpy(), w)


def test_nearest(ora):
    from torch_cluster import nearest

    x, bx = _cloud(7, 500, 2, 3)
    y, by = _cloud(8, 40, 2, 3)
    got = nearest(x.cuda(), y.cuda(), bx.cuda(), by.cuda())
    assert np.array_equal(got.cpu().numpy(), ora.nearest(x.numpy(), y.numpy(), bx.numpy(), by.numpy()))
    got = nearest(x.cuda(), y.cuda())
    assert np.array_equal(got.cpu().numpy(), ora.nearest(x.numpy(), y.numpy()))


def test_random_walk_properties():
    """Random by nature: every step goes to a neighbour of the current node, a node without neighbours stays, the same
    seed gives the same walks, and the draws are close to uniform over a node's neighbours."""
    from torch_cluster import random_walk

    g = torch.Generator().manual_seed(9)
    n, e = 200, 1500
    row, col = torch.randint(0, n - 1, (e,), generator=g), torch.randint(0, n, (e,), generator=g)   # node n-1 has no out-edge
    start = torch.arange(n).repeat(4)
    walks = random_walk(row.cuda(), col.cuda(), start.cuda(), 12, num_nodes=n, seed=5).cpu()
    assert walks.shape == (4 * n, 13) and torch.equal(walks[:, 0], start)
    adj = torch.zeros(n, n, dtype=torch.bool)
    adj[row, col] = True
    has_out = adj.any(1)
    a, b = walks[:, :-1].reshape(-1), walks[:, 1:].reshape(-1)
    assert bool((adj[a, b] | (~has_out[a] & (a == b))).all())
    assert torch.equal(walks, random_walk(row.cuda(), col.cuda(), start.cuda(), 12, num_nodes=n, seed=5).cpu())
    assert not torch.equal(walks, random_walk(row.cuda(), col.cuda(), start.cuda(), 12, num_nodes=n, seed=6).cpu())
    # uniformity: a node with 4 distinct neighbours, 40 000 single steps
    r2, c2 = torch.tensor([0, 0, 0, 0]), torch.tensor([1, 2, 3, 4])
    one = random_walk(r2.cuda(), c2.cuda(), torch.zeros(40000, dtype=torch.int64).cuda(), 1, num_nodes=5, seed=1)[:, 1].cpu()
    freq = torch.bincount(one, minlength=5)[1:].float() / 40000
    assert bool(((freq - 0.25).abs() < 0.01).all()), freq
    with pytest.raises(ValueError):
        random_walk(row.cuda(), col.cuda(), start.cuda(), 3, p=0.0)


@pytest.mark.parametrize("p,q", [(4.0, 1.0), (0.25, 2.0), (1.0, 0.25), (2.0, 4.0)])
def test_random_walk_node2vec_bias(p, q):
    """torch_cluster.random_walk(p, q) (ops.txt:41; node2vec): second-order walk. From t = 0, via v = 1, the next node x is
    drawn with weight 1/p if x == t, 1 if x is adjacent to t, 1/q otherwise. Small undirected graph where all three kinds
    exist among v's neighbours; 60 000 walks; empirical frequencies of the third node against the exact distribution, and
    every step along an edge. Random by nature in the package too: a property test, parity unpinned."""
    from torch_cluster import random_walk

    #   0 - 1, 0 - 2, 1 - 2, 1 - 3, 1 - 4, 2 - 4   (both directions). From 0 via 1: back to 0 (1/p), 2 (adjacent to 0: 1), 3 and 4 (1/q)
    und = [(0, 1), (0, 2), (1, 2), (1, 3), (1, 4), (2, 4)]
    row = torch.tensor([a for a, b in und] + [b for a, b in und])
    col = torch.tensor([b for a, b in und] + [a for a, b in und])
    perm = torch.randperm(row.numel(), generator=torch.Generator().manual_seed(3))      # the caller's order is arbitrary
    row, col = row[perm], col[perm]
    S = 60000
    walks = random_walk(row.cuda(), col.cuda(), torch.zeros(S, dtype=torch.int64).cuda(), 2, p=p, q=q, num_nodes=5, seed=11).cpu()
    adj = torch.zeros(5, 5, dtype=torch.bool)
    adj[row, col] = True
    assert bool(adj[walks[:, 0], walks[:, 1]].all()) and bool(adj[walks[:, 1], walks[:, 2]].all())
    first = torch.bincount(walks[:, 1], minlength=5).float() / S                         # first step uniform over {1, 2}
    assert abs(float(first[1]) - 0.5) < 0.01 and abs(float(first[2]) - 0.5) < 0.01
    via1 = walks[walks[:, 1] == 1, 2]
    w = {0: 1 / p, 2: 1.0, 3: 1 / q, 4: 1 / q}
    z = sum(w.values())
    freq = torch.bincount(via1, minlength=5).float() / via1.numel()
    for x, wx in w.items():
        assert abs(float(freq[x]) - wx / z) < 0.012, (x, float(freq[x]), wx / z)
    assert float(freq[1]) == 0.0
    assert torch.equal(walks, random_walk(row.cuda(), col.cuda(), torch.zeros(S, dtype=torch.int64).cuda(), 2, p=p, q=q, num_nodes=5, seed=11).cpu())


@pytest.mark.parametrize("weighted", [False, True])
def test_graclus_cluster_is_a_maximal_matching(weighted):
    """Random by nature (the package shuffles the nodes): the result must be a MAXIMAL matching of the graph — every cluster is
    a node alone or two adjacent nodes named by the smaller id, and no two nodes left alone are adjacent — reproducible per
    seed; with weights, a node's partner is never lighter than an edge to a node that was left alone."""
    from torch_cluster import graclus_cluster

    g = torch.Generator().manual_seed(11)
    n, e = 500, 1800
    a, b = torch.randint(0, n - 3, (e,), generator=g), torch.randint(0, n - 3, (e,), generator=g)   # the last 3 nodes are isolated
    row, col = torch.cat([a, b]), torch.cat([b, a])                                                   # symmetric, with some self loops
    w = torch.rand(e, generator=g)
    weight = torch.cat([w, w]) if weighted else None
    cl = graclus_cluster(row.cuda(), col.cuda(), None if weight is None else weight.cuda(), n, seed=3).cpu()
    assert cl.shape == (n,) and bool((cl <= torch.arange(n)).all())
    sizes = torch.bincount(cl, minlength=n)
    assert int(sizes.max()) <= 2 and bool((cl[cl] == cl).all())
    adj = torch.zeros(n, n, dtype=torch.bool)
    adj[row, col] = True
    adj.fill_diagonal_(False)
    pairs = torch.nonzero(cl != torch.arange(n)).view(-1)
    assert bool(adj[pairs, cl[pairs]].all()), "partners must be neighbours"
    single = sizes[cl] == 1
    assert not bool(adj[single][:, single].any()), "two adjacent nodes were both left alone: the matching is not maximal"
    assert torch.equal(cl[-3:], torch.arange(n - 3, n))
    assert torch.equal(cl, graclus_cluster(row.cuda(), col.cuda(), None if weight is None else weight.cuda(), n, seed=3).cpu())
    if not weighted:
        assert not torch.equal(cl, graclus_cluster(row.cuda(), col.cuda(), None, n, seed=4).cpu())
    # known answer with weights: path 0 -1- 1 -5- 2 -1- 3: the heavy middle edge is taken, the ends stay alone
    r = torch.tensor([0, 1, 1, 2, 2, 3]).cuda()
    c = torch.tensor([1, 0, 2, 1, 3, 2]).cuda()
    ww = torch.tensor([1.0, 1.0, 5.0, 5.0, 1.0, 1.0]).cuda()
    assert graclus_cluster(r, c, ww, 4, seed=0).cpu().tolist() == [0, 1, 1, 3]


def test_spatial_ops_refuse_cpu_tensors():
    from torch_cluster import knn
    from torch_spline_conv import spline_basis

    with pytest.raises((RuntimeError, ValueError)):
        knn(torch.rand(5, 2), torch.rand(3, 2), 2)
    with pytest.raises((RuntimeError, ValueError)):
        spline_basis(torch.rand(5, 2), torch.tensor([3, 3]), torch.tensor([1, 1], dtype=torch.uint8), 1)
