"""CPU oracle for the spline-convolution and clustering ops of the reference's list (ops.txt:17-19, 29-41; SURVEY.md §8f
rank 4). TEST INFRASTRUCTURE ONLY: only ``tests/`` may import this module; the product package never does.

PARITY UNPINNED: torch-spline-conv 1.2.1 and torch-cluster 1.5.9 (requirements.txt:214, :210) are neither in /root/reference
nor installed here, and the reference holds no script, output or fixture for these ops. The functions restate the packages'
published definitions in float64 numpy loops (small cases only):
  spline_basis / spline_weighting / spline_conv   SplineCNN (Fey et al., CVPR 2018) closed B-splines of degree 1-3 over
      pseudo-coordinates in [0, 1]; open splines use kernel_size - degree intervals; the product basis over dimensions;
      spline_conv sums at edge_index[0] the weighted x[edge_index[1]], mean-normalised by degree, + root + bias.
  grid_cluster   voxel index = sum_d trunc((pos_d - start_d) / size_d) * prod_{d' < d} (trunc((end - start) / size) + 1)
  fps            iterative farthest point (squared Euclidean), first maximum on ties
  knn            k smallest (distance, index) per query inside its batch; radius: ascending index, distance^2 < r^2, capped
  nearest        argmin over the batch's candidates
"""
import itertools

import numpy as np


def _bspline(v, k, m):
    if m == 1:
        return 1 - v if k == 0 else v
    if m == 2:
        return (0.5 * v * v - v + 0.5, -v * v + v + 0.5, 0.5 * v * v)[k]
    return ((1 - v) ** 3 / 6, (3 * v ** 3 - 6 * v * v + 4) / 6, (-3 * v ** 3 + 3 * v * v + 3 * v + 1) / 6, v ** 3 / 6)[k]


def spline_basis(pseudo, kernel_size, is_open_spline, degree):
    pseudo = np.asarray(pseudo, np.float64)
    E, D = pseudo.shape
    S = (degree + 1) ** D
    basis = np.ones((E, S))
    wi = np.zeros((E, S), np.int64)
    for e in range(E):
        for s, ks in enumerate(itertools.product(range(degree + 1), repeat=D)):
            ks = ks[::-1]                                  # dimension 0 is the fastest digit of s
            off = 1
            for d in range(D):
                v = pseudo[e, d] * (int(kernel_size[d]) - degree * int(is_open_spline[d]))
                fl = np.floor(v)
                wi[e, s] += ((int(fl) + ks[d]) % int(kernel_size[d])) * off
                off *= int(kernel_size[d])
                basis[e, s] *= _bspline(v - fl, ks[d], degree)
    return basis, wi


def spline_weighting(x, weight, basis, weight_index):
    x, weight, basis = (np.asarray(t, np.float64) for t in (x, weight, basis))
    out = np.zeros((x.shape[0], weight.shape[2]))
    for e in range(x.shape[0]):
        for s in range(basis.shape[1]):
            out[e] += basis[e, s] * (x[e] @ weight[weight_index[e, s]])
    return out


def spline_conv(x, edge_index, pseudo, weight, kernel_size, is_open_spline, degree=1, norm=True, root_weight=None, bias=None):
    x = np.asarray(x, np.float64)
    row, col = edge_index
    basis, wi = spline_basis(pseudo, kernel_size, is_open_spline, degree)
    msg = spline_weighting(x[col], weight, basis, wi)
    out = np.zeros((x.shape[0], msg.shape[1]))
    np.add.at(out, row, msg)
    if norm:
        deg = np.maximum(np.bincount(row, minlength=x.shape[0]), 1).reshape(-1, 1)
        out = out / deg
    if root_weight is not None:
        out = out + x @ np.asarray(root_weight, np.float64)
    if bias is not None:
        out = out + np.asarray(bias, np.float64)
    return out


def grid_cluster(pos, size, start, end, dtype=np.float32):
    pos = np.asarray(pos, dtype)
    size, start, end = (np.asarray(t, dtype) for t in (size, start, end))
    c = ((pos - start) / size).astype(np.int64)
    nvox = ((end - start) / size).astype(np.int64) + 1
    mult = np.concatenate([[1], np.cumprod(nvox)[:-1]])
    return (c * mult).sum(1)


def _segments(batch, n):
    if batch is None:
        return [np.arange(n)]
    return [np.nonzero(np.asarray(batch) == b)[0] for b in range(int(np.max(batch)) + 1)]


def fps(x, batch, ratio, start):
    """start: first index of every batch (absolute)."""
    x = np.asarray(x, np.float32)
    out = []
    for b, seg in enumerate(_segments(batch, len(x))):
        if len(seg) == 0:
            continue
        k = int(np.ceil(len(seg) * ratio))
        cur = int(start[b])
        dist = np.full(len(seg), np.inf, np.float32)
        chosen = [cur]
        for _ in range(k - 1):
            d = ((x[seg] - x[cur]) ** 2).sum(1, dtype=np.float32)
            dist = np.minimum(dist, d)
            cur = int(seg[np.argmax(dist)])
            chosen.append(cur)
        out += chosen
    return np.array(out, np.int64)


def _dist(a, b, cosine):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if cosine:
        return 1 - (a @ b) / (np.linalg.norm(a) * np.linalg.norm(b))
    return ((a - b) ** 2).sum()


def knn(x, y, k, batch_x=None, batch_y=None, cosine=False):
    rows, cols = [], []
    for j in range(len(y)):
        cand = np.arange(len(x)) if batch_x is None else np.nonzero(np.asarray(batch_x) == batch_y[j])[0]
        d = np.array([_dist(x[i], y[j], cosine) for i in cand])
        order = np.lexsort((cand, d))[:k]
        rows += [j] * len(order)
        cols += list(cand[order])
    return np.array([rows, cols], np.int64)


def radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32):
    rows, cols = [], []
    for j in range(len(y)):
        cand = np.arange(len(x)) if batch_x is None else np.nonzero(np.asarray(batch_x) == batch_y[j])[0]
        hit = [i for i in cand if ((np.asarray(x[i], np.float32) - np.asarray(y[j], np.float32)) ** 2).sum(dtype=np.float32) < np.float32(r * r)]
        hit = hit[:max_num_neighbors]
        rows += [j] * len(hit)
        cols += hit
    return np.array([rows, cols], np.int64)


def nearest(x, y, batch_x=None, batch_y=None):
    out = np.zeros(len(x), np.int64)
    for i in range(len(x)):
        cand = np.arange(len(y)) if batch_x is None else np.nonzero(np.asarray(batch_y) == batch_x[i])[0]
        d = np.array([_dist(y[c], x[i], False) for c in cand])
        out[i] = cand[np.lexsort((cand, d))[0]]
    return out
