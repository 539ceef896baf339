// cluster.hip — the torch_cluster ops the reference lists among what it means to benchmark (ops.txt:33-41: grid_cluster,
// fps, knn_graph, radius_graph, nearest, random_walk) — SURVEY.md §8(f) rank 4. torch-cluster 1.5.9 (requirements.txt:210)
// is not in the reference tree: the definitions below are the package's published ones, parity unpinned, oracle in
// oracle/spatial_oracle.py. Batches are given as CSR pointers over points sorted by batch (ptr [B + 1], int64), the form
// the package itself converts `batch` vectors to.
//
// All of these are small-D geometry (D = 2, 3 coordinates; at most a few hundred thousand points per call) where the
// distance evaluation is a handful of flops on operands that sit in L2: brute force, one WAVE per query so that the 64
// lanes sweep the candidates of the query's batch segment together, no data structure to build, and every choice
// deterministic — ties go to the smaller index, neighbour lists come out in the package's order (knn: ascending distance;
// radius: ascending candidate index, first max_num_neighbors).
#include "common.h"
#include <stdlib.h>

namespace {

template <typename T>
__device__ inline float dist2(const T* __restrict__ a, const T* __restrict__ b, int D) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) {
        const float t = Elem<T>::load(a + d) - Elem<T>::load(b + d);
        s += t * t;
    }
    return s;
}
// 1 - cos(a, b), as knn(..., cosine=True)
template <typename T>
__device__ inline float cos_dist(const T* __restrict__ a, const T* __restrict__ b, int D) {
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (int d = 0; d < D; ++d) {
        const float u = Elem<T>::load(a + d), v = Elem<T>::load(b + d);
        ab += u * v; aa += u * u; bb += v * v;
    }
    return 1.f - ab / (sqrtf(aa) * sqrtf(bb));
}

// segment b with ptr[b] <= i < ptr[b + 1] (B small: binary search over the pointer)
__device__ inline int segment_of(const int64_t* __restrict__ ptr, int B, int64_t i) {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// lexicographic (distance, index) minimum across the wave
__device__ inline void wave_min_pair(float& d, int64_t& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float od = __shfl_xor(d, o);
        const int64_t oi = __shfl_xor(i, o);
        if (od < d || (od == d && oi < i)) { d = od; i = oi; }
    }
}

// ---- grid_cluster: voxel id of every point ----
template <typename T>
__global__ void grid_cluster_kernel(const T* __restrict__ pos, int64_t N, int D, const double* __restrict__ size,
                                    const double* __restrict__ start, const double* __restrict__ end, int64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = 0, k = 1;
        for (int d = 0; d < D; ++d) {
            // the package computes in pos's own type: (pos - start) / size truncated; voxels per axis = trunc((end - start) / size) + 1
            const float p = Elem<T>::load(pos + i * D + d);
            c += (int64_t)((p - (float)start[d]) / (float)size[d]) * k;
            k *= (int64_t)(((float)end[d] - (float)start[d]) / (float)size[d]) + 1;
        }
        out[i] = c;
    }
}

// ---- knn / nearest: wave per query, k rounds of "smallest (distance, index) after the previous pick" ----
template <typename T, bool COSINE>
__global__ __launch_bounds__(256) void knn_kernel(const T* __restrict__ x, const T* __restrict__ y, const int64_t* __restrict__ ptr_x,
                                                  const int64_t* __restrict__ ptr_y, int B, int64_t Ny, int D, int k,
                                                  int64_t* __restrict__ col) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t qy = wave0; qy < Ny; qy += nwaves) {
        const int b = segment_of(ptr_y, B, qy);
        const int64_t xb = ptr_x[b], xe = ptr_x[b + 1];
        const T* yq = y + qy * D;
        float last_d = -__builtin_huge_valf();
        int64_t last_i = -1;
        for (int r = 0; r < k; ++r) {
            float best_d = __builtin_huge_valf();
            int64_t best_i = INT64_MAX;
            for (int64_t i = xb + lane; i < xe; i += 64) {
                const float d = COSINE ? cos_dist<T>(x + i * D, yq, D) : dist2<T>(x + i * D, yq, D);
                const bool after = d > last_d || (d == last_d && i > last_i);
                if (after && (d < best_d || (d == best_d && i < best_i))) { best_d = d; best_i = i; }
            }
            wave_min_pair(best_d, best_i);
            if (lane == 0) col[qy * k + r] = best_i == INT64_MAX ? -1 : best_i;   // fewer than k candidates: -1 from here on
            last_d = best_d;
            last_i = best_i;
            if (best_i == INT64_MAX) {
                for (int r2 = r + 1 + lane; r2 < k; r2 += 64) col[qy * k + r2] = -1;
                break;
            }
        }
    }
}

// ---- knn, k <= 64: ONE pass over the candidates. The wave keeps the best 64 (distance, index) keys seen so far SORTED
// ACROSS ITS LANES (lane j holds the j-th smallest); a batch of 64 candidates is compared with the k-th key in one go and
// the few that beat it are inserted one by one (rank by ballot, shift by one lane). A uniform stream of n candidates makes
// about k (1 + ln(n / k)) insertions, so a query costs n / 64 distance evaluations per lane instead of k n / 64: the k rounds of
// knn_kernel above compute every distance k times. Keys are strictly ordered (distance image, then index), so the result is
// the brute-force one: nearest first, ties to the smaller index, NaN distances never chosen.
__device__ inline uint32_t knn_order(float v) {
    uint32_t u = __float_as_uint(v);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// lane `src` (the same in every lane) of a 64-bit value: two v_readlane, no trip through the LDS crossbar
__device__ inline uint64_t shfl_u64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
// the value of the lane below (lane 0 keeps its own): one DPP wave shift per half
__device__ inline uint64_t shfl_up_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);          // wave_shr:1
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
template <typename T, bool COSINE>
__global__ __launch_bounds__(256) void knn_topk_kernel(const T* __restrict__ x, const T* __restrict__ y, const int64_t* __restrict__ ptr_x,
                                                       const int64_t* __restrict__ ptr_y, int B, int64_t Ny, int D, int k,
                                                       int64_t* __restrict__ col) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    constexpr uint64_t NONE = ~(uint64_t)0;
    for (int64_t qy = wave0; qy < Ny; qy += nwaves) {
        const int b = segment_of(ptr_y, B, qy);
        const int64_t xb = ptr_x[b], xe = ptr_x[b + 1];
        const T* yq = y + qy * D;
        uint64_t mine = NONE, kth = NONE;
        for (int64_t base = xb; base < xe; base += 64) {
            const int64_t i = base + lane;
            uint64_t key = NONE;
            if (i < xe) {
                const float d = COSINE ? cos_dist<T>(x + i * D, yq, D) : dist2<T>(x + i * D, yq, D);
                if (d == d) key = ((uint64_t)knn_order(d) << 32) | (uint32_t)(i - xb);
            }
            uint64_t todo = __ballot(key < kth);
            while (todo) {
                const int src = __builtin_ctzll(todo);
                todo &= todo - 1;
                const uint64_t c = shfl_u64(key, src);
                if (!(c < kth)) continue;                          // the bound has tightened since the ballot
                const int p = __popcll(__ballot(mine < c));       // sorted: the smaller keys are lanes 0 .. p-1
                const uint64_t up = shfl_up_u64(mine);
                mine = lane < p ? mine : (lane == p ? c : up);
                kth = shfl_u64(mine, k - 1);
            }
        }
        if (lane < k) col[qy * k + lane] = mine == NONE ? (int64_t)-1 : xb + (int64_t)(uint32_t)mine;
    }
}

// ---- knn through a uniform grid (one cloud, Euclidean, D <= 3, fp32, k <= 64) ------------------------------------------------
// torch_cluster's GPU knn is exhaustive (and so are the two kernels above); its CPU path uses a tree. Here: the points are
// binned into G^D cells of their bounding box (gnnops_knn_grid_cells), sorted by cell with the plan builder (cell = "destination":
// rowptr = first point of every cell, perm = points in cell order), and a query walks the cells around its own in shells of
// growing Chebyshev radius s — a row of cells along x is one contiguous range of perm — feeding the same wave-sorted best-64
// list. It stops when the k-th distance is below the distance to the nearest face of the (2 s + 1)^D block that is not the
// grid's own edge (less a margin for points the float cell arithmetic put one cell off), or when the block is the whole grid.
// Same distances, same keys: the pairs are those of the exhaustive kernels, ties and all.
struct KnnGrid {
    float lo[3], h[3];   // box origin and cell size per axis (h = 0: the axis is flat, one cell)
    int g[3];            // cells per axis (1 for axes beyond D)
};

__global__ void knn_bbox_kernel(const float* __restrict__ x, int64_t n, int D, unsigned* __restrict__ box) {
    // box[0..2] = min image, box[3..5] = max image per axis (knn_order images: unsigned order == float order); non-finite skipped
    float lo[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()};
    float hi[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int d = 0; d < D; ++d) {
            const float v = x[i * D + d];
            if (v - v == 0.f) { lo[d] = v < lo[d] ? v : lo[d]; hi[d] = v > hi[d] ? v : hi[d]; }
        }
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float a = __shfl_xor(lo[d], o), b = __shfl_xor(hi[d], o);
            lo[d] = a < lo[d] ? a : lo[d];
            hi[d] = b > hi[d] ? b : hi[d];
        }
        if ((threadIdx.x & 63) == 0) {
            if (lo[d] <= hi[d]) {
                atomicMin(box + d, knn_order(lo[d]));
                atomicMax(box + 3 + d, knn_order(hi[d]));
            }
        }
    }
}

__device__ inline float knn_unorder(uint32_t img) { return __uint_as_float((img & 0x80000000u) ? (img & 0x7fffffffu) : ~img); }

__device__ inline KnnGrid knn_grid_of(const unsigned* __restrict__ box, int D, int G) {
    KnnGrid g;
    for (int d = 0; d < 3; ++d) {
        g.lo[d] = 0.f; g.h[d] = 0.f; g.g[d] = 1;
        if (d < D) {
            const float lo = knn_unorder(box[d]), hi = knn_unorder(box[3 + d]);
            if (lo <= hi) {       // at least one finite coordinate on this axis
                g.lo[d] = lo;
                g.h[d] = (hi - lo) / (float)G;
                g.g[d] = g.h[d] > 0.f ? G : 1;
            }
        }
    }
    return g;
}
__device__ inline int knn_cell_of(const KnnGrid& g, int d, float v) {
    if (g.g[d] == 1) return 0;
    const float t = (v - g.lo[d]) / g.h[d];
    if (!(t >= 0.f)) return 0;                     // below the box, or NaN
    const int c = t >= (float)g.g[d] ? g.g[d] - 1 : (int)t;
    return c;
}

__global__ void knn_cells_kernel(const float* __restrict__ x, int64_t n, int D, int G, const unsigned* __restrict__ box,
                                 int64_t* __restrict__ cell) {
    const KnnGrid g = knn_grid_of(box, D, G);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int c[3] = {0, 0, 0};
        for (int d = 0; d < D; ++d) c[d] = knn_cell_of(g, d, x[i * D + d]);
        cell[i] = ((int64_t)c[2] * g.g[1] + c[1]) * g.g[0] + c[0];
    }
}

// RADIUS: the same walk for torch_cluster.radius — the key is the index alone among the points with |x - y|^2 < r2, so the list
// ends up holding the k = max_num_neighbors SMALLEST indices inside the ball (what the exhaustive kernel's index-order scan
// returns), and the walk ends when no point outside the block can be inside the ball.
template <bool RADIUS>
__global__ __launch_bounds__(256) void knn_grid_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t Ny, int D, int k, int G,
                                                       const unsigned* __restrict__ box, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ perm, int64_t* __restrict__ col, float r2) {
    const KnnGrid g = knn_grid_of(box, D, G);
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    constexpr uint64_t NONE = ~(uint64_t)0;
    for (int64_t qy = wave0; qy < Ny; qy += nwaves) {
        const float* yq = y + qy * D;
        float q[3] = {0.f, 0.f, 0.f};
        int cq[3] = {0, 0, 0};
        for (int d = 0; d < D; ++d) { q[d] = yq[d]; cq[d] = knn_cell_of(g, d, q[d]); }
        uint64_t mine = NONE, kth = NONE;
        auto feed_range = [&](int c_lo, int c_hi, int cy, int cz) {   // cells c_lo .. c_hi of row (cy, cz): one range of perm
            const int64_t row = ((int64_t)cz * g.g[1] + cy) * g.g[0];
            const int p0 = rowptr[row + c_lo], p1 = rowptr[row + c_hi + 1];
            for (int base = p0; base < p1; base += 64) {
                const int p = base + lane;
                uint64_t key = NONE;
                if (p < p1) {
                    const int i = perm[p];
                    const float d = dist2<float>(x + (int64_t)i * D, yq, D);
                    if constexpr (RADIUS) { if (d < r2) key = (uint64_t)(uint32_t)i; }
                    else if (d == d) key = ((uint64_t)knn_order(d) << 32) | (uint32_t)i;
                }
                uint64_t todo = __ballot(key < kth);
                while (todo) {
                    const int src = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const uint64_t c = shfl_u64(key, src);
                    if (!(c < kth)) continue;
                    const int pos = __popcll(__ballot(mine < c));
                    const uint64_t up = shfl_up_u64(mine);
                    mine = lane < pos ? mine : (lane == pos ? c : up);
                    kth = shfl_u64(mine, k - 1);
                }
            }
        };
        const int smax = (g.g[0] > g.g[1] ? (g.g[0] > g.g[2] ? g.g[0] : g.g[2]) : (g.g[1] > g.g[2] ? g.g[1] : g.g[2]));
        for (int s = 0; s < smax; ++s) {
            const int z0 = cq[2] - s < 0 ? 0 : cq[2] - s, z1 = cq[2] + s > g.g[2] - 1 ? g.g[2] - 1 : cq[2] + s;
            const int y0 = cq[1] - s < 0 ? 0 : cq[1] - s, y1 = cq[1] + s > g.g[1] - 1 ? g.g[1] - 1 : cq[1] + s;
            const int x0 = cq[0] - s < 0 ? 0 : cq[0] - s, x1 = cq[0] + s > g.g[0] - 1 ? g.g[0] - 1 : cq[0] + s;
            for (int cz = z0; cz <= z1; ++cz)
                for (int cy = y0; cy <= y1; ++cy) {
                    const int az = cz > cq[2] ? cz - cq[2] : cq[2] - cz, ay = cy > cq[1] ? cy - cq[1] : cq[1] - cy;
                    if ((az > ay ? az : ay) == s) {
                        feed_range(x0, x1, cy, cz);            // a row of the shell's outer faces: all of it is new
                    } else {                                    // an inner row: only its two end cells are new
                        if (cq[0] - s >= 0) feed_range(cq[0] - s, cq[0] - s, cy, cz);
                        if (cq[0] + s <= g.g[0] - 1) feed_range(cq[0] + s, cq[0] + s, cy, cz);
                    }
                }
            // done? the block covers the grid, or nothing outside it can beat the k-th key
            bool whole = true;
            float bound = __builtin_huge_valf();
            for (int d = 0; d < 3; ++d) {
                if (g.g[d] == 1) continue;
                // box-relative coordinates, as the cell arithmetic uses them: a cloud far from the origin (coordinates 1000.0 ..
                // 1000.1) would lose the faces to the rounding of lo + c h. The margin covers a point the float division put
                // one cell off (~1e-4 h at G = 1000) and the rounding of q - lo for a query far outside the box.
                const float qr = q[d] - g.lo[d];
                const float margin = 1e-3f * g.h[d] + 1e-6f * (qr < 0.f ? -qr : qr);
                if (cq[d] - s > 0) {
                    whole = false;
                    const float f = qr - (float)(cq[d] - s) * g.h[d] - margin;
                    bound = f < bound ? f : bound;
                }
                if (cq[d] + s < g.g[d] - 1) {
                    whole = false;
                    const float f = (float)(cq[d] + s + 1) * g.h[d] - qr - margin;
                    bound = f < bound ? f : bound;
                }
            }
            if (whole) break;
            if constexpr (RADIUS) { if (bound > 0.f && bound * bound >= r2) break; }
            else if (kth != NONE && bound > 0.f && knn_unorder((uint32_t)(kth >> 32)) < bound * bound) break;
        }
        if (lane < k) col[qy * k + lane] = mine == NONE ? (int64_t)-1 : (int64_t)(uint32_t)mine;
    }
}

// ---- radius: wave per query, candidates in index order, the first `max_nb` with squared distance < r^2 ----
template <typename T>
__global__ __launch_bounds__(256) void radius_kernel(const T* __restrict__ x, const T* __restrict__ y, const int64_t* __restrict__ ptr_x,
                                                     const int64_t* __restrict__ ptr_y, int B, int64_t Ny, int D, float r2, int max_nb,
                                                     int64_t* __restrict__ col) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t qy = wave0; qy < Ny; qy += nwaves) {
        const int b = segment_of(ptr_y, B, qy);
        const int64_t xb = ptr_x[b], xe = ptr_x[b + 1];
        const T* yq = y + qy * D;
        int count = 0;
        for (int64_t base = xb; base < xe && count < max_nb; base += 64) {
            const int64_t i = base + lane;
            const bool hit = i < xe && dist2<T>(x + i * D, yq, D) < r2;
            const uint64_t m = __ballot(hit);
            const int pos = count + __popcll(m & ((1ull << lane) - 1));
            if (hit && pos < max_nb) col[qy * max_nb + pos] = i;
            count += __popcll(m);
        }
        if (count > max_nb) count = max_nb;
        for (int p = count + lane; p < max_nb; p += 64) col[qy * max_nb + p] = -1;
    }
}

// Wave-wide maximum of a 64-bit key by DPP moves (row shifts inside the 16-lane rows, then the two row broadcasts): a handful of
// VALU instructions per step where __shfl_xor is a ds_bpermute round trip (~100 cycles, and a (distance, int64 index) pair is
// three of them per step) — the farthest-point loop below is ONE dependent reduction per sampled point, so this latency is
// its running time. 0 is the identity (lanes a move does not reach read 0). Every lane gets the result.
__device__ inline uint64_t wave_max_u64(uint64_t v) {
#define GNNOPS_DPP_MAX(CTRL, ROWMASK)                                                                                      \
    {                                                                                                                      \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROWMASK, 0xf, false);         \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROWMASK, 0xf, false); \
        const uint64_t o = ((uint64_t)hi << 32) | lo;                                                                      \
        v = o > v ? o : v;                                                                                                 \
    }
    GNNOPS_DPP_MAX(0x111, 0xf)   // row_shr:1
    GNNOPS_DPP_MAX(0x112, 0xf)   // row_shr:2
    GNNOPS_DPP_MAX(0x114, 0xf)   // row_shr:4
    GNNOPS_DPP_MAX(0x118, 0xf)   // row_shr:8   -> lane 15 of every row holds the row's maximum
    GNNOPS_DPP_MAX(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
    GNNOPS_DPP_MAX(0x143, 0xc)   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef GNNOPS_DPP_MAX
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}
// (distance, index) -> a key whose maximum is the farthest point, the smaller index among equals (rel = index - segment start);
// 0 = "no point" loses to every key
__device__ inline uint64_t fps_key(float d, uint32_t rel) { return ((uint64_t)knn_order(d) << 32) | (0xffffffffu - rel); }

// ---- fps: one workgroup per batch segment; dist[] = squared distance to the nearest chosen point so far ----
template <typename T, int DD>
__device__ inline void fps_in_registers(const T* __restrict__ x, int64_t xb, int64_t xe, int64_t ob, int64_t oe, int64_t cur,
                                        int64_t* __restrict__ out) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    {
        // A cloud of at most 8192 points in <= 3 dimensions (the sampling layers of point networks): every thread keeps its (up to
        // eight) points and their running distances in REGISTERS, and the chosen point's coordinates travel through LDS with the
        // reduction — nothing is read from or written to memory inside the sampling loop but the output index. The general
        // loop below re-reads dist[] and x[] from L2 and fetches x[cur] anew in every one of the (dependent) iterations.
        // Only as many waves as give a thread four to eight points (1024 points: four waves, one per SIMD) — the others leave at
        // once: the loop is one dependent instruction stream per wave, so fewer, shorter streams and fewer waves at the barriers.
        __shared__ float s_x[16][3];
        constexpr int PPT = 8;
        const int per = xe - xb <= 2048 ? 4 : 8;                 // points per thread: four for small clouds, eight above (measured)
        int nthr = (int)(((xe - xb + per - 1) / per + 63) / 64 * 64);   // 64 .. 1024 threads
        if (nthr > 1024) nthr = 1024;
        const int nw = nthr >> 6;
        if (t >= nthr) return;
        float px[PPT][3], pd[PPT];
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            const int64_t i = xb + t + (int64_t)p * nthr;
#pragma unroll
            for (int d = 0; d < 3; ++d) px[p][d] = (i < xe && d < DD) ? Elem<T>::load(x + i * DD + d) : 0.f;
            pd[p] = __builtin_huge_valf();
        }
        float cx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) cx[d] = d < DD ? Elem<T>::load(x + cur * DD + d) : 0.f;
        __shared__ uint64_t s_k[16];
        for (int64_t m = ob + 1; m < oe; ++m) {
            uint64_t best = 0;
            float bx[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                const int64_t i = xb + t + (int64_t)p * nthr;
                if (i < xe) {
                    float sq = 0.f;
                    #pragma unroll
                    for (int d = 0; d < DD; ++d) {         // the sum of dist2<T>, term by term
                        const float df = px[p][d] - cx[d];
                        sq += df * df;
                    }
                    const float dd = fminf(pd[p], sq);
                    pd[p] = dd;
                    const uint64_t key = fps_key(dd, (uint32_t)(i - xb));
                    if (key > best) { best = key; bx[0] = px[p][0]; bx[1] = px[p][1]; bx[2] = px[p][2]; }
                }
            }
            const uint64_t wbest = wave_max_u64(best);
            if (best == wbest && best != 0) { s_x[wave][0] = bx[0]; s_x[wave][1] = bx[1]; s_x[wave][2] = bx[2]; }
            if (lane == 0) s_k[wave] = wbest;
            __syncthreads();
            int win = 0;
            uint64_t top = s_k[0];
            for (int w = 1; w < nw; ++w)
                if (s_k[w] > top) { top = s_k[w]; win = w; }
            cx[0] = s_x[win][0]; cx[1] = s_x[win][1]; cx[2] = s_x[win][2];
            __syncthreads();
            cur = xb + (int64_t)(0xffffffffu - (uint32_t)top);
            if (t == 0) out[m] = cur;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void fps_kernel(const T* __restrict__ x, const int64_t* __restrict__ ptr, const int64_t* __restrict__ out_ptr,
                                                   const int64_t* __restrict__ start, int D, float* __restrict__ dist,
                                                   int64_t* __restrict__ out, int in_registers) {
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t xb = ptr[b], xe = ptr[b + 1];
    const int64_t ob = out_ptr[b], oe = out_ptr[b + 1];
    if (ob == oe || xb == xe) return;
    int64_t cur = start[b];
    if (t == 0) out[ob] = cur;
    if (in_registers && D <= 3 && xe - xb <= 8 * 1024) {
        if (D == 1) fps_in_registers<T, 1>(x, xb, xe, ob, oe, cur, out);
        else if (D == 2) fps_in_registers<T, 2>(x, xb, xe, ob, oe, cur, out);
        else fps_in_registers<T, 3>(x, xb, xe, ob, oe, cur, out);
        return;
    }
    for (int64_t i = xb + t; i < xe; i += 1024) dist[i] = __builtin_huge_valf();
    __shared__ uint64_t s_k[16];
    for (int64_t m = ob + 1; m < oe; ++m) {
        uint64_t best = 0;   // fps_key: farthest point, the smaller index among equals (torch.argmax's first maximum)
        for (int64_t i = xb + t; i < xe; i += 1024) {
            const float d = fminf(dist[i], dist2<T>(x + i * D, x + cur * D, D));
            dist[i] = d;
            const uint64_t key = fps_key(d, (uint32_t)(i - xb));
            best = key > best ? key : best;
        }
        const uint64_t wbest = wave_max_u64(best);
        if (lane == 0) s_k[wave] = wbest;
        __syncthreads();
        uint64_t top = s_k[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) top = s_k[w] > top ? s_k[w] : top;
        __syncthreads();
        cur = xb + (int64_t)(0xffffffffu - (uint32_t)top);
        if (t == 0) out[m] = cur;
    }
}

// ---- random_walk: uniform next neighbour from a CSR adjacency; walkers without neighbours stay where they are ----
__device__ inline uint32_t mix32(uint64_t z) {   // splitmix64 finaliser: counter-based, one draw per (seed, walker, step)
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
__global__ void random_walk_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col, const int64_t* __restrict__ start,
                                   int64_t S, int L, uint64_t seed, int64_t* __restrict__ out) {
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < S; w += (int64_t)gridDim.x * blockDim.x) {
        int64_t cur = start[w];
        out[w * (L + 1)] = cur;
        for (int l = 0; l < L; ++l) {
            const int64_t beg = rowptr[cur], deg = rowptr[cur + 1] - beg;
            if (deg > 0) {
                const uint32_t r = mix32(seed ^ ((uint64_t)w * 0x100000001b3ull + (uint64_t)l));
                cur = col[beg + (int64_t)(((uint64_t)r * (uint64_t)deg) >> 32)];
            }
            out[w * (L + 1) + l + 1] = cur;
        }
    }
}

// ---- random_walk with node2vec's return / in-out bias (p, q): rejection sampling as the package does it. The first step is
// uniform; afterwards a candidate x drawn uniformly from the neighbours of the current node v is accepted with probability
// (1/p, 1, 1/q) / max(1/p, 1, 1/q) depending on whether it IS the previous node t, is a neighbour of t, or neither. The
// neighbour test is a binary search in x's adjacency, which must be sorted (the host side sorts it). One draw pair per
// (walker, step, attempt): counter-based, reproducible per seed. ATTEMPTS bounds the loop (every wave must finish): after
// that many rejections the last candidate is taken — at p, q within [1/64, 64] the chance of getting there is < 1e-6.
constexpr int N2V_ATTEMPTS = 1024;
__device__ inline bool has_neighbour(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col, int64_t v, int64_t w) {
    int64_t lo = rowptr[v], hi = rowptr[v + 1];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t c = col[mid];
        if (c == w) return true;
        if (c < w) lo = mid + 1; else hi = mid;
    }
    return false;
}
__global__ void random_walk_n2v_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col, const int64_t* __restrict__ start,
                                       int64_t S, int L, float prob_t, float prob_nb, float prob_far, uint64_t seed,
                                       int64_t* __restrict__ out) {
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < S; w += (int64_t)gridDim.x * blockDim.x) {
        int64_t cur = start[w], prev = -1;
        out[w * (L + 1)] = cur;
        for (int l = 0; l < L; ++l) {
            const int64_t beg = rowptr[cur], deg = rowptr[cur + 1] - beg;
            int64_t next = cur;
            if (deg > 0) {
                const uint64_t ctr = seed ^ ((uint64_t)w * 0x100000001b3ull + (uint64_t)l);
                if (l == 0 || deg == 1) {
                    next = col[beg + (int64_t)(((uint64_t)mix32(ctr) * (uint64_t)deg) >> 32)];
                } else {
                    for (int a = 0; a < N2V_ATTEMPTS; ++a) {
                        const uint64_t c2 = ctr + (uint64_t)(a + 1) * 0x9e3779b97f4a7c15ull;
                        next = col[beg + (int64_t)(((uint64_t)mix32(c2) * (uint64_t)deg) >> 32)];
                        const float r = (float)(mix32(c2 ^ 0xd6e8feb86659fd93ull) >> 8) * (1.f / 16777216.f);   // [0, 1)
                        const float accept = next == prev ? prob_t : (has_neighbour(rowptr, col, next, prev) ? prob_nb : prob_far);
                        if (r < accept) break;
                    }
                }
            }
            prev = cur;
            cur = next;
            out[w * (L + 1) + l + 1] = cur;
        }
    }
}

// ---- graclus_cluster: greedy pairing of every node with one unmatched neighbour (the heaviest edge when weights are given) ----
// The package walks the nodes in a random order, sequentially. The parallel form is handshake matching: every unmatched node
// proposes along its best still-available edge; an edge whose two ends propose to each other is matched. "Best" is a total
// order on edges — (weight, a seeded hash of the unordered pair) — identical from both ends, so the best available edge of
// the whole graph is always mutual: every round matches at least one pair and the result is a maximal matching, random
// through the seed exactly as the package's is through its permutation. cluster[n] = min(n, partner), or n when alone.
__device__ inline uint32_t pair_hash(int64_t a, int64_t b, uint64_t seed) {
    const uint64_t lo = (uint64_t)(a < b ? a : b), hi = (uint64_t)(a < b ? b : a);
    return mix32(seed ^ (lo * 0x9e3779b97f4a7c15ull + hi));
}

template <typename T>
__global__ void graclus_propose_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col, const T* __restrict__ weight,
                                       int64_t N, uint64_t seed, const int64_t* __restrict__ cluster, int64_t* __restrict__ proposal,
                                       int* __restrict__ active) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        int64_t best = -1;
        if (cluster[n] < 0) {
            float bw = -__builtin_huge_valf();
            uint32_t bh = 0;
            for (int64_t e = rowptr[n]; e < rowptr[n + 1]; ++e) {
                const int64_t v = col[e];
                if (v == n || cluster[v] >= 0) continue;
                const float w = weight ? Elem<T>::load(weight + e) : 0.f;
                const uint32_t h = pair_hash(n, v, seed);
                if (best < 0 || w > bw || (w == bw && (h > bh || (h == bh && v < best)))) { best = v; bw = w; bh = h; }
            }
            if (best >= 0) atomicAdd(active, 1);
        }
        proposal[n] = best;
    }
}
__global__ void graclus_match_kernel(int64_t N, const int64_t* __restrict__ proposal, int64_t* __restrict__ cluster) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = proposal[n];
        if (v >= 0 && proposal[v] == n) cluster[n] = n < v ? n : v;   // both ends write the same id to their own slot
    }
}
__global__ void graclus_finish_kernel(int64_t N, int64_t* __restrict__ cluster) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x)
        if (cluster[n] < 0) cluster[n] = n;
}

}  // namespace

#define GNNOPS_BY_DTYPE(dtype, CALL, what)                                   \
    switch (dtype) {                                                         \
        case GNNOPS_F32: { using T = float; CALL; } break;                   \
        case GNNOPS_F16: { using T = __half; CALL; } break;                  \
        case GNNOPS_BF16: { using T = __hip_bfloat16; CALL; } break;         \
        default: gnnops_set_error(what ": unknown dtype %d", dtype); return GNNOPS_EINVAL; \
    }

extern "C" int gnnops_grid_cluster(const void* pos, int64_t N, int D, const double* d_size, const double* d_start, const double* d_end,
                                   int64_t* cluster, int dtype, gnnops_stream_t s) {
    GNNOPS_REQUIRE(N >= 0 && D >= 1 && D <= 16, GNNOPS_EINVAL, "grid_cluster: bad shape");
    if (N == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(pos && d_size && d_start && d_end && cluster, GNNOPS_EINVAL, "grid_cluster: null pointer");
    const int grid = gnnops_grid_cap(gnnops_cdiv(N, 256));
    GNNOPS_BY_DTYPE(dtype, hipLaunchKernelGGL((grid_cluster_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)pos, N, D, d_size,
                                              d_start, d_end, cluster), "grid_cluster")
    return gnnops_check_launch("grid_cluster");
}

extern "C" int gnnops_knn(const void* x, const void* y, const int64_t* ptr_x, const int64_t* ptr_y, int64_t batches, int64_t Ny, int D,
                          int k, int cosine, int64_t* col, int dtype, gnnops_stream_t s) {
    GNNOPS_REQUIRE(Ny >= 0 && D >= 1 && k >= 1 && batches >= 1 && batches < (1 << 30), GNNOPS_EINVAL, "knn: bad shape");
    if (Ny == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(x && y && ptr_x && ptr_y && col, GNNOPS_EINVAL, "knn: null pointer");
    const int grid = gnnops_grid_cap(gnnops_cdiv(Ny, 4), 256 * 32);
    const char* kf = getenv("GNNOPS_KNN_ROUNDS");   // A/B (tools/time_knn.py): 1 = the k-round kernel for every k
    if (k <= 64 && !(kf && kf[0] == '1')) {
        if (cosine) {
            GNNOPS_BY_DTYPE(dtype, hipLaunchKernelGGL((knn_topk_kernel<T, true>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)y,
                                                      ptr_x, ptr_y, (int)batches, Ny, D, k, col), "knn")
        } else {
            GNNOPS_BY_DTYPE(dtype, hipLaunchKernelGGL((knn_topk_kernel<T, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)y,
                                                      ptr_x, ptr_y, (int)batches, Ny, D, k, col), "knn")
        }
        return gnnops_check_launch("knn");
    }
    if (cosine) {
        GNNOPS_BY_DTYPE(dtype, hipLaunchKernelGGL((knn_kernel<T, true>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)y, ptr_x,
                                                  ptr_y, (int)batches, Ny, D, k, col), "knn")
    } else {
        GNNOPS_BY_DTYPE(dtype, hipLaunchKernelGGL((knn_kernel<T, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)y, ptr_x,
                                                  ptr_y, (int)batches, Ny, D, k, col), "knn")
    }
    return gnnops_check_launch("knn");
}

// Step 1 of the grid form: bounding box of x (6 words at `box`, device) and the cell id of every point (int64, the plan
// builder's index type). G cells per axis, G^D < 2^31. Then gnnops_plan_build(cell, Nx, G^D, rowptr, perm) and step 2.
extern "C" int gnnops_knn_grid_cells(const void* x, int64_t Nx, int D, int G, void* box, int64_t* cell, gnnops_stream_t s) {
    GNNOPS_REQUIRE(Nx >= 0 && D >= 1 && D <= 3 && G >= 1, GNNOPS_EINVAL, "knn_grid_cells: bad shape");
    GNNOPS_REQUIRE(box && (Nx == 0 || (x && cell)), GNNOPS_EINVAL, "knn_grid_cells: null pointer");
    hipStream_t stream = (hipStream_t)s;
    // images: min side starts at all ones, max side at zero
    if (gnnops_memset_async(box, 0xff, 12, stream) != hipSuccess || gnnops_memset_async((char*)box + 12, 0, 12, stream) != hipSuccess)
        return gnnops_check_launch("knn_grid_cells init");
    if (Nx == 0) return GNNOPS_OK;
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(gnnops_grid_cap(gnnops_cdiv(Nx, 1024), 1024)), dim3(256), 0, stream, (const float*)x, Nx, D,
                       (unsigned*)box);
    hipLaunchKernelGGL(knn_cells_kernel, dim3(gnnops_grid_cap(gnnops_cdiv(Nx, 256))), dim3(256), 0, stream, (const float*)x, Nx, D, G,
                       (const unsigned*)box, cell);
    return gnnops_check_launch("knn_grid_cells");
}

// Step 2: for every y its k nearest x (fp32, Euclidean, one cloud, k <= 64), nearest first, -1 where there are fewer: [Ny, k].
extern "C" int gnnops_knn_grid_query(const void* x, const void* y, int64_t Ny, int D, int k, int G, const void* box, const int32_t* rowptr,
                                     const int32_t* perm, int64_t* col, gnnops_stream_t s) {
    GNNOPS_REQUIRE(Ny >= 0 && D >= 1 && D <= 3 && G >= 1 && k >= 1 && k <= 64, GNNOPS_EINVAL, "knn_grid_query: bad shape");
    if (Ny == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(x && y && box && rowptr && perm && col, GNNOPS_EINVAL, "knn_grid_query: null pointer");
    const int grid = gnnops_grid_cap(gnnops_cdiv(Ny, 4), 256 * 32);
    hipLaunchKernelGGL(knn_grid_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const float*)x, (const float*)y, Ny, D, k, G,
                       (const unsigned*)box, rowptr, perm, col, 0.f);
    return gnnops_check_launch("knn_grid_query");
}

// torch_cluster.radius through the same grid: col [Ny, max_num_neighbors] = the smallest indices with |x - y|^2 < r^2, ascending.
extern "C" int gnnops_radius_grid_query(const void* x, const void* y, int64_t Ny, int D, double r, int max_num_neighbors, int G, const void* box,
                                        const int32_t* rowptr, const int32_t* perm, int64_t* col, gnnops_stream_t s) {
    GNNOPS_REQUIRE(Ny >= 0 && D >= 1 && D <= 3 && G >= 1 && max_num_neighbors >= 1 && max_num_neighbors <= 64, GNNOPS_EINVAL,
                   "radius_grid_query: bad shape");
    if (Ny == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(x && y && box && rowptr && perm && col, GNNOPS_EINVAL, "radius_grid_query: null pointer");
    const int grid = gnnops_grid_cap(gnnops_cdiv(Ny, 4), 256 * 32);
    hipLaunchKernelGGL(knn_grid_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const float*)x, (const float*)y, Ny, D, max_num_neighbors, G,
                       (const unsigned*)box, rowptr, perm, col, (float)(r * r));
    return gnnops_check_launch("radius_grid_query");
}

extern "C" int gnnops_radius(const void* x, const void* y, const int64_t* ptr_x, const int64_t* ptr_y, int64_t batches, int64_t Ny, int D,
                             double r, int max_num_neighbors, int64_t* col, int dtype, gnnops_stream_t s) {
    GNNOPS_REQUIRE(Ny >= 0 && D >= 1 && max_num_neighbors >= 1 && batches >= 1 && batches < (1 << 30), GNNOPS_EINVAL, "radius: bad shape");
    if (Ny == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(x && y && ptr_x && ptr_y && col, GNNOPS_EINVAL, "radius: null pointer");
    const int grid = gnnops_grid_cap(gnnops_cdiv(Ny, 4), 256 * 32);
    const float r2 = (float)(r * r);
    GNNOPS_BY_DTYPE(dtype, hipLaunchKernelGGL((radius_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)y, ptr_x, ptr_y,
                                              (int)batches, Ny, D, r2, max_num_neighbors, col), "radius")
    return gnnops_check_launch("radius");
}

extern "C" int gnnops_fps(const void* x, const int64_t* ptr, const int64_t* out_ptr, const int64_t* start, int64_t batches, int D,
                          float* dist_workspace, int64_t* out, int dtype, gnnops_stream_t s) {
    GNNOPS_REQUIRE(batches >= 0 && batches < (1 << 30) && D >= 1, GNNOPS_EINVAL, "fps: bad shape");
    if (batches == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(x && ptr && out_ptr && start && dist_workspace && out, GNNOPS_EINVAL, "fps: null pointer");
    const char* fr = getenv("GNNOPS_FPS_REGISTERS");   // A/B and tests: 0 = the general loop for every cloud
    const int in_registers = !(fr && fr[0] == '0');
    GNNOPS_BY_DTYPE(dtype, hipLaunchKernelGGL((fps_kernel<T>), dim3((unsigned)batches), dim3(1024), 0, (hipStream_t)s, (const T*)x, ptr, out_ptr,
                                              start, D, dist_workspace, out, in_registers), "fps")
    return gnnops_check_launch("fps");
}

extern "C" int gnnops_random_walk(const int64_t* rowptr, const int64_t* col, const int64_t* start, int64_t walkers, int walk_length,
                                  uint64_t seed, int64_t* out, gnnops_stream_t s) {
    GNNOPS_REQUIRE(walkers >= 0 && walk_length >= 0, GNNOPS_EINVAL, "random_walk: bad shape");
    if (walkers == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && start && out && col, GNNOPS_EINVAL, "random_walk: null pointer");
    const int grid = gnnops_grid_cap(gnnops_cdiv(walkers, 256));
    hipLaunchKernelGGL(random_walk_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, rowptr, col, start, walkers, walk_length, seed, out);
    return gnnops_check_launch("random_walk");
}

extern "C" int gnnops_random_walk_node2vec(const int64_t* rowptr, const int64_t* col, const int64_t* start, int64_t walkers, int walk_length,
                                           double p, double q, uint64_t seed, int64_t* out, gnnops_stream_t s) {
    GNNOPS_REQUIRE(walkers >= 0 && walk_length >= 0 && p > 0 && q > 0, GNNOPS_EINVAL, "random_walk_node2vec: bad argument");
    if (walkers == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && start && out && col, GNNOPS_EINVAL, "random_walk_node2vec: null pointer");
    const double mx = fmax(fmax(1.0 / p, 1.0), 1.0 / q);
    const int grid = gnnops_grid_cap(gnnops_cdiv(walkers, 256));
    hipLaunchKernelGGL(random_walk_n2v_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, rowptr, col, start, walkers, walk_length,
                       (float)(1.0 / p / mx), (float)(1.0 / mx), (float)(1.0 / q / mx), seed, out);
    return gnnops_check_launch("random_walk_node2vec");
}

// `rounds` propose + match rounds over a CSR adjacency (int64 rowptr / col, weight in CSR order or NULL). cluster: int64 [N],
// -1 = unmatched, set by the caller before the first call; d_active (device int) = proposals made in the LAST round run here:
// 0 means no unmatched node has an unmatched neighbour any more. finish != 0 then gives the nodes left alone their own id.
extern "C" int gnnops_graclus_rounds(const int64_t* rowptr, const int64_t* col, const void* weight, int64_t N, uint64_t seed, int rounds,
                                     int64_t* cluster, int64_t* proposal, int* d_active, int finish, int dtype, gnnops_stream_t s) {
    GNNOPS_REQUIRE(N >= 0 && rounds >= 0, GNNOPS_EINVAL, "graclus: bad shape");
    if (N == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && cluster && proposal && d_active, GNNOPS_EINVAL, "graclus: null pointer");
    hipStream_t stream = (hipStream_t)s;
    const int grid = gnnops_grid_cap(gnnops_cdiv(N, 256));
    for (int r = 0; r < rounds; ++r) {
        if (gnnops_memset_async(d_active, 0, sizeof(int), stream) != hipSuccess) return gnnops_check_launch("graclus memset");
        GNNOPS_BY_DTYPE(dtype, hipLaunchKernelGGL((graclus_propose_kernel<T>), dim3(grid), dim3(256), 0, stream, rowptr, col, (const T*)weight, N,
                                                  seed, cluster, proposal, d_active), "graclus")
        hipLaunchKernelGGL(graclus_match_kernel, dim3(grid), dim3(256), 0, stream, N, proposal, cluster);
    }
    if (finish) hipLaunchKernelGGL(graclus_finish_kernel, dim3(grid), dim3(256), 0, stream, N, cluster);
    return gnnops_check_launch("graclus");
}
