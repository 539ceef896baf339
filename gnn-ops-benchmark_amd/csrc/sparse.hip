// sparse.hip — COO housekeeping: torch_sparse.coalesce / Tensor.coalesce()
// (reference: op_bm_scripts/benchmark_sparse_coalesce.py:35-42), torch_sparse.transpose (= swap + coalesce;
// data/sparse_transpose.csv) and the dense `transpose(0,1).contiguous()` the current script times
// (benchmark_sparse_transpose.py:13-16).
//
// coalesce: sort entries by key = row << bits(n) | col with the radix engine (stable, 64-bit keys, only the bytes
// the key range needs), mark the first entry of every distinct key, compact (row, col) and reduce the
// values of each run in sorted order (fp32 accumulation, one rounding). The number of distinct keys is
// data dependent: outputs are sized nnz and the count is left in device memory for the caller.
// All phases are HBM-bound index work (8-16 B per entry per pass).
#include "common.h"
#include <stdlib.h>
#include "sort_engine.h"

namespace {

#define GRID_STRIDE(i, total) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_IPT = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_IPT;

// key = row << cbits | col (cbits = bits of n-1): the order of row * n + col, taken apart again by shift and mask
// K = uint32_t when row and column bits fit 32 (the reference's shapes): the sort then moves 8 instead of 12 B per entry
template <typename K>
__global__ void build_coo_keys_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                      K* __restrict__ keys, int64_t nnz, int cbits) {
    GRID_STRIDE(i, nnz) keys[i] = (K)(((uint64_t)row[i] << cbits) | (uint64_t)col[i]);
}

template <typename K>
__device__ inline bool is_head(const K* keys, int64_t p) { return p == 0 || keys[p] != keys[p - 1]; }

template <typename K>
__global__ __launch_bounds__(SCAN_THREADS) void count_heads_kernel(const K* __restrict__ keys, int64_t nnz,
                                                                   uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_tmp[SCAN_THREADS / 64];
    const int64_t p0 = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_IPT;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j)
        if (p0 + j < nnz && is_head(keys, p0 + j)) ++c;
    uint32_t tot;
    block_excl_scan_u32<SCAN_THREADS / 64>(c, s_tmp, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block_sums in place, total -> *d_count (int64)
__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(uint32_t* __restrict__ block_sums, int nb,
                                                                 int64_t* __restrict__ d_count) {
    __shared__ uint32_t s_tmp[SCAN_THREADS / 64];
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += SCAN_THREADS) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nb ? block_sums[i] : 0u;
        uint32_t tot;
        const uint32_t off = carry + block_excl_scan_u32<SCAN_THREADS / 64>(v, s_tmp, &tot);
        if (i < nb) block_sums[i] = off;
        carry += tot;
    }
    if (threadIdx.x == 0) *d_count = (int64_t)carry;
}

// compact: for every head at sorted position p with rank u: out_row[u], out_col[u], seg_start[u] = p.
// A block walks its 2048 positions in 8 rounds of 256 consecutive ones (coalesced key loads); a head's rank is the block's
// offset + heads of earlier rounds + heads of earlier waves (LDS) + heads of lower lanes (ballot), so the lanes of a wave
// store to consecutive slots. (One thread per 8 consecutive positions, each storing its heads one by one, took 266-294 us
// at the reference's spspmm shape — 8.8M positions, nearly all heads — against 60 us for this form.)
template <typename K>
__global__ __launch_bounds__(SCAN_THREADS) void emit_heads_kernel(const K* __restrict__ keys, int64_t nnz,
                                                                  int cbits, const uint32_t* __restrict__ block_off,
                                                                  int64_t* __restrict__ out_row,
                                                                  int64_t* __restrict__ out_col,
                                                                  uint32_t* __restrict__ seg_start) {
    __shared__ uint32_t s_wave[SCAN_IPT][SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    const uint64_t cmask = (1ull << cbits) - 1;
    K k[SCAN_IPT];
    uint64_t below[SCAN_IPT];  // ballot of heads among this wave's lanes, per round
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        const int64_t p = base + j * SCAN_THREADS + threadIdx.x;
        bool h = false;
        k[j] = 0;
        if (p < nnz) {
            k[j] = keys[p];
            h = p == 0 || k[j] != keys[p - 1];
        }
        below[j] = __ballot(h);
        if (lane == 0) s_wave[j][wave] = (uint32_t)__popcll(below[j]);
    }
    __syncthreads();
    uint32_t run = block_off[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; ++w) {
            const uint32_t c = s_wave[j][w];
            before += w < wave ? c : 0u;
            total += c;
        }
        if ((below[j] >> lane) & 1ull) {
            const uint32_t u = run + before + (uint32_t)__popcll(below[j] & ((1ull << lane) - 1));
            out_row[u] = (int64_t)((uint64_t)k[j] >> cbits);
            out_col[u] = (int64_t)((uint64_t)k[j] & cmask);
            seg_start[u] = (uint32_t)(base + j * SCAN_THREADS + threadIdx.x);
        }
        run += total;
    }
}

// out_value[u, c] = sum over sorted positions of run u, in order, of value[perm[p], c]
template <typename T>
__global__ void reduce_runs_kernel(const T* __restrict__ value, const uint32_t* __restrict__ perm,
                                   const uint32_t* __restrict__ seg_start, const int64_t* __restrict__ d_count,
                                   T* __restrict__ out_value, int64_t nnz, int64_t C) {
    const int64_t count = *d_count;
    GRID_STRIDE(o, count * C) {
        const int64_t u = o / C, c = o % C;
        const int64_t beg = seg_start[u];
        const int64_t end = (u + 1 < count) ? (int64_t)seg_start[u + 1] : nnz;
        float acc = 0.f;
        for (int64_t p = beg; p < end; ++p)
            acc = __fadd_rn(acc, Elem<T>::load(value + (perm ? (int64_t)perm[p] : p) * C + c));
        Elem<T>::store(out_value + o, acc);
    }
}

// ---- dense 2-D transpose through a padded LDS tile (64 x 64 elements, conflict-free column reads) ----
// O != U: the elements are converted on the way (int64 -> int32 index narrowing / int32 -> int64 arg widening of the
// dim-0 route of large full-index scatters, ops.py: the tile holds the OUTPUT type).
// TRACK (int64 in only): the largest element read goes to *max_out (atomicMax, one per wave) — torch_scatter's implicit
// dim_size = index.max() + 1 comes out of the index's own transpose instead of one more pass over it (ops.py).
template <typename U, typename O = U, bool TRACK = false>
__global__ __launch_bounds__(256) void transpose_kernel(const U* __restrict__ in, O* __restrict__ out, int64_t R,
                                                        int64_t C, long long* __restrict__ max_out = nullptr) {
    __shared__ O tile[64][64 + (sizeof(O) >= 4 ? 1 : 4 / sizeof(O))];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    in += (int64_t)blockIdx.z * R * C;   // batch of independent [R, C] matrices
    out += (int64_t)blockIdx.z * R * C;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    long long seen = -1;
    if (r0 + 64 <= R && c0 + 64 <= C) {   // interior tile: the sixteen loads are issued together, then parked in LDS
        U v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = in[(r0 + ty + 4 * j) * C + c0 + tx];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            tile[ty + 4 * j][tx] = (O)v[j];
            if constexpr (TRACK) seen = (long long)v[j] > seen ? (long long)v[j] : seen;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t r = r0 + ty + 4 * j, c = c0 + tx;
            if (r < R && c < C) {
                const U x = in[r * C + c];
                tile[ty + 4 * j][tx] = (O)x;
                if constexpr (TRACK) seen = (long long)x > seen ? (long long)x : seen;
            }
        }
    }
    if constexpr (TRACK) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const long long o = __shfl_xor(seen, d, 64);
            seen = o > seen ? o : seen;
        }
        // a look first: 1.4 M waves adding to ONE address serialise at the memory side (+10 ms at (38000)^2); once the running
        // maximum is near the top almost no wave has anything to add (a stale look only costs an atomic that changes nothing)
        if (tx == 0 && seen > __hip_atomic_load(max_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_out, seen);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t c = c0 + ty + 4 * j, r = r0 + tx;
        if (r < R && c < C) out[c * R + r] = tile[tx][ty + 4 * j];
    }
}

// 2-byte elements (the reference's dense fp16 transpose, benchmark_sparse_transpose.py:13-16): with one element per lane a
// wave moves 128 B per instruction. Here a lane moves PAIRS both ways — 128 x 128 tiles, 256-B row pieces in and out, the
// halves of two input rows recombined on the way out. Rows of odd length start 2 bytes off a dword: the 4-B accesses are
// declared 2-byte aligned (one dword access each in hardware).
typedef uint32_t u32_a2 __attribute__((aligned(2)));
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int64_t R,
                                                          int64_t C) {
    __shared__ uint32_t tile[128][65];   // [input row][input column pair]
    const int64_t r0 = (int64_t)blockIdx.y * 128, c0 = (int64_t)blockIdx.x * 128;
    in += (int64_t)blockIdx.z * R * C;
    out += (int64_t)blockIdx.z * R * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool interior = r0 + 128 <= R && c0 + 128 <= C;
    if (interior) {
        uint32_t v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = *reinterpret_cast<const u32_a2*>(in + (r0 + wave + 4 * j) * C + c0 + 2 * lane);
#pragma unroll
        for (int j = 0; j < 32; ++j) tile[wave + 4 * j][lane] = v[j];
    } else {
#pragma unroll 4
        for (int j = 0; j < 32; ++j) {
            const int64_t r = r0 + wave + 4 * j, c = c0 + 2 * lane;
            uint32_t lo = 0, hi = 0;
            if (r < R && c < C) lo = in[r * C + c];
            if (r < R && c + 1 < C) hi = in[r * C + c + 1];
            tile[wave + 4 * j][lane] = lo | (hi << 16);
        }
    }
    __syncthreads();
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
        const int c = wave + 4 * j;                      // output row = input column c0 + c; this lane: input rows 2 lane, 2 lane + 1
        const uint32_t a = tile[2 * lane][c >> 1], b = tile[2 * lane + 1][c >> 1];
        const uint32_t w = (c & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
        const int64_t oc = c0 + c, orow = r0 + 2 * lane;
        if (interior) {
            *reinterpret_cast<u32_a2*>(out + oc * R + orow) = w;
        } else if (oc < C) {
            if (orow < R) out[oc * R + orow] = (uint16_t)w;
            if (orow + 1 < R) out[oc * R + orow + 1] = (uint16_t)(w >> 16);
        }
    }
}

// 4-byte elements, wider pieces: a lane loads LW and stores SW consecutive elements (8 B when 2): tiles of 64 SW rows x 64 LW
// columns, row pieces of 256 LW bytes in and 256 SW bytes out. Rows need only 4-byte alignment (odd lengths).
struct __attribute__((aligned(4))) u32x2_a4 { uint32_t x, y; };
template <int LW, int SW>
__global__ __launch_bounds__(256) void transpose32_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t R,
                                                          int64_t C) {
    constexpr int TR = 64 * SW, TC = 64 * LW;
    __shared__ uint32_t tile[TR][TC + 1];
    const int64_t r0 = (int64_t)blockIdx.y * TR, c0 = (int64_t)blockIdx.x * TC;
    in += (int64_t)blockIdx.z * R * C;
    out += (int64_t)blockIdx.z * R * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool interior = r0 + TR <= R && c0 + TC <= C;
    if (interior) {
        uint32_t v[TR / 4][LW];
#pragma unroll
        for (int j = 0; j < TR / 4; ++j) {
            const uint32_t* p = in + (r0 + wave + 4 * j) * C + c0 + LW * lane;
            if constexpr (LW == 2) {
                const u32x2_a4 t = *reinterpret_cast<const u32x2_a4*>(p);
                v[j][0] = t.x; v[j][1] = t.y;
            } else {
                v[j][0] = *p;
            }
        }
#pragma unroll
        for (int j = 0; j < TR / 4; ++j)
#pragma unroll
            for (int w = 0; w < LW; ++w) tile[wave + 4 * j][LW * lane + w] = v[j][w];
    } else {
        for (int j = 0; j < TR / 4; ++j)
            for (int w = 0; w < LW; ++w) {
                const int64_t r = r0 + wave + 4 * j, c = c0 + LW * lane + w;
                if (r < R && c < C) tile[wave + 4 * j][LW * lane + w] = in[r * C + c];
            }
    }
    __syncthreads();
#pragma unroll 8
    for (int j = 0; j < TC / 4; ++j) {
        const int c = wave + 4 * j;
        const int64_t oc = c0 + c, orow = r0 + SW * lane;
        if (interior) {
            if constexpr (SW == 2) {
                u32x2_a4 t;
                t.x = tile[2 * lane][c]; t.y = tile[2 * lane + 1][c];
                *reinterpret_cast<u32x2_a4*>(out + oc * R + orow) = t;
            } else {
                out[oc * R + orow] = tile[lane][c];
            }
        } else if (oc < C) {
            for (int w = 0; w < SW; ++w)
                if (orow + w < R) out[oc * R + orow + w] = tile[SW * lane + w][c];
        }
    }
}

// int64 -> int32 with 8-B pieces BOTH ways: a lane loads one int64 and stores two int32 of consecutive input rows (128-row
// tiles, 512-B row pieces in and out) — the index narrowing of the dim-0 route at its own size ((38000)^2: 17.4 GB).
template <bool TRACK>
__global__ __launch_bounds__(256) void transpose_narrow_kernel(const int64_t* __restrict__ in, int32_t* __restrict__ out, int64_t R,
                                                               int64_t C, long long* __restrict__ max_out) {
    __shared__ int32_t tile[128][65];
    const int64_t r0 = (int64_t)blockIdx.y * 128, c0 = (int64_t)blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool interior = r0 + 128 <= R && c0 + 64 <= C;
    long long seen = -1;
    if (interior) {
        int64_t v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = in[(r0 + wave + 4 * j) * C + c0 + lane];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            tile[wave + 4 * j][lane] = (int32_t)v[j];
            if constexpr (TRACK) seen = v[j] > seen ? v[j] : seen;
        }
    } else {
        for (int j = 0; j < 32; ++j) {
            const int64_t r = r0 + wave + 4 * j, c = c0 + lane;
            if (r < R && c < C) {
                const int64_t x = in[r * C + c];
                tile[wave + 4 * j][lane] = (int32_t)x;
                if constexpr (TRACK) seen = x > seen ? x : seen;
            }
        }
    }
    if constexpr (TRACK) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const long long o = __shfl_xor(seen, d, 64);
            seen = o > seen ? o : seen;
        }
        if (lane == 0 && seen > __hip_atomic_load(max_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_out, seen);
    }
    __syncthreads();
#pragma unroll 8
    for (int j = 0; j < 16; ++j) {
        const int c = wave + 4 * j;
        const int64_t oc = c0 + c, orow = r0 + 2 * lane;
        if (interior) {
            u32x2_a4 t;
            t.x = (uint32_t)tile[2 * lane][c]; t.y = (uint32_t)tile[2 * lane + 1][c];
            *reinterpret_cast<u32x2_a4*>(out + oc * R + orow) = t;
        } else if (oc < C) {
            if (orow < R) out[oc * R + orow] = tile[2 * lane][c];
            if (orow + 1 < R) out[oc * R + orow + 1] = tile[2 * lane + 1][c];
        }
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int grid_for(int64_t n) { return gnnops_grid_cap(gnnops_cdiv(n, 256), 256 * 16); }

inline int bits_of(int64_t count) {  // bits that hold 0 .. count-1 (at least 1)
    int b = 1;
    while (b < 63 && ((int64_t)1 << b) < count) ++b;
    return b;
}

inline int sort_pass(const uint32_t* ki, const uint32_t* vi, uint32_t* ko, uint32_t* vo, int64_t n, int shift, uint32_t* th,
                     uint32_t* dt, int tiles, hipStream_t st) {
    return vi ? sortengine::pass_u32(ki, vi, ko, vo, n, shift, th, dt, tiles, st)
              : sortengine::pass_first_u32(ki, ko, vo, n, shift, th, dt, tiles, st);
}
inline int sort_pass(const uint64_t* ki, const uint32_t* vi, uint64_t* ko, uint32_t* vo, int64_t n, int shift, uint32_t* th,
                     uint32_t* dt, int tiles, hipStream_t st) {
    return vi ? sortengine::pass_u64(ki, vi, ko, vo, n, shift, th, dt, tiles, st)
              : sortengine::pass_first_u64(ki, ko, vo, n, shift, th, dt, tiles, st);
}

// keys, stable LSD sort over the key bits in use, head count and compaction; *sorted_vals = the payload in sorted order
// (the permutation, or `carried` — the caller's scalar values — when given)
template <typename K>
int sort_and_compact(const int64_t* row, const int64_t* col, const uint32_t* carried, int64_t nnz, int cbits, int rbits,
                     K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, uint32_t* tile_hist, uint32_t* digit_total,
                     int tiles, uint32_t* block_sums, int nb, int64_t* d_count, int64_t* out_row, int64_t* out_col,
                     uint32_t* seg_start, const uint32_t** sorted_vals, hipStream_t stream) {
    hipLaunchKernelGGL(build_coo_keys_kernel<K>, dim3(grid_for(nnz)), dim3(256), 0, stream, row, col, keys_b, nnz, cbits);
    const int passes = (cbits + rbits + 7) / 8;  // 8-bit passes over the key bits in use
    K* kin = keys_b;
    K* kout = keys_a;
    const uint32_t* vin = carried;
    uint32_t* vout = vals_a;
    for (int p = 0; p < passes; ++p) {
        const int rc = sort_pass(kin, vin, kout, vout, nnz, 8 * p, tile_hist, digit_total, tiles, stream);
        if (rc) return rc;
        K* tk = kin; kin = kout; kout = tk;
        vin = vout;
        vout = (vout == vals_a) ? vals_b : vals_a;
    }
    hipLaunchKernelGGL(count_heads_kernel<K>, dim3(nb), dim3(SCAN_THREADS), 0, stream, (const K*)kin, nnz, block_sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, block_sums, nb, d_count);
    hipLaunchKernelGGL(emit_heads_kernel<K>, dim3(nb), dim3(SCAN_THREADS), 0, stream, (const K*)kin, nnz, cbits, block_sums,
                       out_row, out_col, seg_start);
    *sorted_vals = vin;
    return GNNOPS_OK;
}

}  // namespace

// workspace: keys_a[nnz] u64 | keys_b[nnz] u64 | vals_a[nnz] u32 | vals_b[nnz] u32 | tile_hist | digit_total |
//            block_sums[ceil(nnz/2048)] u32 | seg_start[nnz] u32
extern "C" size_t gnnops_coalesce_workspace_bytes(int64_t nnz) {
    if (nnz < 0) return 0;
    const size_t n = (size_t)nnz;
    const size_t tiles = (size_t)gnnops_cdiv(nnz > 0 ? nnz : 1, sortengine::TILE);
    const size_t nb = (size_t)gnnops_cdiv(nnz > 0 ? nnz : 1, SCAN_TILE);
    return 2 * align_up(n * 8, 256) + 2 * align_up(n * 4, 256) + align_up(256 * tiles * 4, 256) + 1024 +
           align_up(nb * 4, 256) + align_up(n * 4, 256);
}

extern "C" int gnnops_coalesce(const int64_t* row, const int64_t* col, const void* value, int64_t nnz, int64_t m,
                               int64_t n, int64_t C, int dtype, int64_t* out_row, int64_t* out_col, void* out_value,
                               int64_t* d_count, void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(nnz >= 0 && m >= 0 && n >= 0 && C >= 0, GNNOPS_EINVAL, "coalesce: negative size");
    GNNOPS_REQUIRE(nnz < ((int64_t)1 << 32), GNNOPS_EUNSUPPORTED, "coalesce: nnz must be < 2^32");
    GNNOPS_REQUIRE(d_count != nullptr, GNNOPS_EINVAL, "coalesce: d_count is null");
    if (nnz == 0) {
        if (gnnops_memset_async(d_count, 0, sizeof(int64_t), stream) != hipSuccess) return gnnops_check_launch("coalesce memset");
        return GNNOPS_OK;
    }
    GNNOPS_REQUIRE(row && col && out_row && out_col, GNNOPS_EINVAL, "coalesce: null pointer");
    GNNOPS_REQUIRE(value == nullptr || out_value != nullptr, GNNOPS_EINVAL, "coalesce: value given without out_value");
    GNNOPS_REQUIRE(workspace && workspace_bytes >= gnnops_coalesce_workspace_bytes(nnz), GNNOPS_EWORKSPACE,
                   "coalesce: workspace %zu < %zu", workspace_bytes, gnnops_coalesce_workspace_bytes(nnz));
    const int tiles = (int)gnnops_cdiv(nnz, sortengine::TILE);
    const int nb = (int)gnnops_cdiv(nnz, SCAN_TILE);
    char* w = (char*)workspace;
    uint64_t* keys_a = (uint64_t*)w; w += align_up((size_t)nnz * 8, 256);
    uint64_t* keys_b = (uint64_t*)w; w += align_up((size_t)nnz * 8, 256);
    uint32_t* vals_a = (uint32_t*)w; w += align_up((size_t)nnz * 4, 256);
    uint32_t* vals_b = (uint32_t*)w; w += align_up((size_t)nnz * 4, 256);
    uint32_t* tile_hist = (uint32_t*)w; w += align_up((size_t)256 * tiles * 4, 256);
    uint32_t* digit_total = (uint32_t*)w; w += 1024;
    uint32_t* block_sums = (uint32_t*)w; w += align_up((size_t)nb * 4, 256);
    uint32_t* seg_start = (uint32_t*)w;

    const int cbits = bits_of(n), rbits = bits_of(m);
    GNNOPS_REQUIRE(cbits + rbits <= 64, GNNOPS_EUNSUPPORTED, "coalesce: m x n does not fit a 64-bit key");
    // scalar fp32 values (the reference's case) ride through the sort as the 32-bit payload themselves, so the run
    // reduction streams them instead of gathering value[perm[p]]; the stable order of equal keys is the same either way
    const bool carry_values = value != nullptr && C == 1 && dtype == GNNOPS_F32;
    const uint32_t* vin = nullptr;  // after the sort: the permutation (or the carried values)
    const int rc = cbits + rbits <= 32
                       ? sort_and_compact<uint32_t>(row, col, carry_values ? (const uint32_t*)value : nullptr, nnz, cbits, rbits,
                                                    (uint32_t*)keys_a, (uint32_t*)keys_b, vals_a, vals_b, tile_hist, digit_total,
                                                    tiles, block_sums, nb, d_count, out_row, out_col, seg_start, &vin, stream)
                       : sort_and_compact<uint64_t>(row, col, carry_values ? (const uint32_t*)value : nullptr, nnz, cbits, rbits,
                                                    keys_a, keys_b, vals_a, vals_b, tile_hist, digit_total, tiles, block_sums, nb,
                                                    d_count, out_row, out_col, seg_start, &vin, stream);
    if (rc) return rc;
    if (value && C > 0) {
        const int grid = grid_for(nnz * C);
        switch (dtype) {
            case GNNOPS_F32:
                if (carry_values)
                    hipLaunchKernelGGL((reduce_runs_kernel<float>), dim3(grid), dim3(256), 0, stream, (const float*)vin,
                                       (const uint32_t*)nullptr, seg_start, d_count, (float*)out_value, nnz, C);
                else
                    hipLaunchKernelGGL((reduce_runs_kernel<float>), dim3(grid), dim3(256), 0, stream, (const float*)value,
                                       vin, seg_start, d_count, (float*)out_value, nnz, C);
                break;
            case GNNOPS_F16:
                hipLaunchKernelGGL((reduce_runs_kernel<__half>), dim3(grid), dim3(256), 0, stream, (const __half*)value, vin,
                                   seg_start, d_count, (__half*)out_value, nnz, C);
                break;
            case GNNOPS_BF16:
                hipLaunchKernelGGL((reduce_runs_kernel<__hip_bfloat16>), dim3(grid), dim3(256), 0, stream,
                                   (const __hip_bfloat16*)value, vin, seg_start, d_count, (__hip_bfloat16*)out_value, nnz, C);
                break;
            default:
                gnnops_set_error("coalesce: unknown dtype %d", dtype);
                return GNNOPS_EINVAL;
        }
    }
    return gnnops_check_launch("coalesce");
}

extern "C" int gnnops_transpose2d(const void* in, void* out, int64_t R, int64_t C, int elem_bytes, gnnops_stream_t s) {
    return gnnops_transpose_batched(in, out, 1, R, C, elem_bytes, s);
}

// [R, C] int64 -> [C, R] int32 (mode 0: every value must fit) or [R, C] int32 -> [C, R] int64 (mode 1, sign-extended).
extern "C" int gnnops_transpose2d_cvt(const void* in, void* out, int64_t R, int64_t C, int mode, gnnops_stream_t s) {
    GNNOPS_REQUIRE(R >= 0 && C >= 0 && (mode == 0 || mode == 1), GNNOPS_EINVAL, "transpose2d_cvt: bad argument");
    if (R * C == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(in && out, GNNOPS_EINVAL, "transpose2d_cvt: null pointer");
    GNNOPS_REQUIRE(gnnops_cdiv(R, 64) < 65536, GNNOPS_EUNSUPPORTED, "transpose2d_cvt: too many rows");
    dim3 grid((unsigned)gnnops_cdiv(C, 64), (unsigned)gnnops_cdiv(R, 64), 1);
    if (mode == 0 && R >= 128 && gnnops_cdiv(R, 128) < 65536)
        hipLaunchKernelGGL((transpose_narrow_kernel<false>), dim3((unsigned)gnnops_cdiv(C, 64), (unsigned)gnnops_cdiv(R, 128), 1), dim3(256), 0,
                           (hipStream_t)s, (const int64_t*)in, (int32_t*)out, R, C, (long long*)nullptr);
    else if (mode == 0)
        hipLaunchKernelGGL((transpose_kernel<int64_t, int32_t>), grid, dim3(256), 0, (hipStream_t)s, (const int64_t*)in, (int32_t*)out, R, C);
    else
        hipLaunchKernelGGL((transpose_kernel<int32_t, int64_t>), grid, dim3(256), 0, (hipStream_t)s, (const int32_t*)in, (int64_t*)out, R, C);
    return gnnops_check_launch("transpose2d_cvt");
}

extern "C" int gnnops_transpose2d_cvt_max(const void* in, void* out, int64_t R, int64_t C, int64_t* max_out, gnnops_stream_t s) {
    GNNOPS_REQUIRE(R >= 0 && C >= 0, GNNOPS_EINVAL, "transpose2d_cvt_max: negative size");
    GNNOPS_REQUIRE(max_out != nullptr, GNNOPS_EINVAL, "transpose2d_cvt_max: null pointer");
    if (gnnops_memset_async(max_out, 0xff, sizeof(int64_t), (hipStream_t)s) != hipSuccess) return gnnops_check_launch("transpose2d_cvt_max init");
    if (R * C == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(in && out, GNNOPS_EINVAL, "transpose2d_cvt_max: null pointer");
    GNNOPS_REQUIRE(gnnops_cdiv(R, 64) < 65536, GNNOPS_EUNSUPPORTED, "transpose2d_cvt_max: too many rows");
    dim3 grid((unsigned)gnnops_cdiv(C, 64), (unsigned)gnnops_cdiv(R, 64), 1);
    if (R >= 128)
        hipLaunchKernelGGL((transpose_narrow_kernel<true>), dim3((unsigned)gnnops_cdiv(C, 64), (unsigned)gnnops_cdiv(R, 128), 1), dim3(256), 0,
                           (hipStream_t)s, (const int64_t*)in, (int32_t*)out, R, C, (long long*)max_out);
    else
        hipLaunchKernelGGL((transpose_kernel<int64_t, int32_t, true>), grid, dim3(256), 0, (hipStream_t)s, (const int64_t*)in, (int32_t*)out, R, C,
                           (long long*)max_out);
    return gnnops_check_launch("transpose2d_cvt_max");
}

extern "C" int gnnops_transpose_batched(const void* in, void* out, int64_t batch, int64_t R, int64_t C, int elem_bytes,
                                        gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(R >= 0 && C >= 0 && batch >= 0, GNNOPS_EINVAL, "transpose2d: negative size");
    if (batch > 65535) {  // grid.z limit: split the batch
        const int64_t bytes = R * C * elem_bytes;
        for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
            const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
            const int rc = gnnops_transpose_batched((const char*)in + b0 * bytes, (char*)out + b0 * bytes, nb, R, C, elem_bytes, s);
            if (rc) return rc;
        }
        return GNNOPS_OK;
    }
    GNNOPS_REQUIRE(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8, GNNOPS_EUNSUPPORTED,
                   "transpose2d: elem_bytes %d", elem_bytes);
    if (R * C * batch == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(in && out, GNNOPS_EINVAL, "transpose2d: null pointer");
    GNNOPS_REQUIRE(gnnops_cdiv(R, 64) < 65536, GNNOPS_EUNSUPPORTED, "transpose2d: too many rows");
    dim3 grid((unsigned)gnnops_cdiv(C, 64), (unsigned)gnnops_cdiv(R, 64), (unsigned)batch);
    if (elem_bytes == 1)
        hipLaunchKernelGGL((transpose_kernel<uint8_t>), grid, dim3(256), 0, stream, (const uint8_t*)in, (uint8_t*)out, R, C);
    else if (elem_bytes == 2 && R >= 128 && C >= 128 && gnnops_cdiv(R, 128) < 65536)
        hipLaunchKernelGGL(transpose16_kernel, dim3((unsigned)gnnops_cdiv(C, 128), (unsigned)gnnops_cdiv(R, 128), (unsigned)batch), dim3(256), 0,
                           stream, (const uint16_t*)in, (uint16_t*)out, R, C);
    else if (elem_bytes == 2)
        hipLaunchKernelGGL((transpose_kernel<uint16_t>), grid, dim3(256), 0, stream, (const uint16_t*)in, (uint16_t*)out, R, C);
    else if (elem_bytes == 4 && R >= 128 && C >= 128 && gnnops_cdiv(R, 128) < 65536 && !(getenv("GNNOPS_T32") && getenv("GNNOPS_T32")[0] == '0')) {
        // A/B (tools/time_transpose.py): 0 = one element per lane, 1 = 8-B loads, 2 = 8-B stores, 3 = both. Measured: 8-B loads win
        // up to (8192)^2 (111 -> 90 us at (7071)^2), 8-B stores from there on ((28200)^2 1650 -> 1309 us, (38000)^2 2892 -> 2516)
        const char* m = getenv("GNNOPS_T32");
        const int mode = m ? atoi(m) : (R * C >= ((int64_t)1 << 27) ? 2 : 1);
        const uint32_t* i = (const uint32_t*)in;
        uint32_t* o = (uint32_t*)out;
        if (mode == 1)
            hipLaunchKernelGGL((transpose32_kernel<2, 1>), dim3((unsigned)gnnops_cdiv(C, 128), (unsigned)gnnops_cdiv(R, 64), (unsigned)batch), dim3(256), 0, stream, i, o, R, C);
        else if (mode == 2)
            hipLaunchKernelGGL((transpose32_kernel<1, 2>), dim3((unsigned)gnnops_cdiv(C, 64), (unsigned)gnnops_cdiv(R, 128), (unsigned)batch), dim3(256), 0, stream, i, o, R, C);
        else
            hipLaunchKernelGGL((transpose32_kernel<2, 2>), dim3((unsigned)gnnops_cdiv(C, 128), (unsigned)gnnops_cdiv(R, 128), (unsigned)batch), dim3(256), 0, stream, i, o, R, C);
    } else if (elem_bytes == 4)
        hipLaunchKernelGGL((transpose_kernel<uint32_t>), grid, dim3(256), 0, stream, (const uint32_t*)in, (uint32_t*)out, R, C);
    else
        hipLaunchKernelGGL((transpose_kernel<uint64_t>), grid, dim3(256), 0, stream, (const uint64_t*)in, (uint64_t*)out, R, C);
    return gnnops_check_launch("transpose2d");
}
