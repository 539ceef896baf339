// sort_rows.hip — torch.sort along the LAST dimension when a row fits in LDS (E <= 22528 fp32 keys): one workgroup
// sorts one row entirely on chip — the (20000, 20000) and 800^3 shapes of op_bm_scripts/benchmark_native_sort.py:37-45.
//
// The generic path (sort.hip) sorts (segment << 32 | key) with six or more passes over HBM. Here a row is read once
// (4 B/element) and written once (4 B value + 8 B index); the four 8-bit LSD passes run between registers and LDS:
//   rank   each wave owns consecutive rows of 64 keys; eight ballots give every lane its equal-digit peers, the lowest
//          peer does one returning LDS add per distinct digit (same scheme as sort_engine_impl.h, stable)
//   place  per-digit offsets across waves and digits, then (key, 16-bit position) go to their sorted slot in LDS
//   reload every wave reads its slice of the sorted image back for the next pass
// Stable; -0.0 / NaN conventions as in sort.hip. fp32 keys; positions fit 16 bits (E < 65536).
#include "common.h"

namespace {

__device__ inline uint32_t f32_key(float x) {
    uint32_t u = __float_as_uint(x);
    if (u == 0x80000000u) u = 0u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key_f32(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(u);
}

constexpr int RADIX = 256;

// THREADS x ROUNDS keys per row at most; LDS: keys u32[CAP] + pos u16[CAP] + per-wave histograms.
// A "row" here is one SEGMENT of a matrix row: segs == 1 is the whole row of E_total keys; segs == 2 sorts the halves
// [0, h0) and [h0, E_total) separately (indices stay positions in the whole row) for merge_halves_kernel below.
// I: the index type written — int64 (what torch.sort returns) or int32 (positions fit: rows are at most 40000 long) for the halves
// the merge reads back and for callers that widen the indices in a later pass (the dim-0 route's transposes, sparse.py).
template <int THREADS, int ROUNDS, typename I>
__global__ __launch_bounds__(THREADS) void sort_rows_kernel(const float* __restrict__ in, float* __restrict__ values,
                                                            I* __restrict__ indices, int64_t rows, int E_total,
                                                            int descending, int segs, int h0) {
    constexpr int WAVES = THREADS / 64;
    constexpr int CAP = THREADS * ROUNDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char sr_raw[];
    uint32_t* s_keys = reinterpret_cast<uint32_t*>(sr_raw);
    uint16_t* s_pos = reinterpret_cast<uint16_t*>(sr_raw + (size_t)CAP * 4);
    uint32_t* s_whist = reinterpret_cast<uint32_t*>(sr_raw + (size_t)CAP * 6);  // [WAVES][RADIX]
    uint32_t* s_tmp = s_whist + WAVES * RADIX;                                     // [WAVES]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint64_t lanes_below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int wave_base = wave * ROUNDS * 64;
    uint32_t* whist = s_whist + wave * RADIX;

    for (int64_t vrow = blockIdx.x; vrow < rows; vrow += gridDim.x) {
        const int64_t row = vrow / segs;
        const int sg = (int)(vrow - row * segs);
        const int off = sg * h0;
        const int E = (sg == segs - 1) ? E_total - off : h0;
        const float* src = in + row * E_total + off;
        uint32_t key[ROUNDS];
        uint32_t px[ROUNDS];  // source position in the high half; the low half is scratch for the ranking
        // the row's elements: ALL loads first (clamped index, nothing guarded), the key transform afterwards — transformed
        // where it is loaded, every element waited for its own load: ROUNDS memory latencies one after the other per row
        float raw[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int i = wave_base + r * 64 + lane;
            raw[r] = src[i < E ? i : E - 1];
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int i = wave_base + r * 64 + lane;
            uint32_t k = (i < E) ? f32_key(raw[r]) : 0u;
            key[r] = descending ? ~k : k;
            px[r] = (uint32_t)i << 16;
        }
#pragma unroll 1
        for (int shift = 0; shift < 32; shift += 8) {
            __syncthreads();  // previous pass's reloads (and the previous row's stores) are done with LDS
            for (int i = tid; i < WAVES * RADIX; i += THREADS) s_whist[i] = 0;
            __syncthreads();
            // rank inside the wave
            uint32_t is_leader = 0;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int i = wave_base + r * 64 + lane;
                const bool valid = i < E;
                const uint32_t d = (key[r] >> shift) & 255u;
                const uint64_t m = match_digit8(d, __ballot(valid));
                const uint32_t below = __popcll(m & lanes_below);
                uint32_t lo;  // 16 bits: the LDS add's return (leader) or below | leader_lane << 8 (others)
                if (valid && below == 0) {
                    lo = atomicAdd(&whist[d], (uint32_t)__popcll(m));
                    is_leader |= 1u << r;
                } else {
                    lo = below | ((uint32_t)(__ffsll((unsigned long long)m) - 1) << 8);
                }
                px[r] = (px[r] & 0xffff0000u) | (lo & 0xffffu);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const bool lead = (is_leader >> r) & 1u;
                const uint32_t lo = px[r] & 0xffffu;
                const int from = lead ? lane : (int)((lo >> 8) & 63u);
                const uint32_t p = __shfl(lo, from);
                const uint32_t rank = lead ? p : p + (lo & 255u);  // rank among equal digits of this wave
                px[r] = (px[r] & 0xffff0000u) | rank;
            }
            __syncthreads();
            // digit offsets: exclusive over waves, then over digits
            {
                const int d = tid & (RADIX - 1);
                const bool act = tid < RADIX;
                uint32_t tot = 0;
                if (act) {
#pragma unroll
                    for (int w = 0; w < WAVES; ++w) {
                        const uint32_t c = s_whist[w * RADIX + d];
                        s_whist[w * RADIX + d] = tot;
                        tot += c;
                    }
                }
                const uint32_t start = block_excl_scan_u32<WAVES>(act ? tot : 0u, s_tmp, nullptr);
                if (act) {
#pragma unroll
                    for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + d] += start;
                }
            }
            __syncthreads();
            // place
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int i = wave_base + r * 64 + lane;
                if (i < E) {
                    const uint32_t d = (key[r] >> shift) & 255u;
                    const uint32_t p = whist[d] + (px[r] & 0xffffu);
                    s_keys[p] = key[r];
                    s_pos[p] = (uint16_t)(px[r] >> 16);
                }
            }
            __syncthreads();
            // reload this wave's slice of the sorted image
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int i = wave_base + r * 64 + lane;
                if (i < E) {
                    key[r] = s_keys[i];
                    px[r] = (uint32_t)s_pos[i] << 16;
                }
            }
        }
        float* vdst = values + row * E_total + off;
        I* idst = indices + row * E_total + off;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int i = wave_base + r * 64 + lane;
            if (i < E) {
                const uint32_t k = descending ? ~key[r] : key[r];
                const uint32_t e = px[r] >> 16;
                // a zero or a NaN: the key does not fix the bits (-0.0 keyed as +0.0, one key for all NaNs) — re-read the
                // element (the row was just streamed: an L2 hit), so values == input.gather(indices) bit for bit
                vdst[i] = (k == 0x80000000u || k == 0xffffffffu) ? src[e] : key_f32(k);
                idst[i] = (I)(e + (uint32_t)off);
            }
        }
    }
}

template <int THREADS, int ROUNDS, typename I>
int launch(const float* in, float* values, I* indices, int64_t rows, int E, int descending, hipStream_t stream,
           int segs = 1, int h0 = 0) {
    constexpr int CAP = THREADS * ROUNDS;
    constexpr size_t LDS = (size_t)CAP * 6 + (size_t)(THREADS / 64) * RADIX * 4 + 64 * 4;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_rows_kernel<THREADS, ROUNDS, I>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) != hipSuccess)
            return gnnops_check_launch("sort_rows attribute");
        configured = true;
    }
    const int grid = gnnops_grid_cap(rows * segs, 256 * 8);
    hipLaunchKernelGGL((sort_rows_kernel<THREADS, ROUNDS, I>), dim3(grid), dim3(THREADS), LDS, stream, in, values, indices,
                       rows * segs, E, descending, segs, segs == 1 ? E : h0);
    return gnnops_check_launch("sort_rows");
}

// Rows of up to twice the on-chip capacity: the two halves come sorted (sort_rows_kernel with segs == 2); a workgroup
// parks the order images of a whole row in LDS and every element finds its place by ONE binary search in the other half
// (rank merge): position = own rank + number of smaller elements of the other half — strictly smaller for the first
// half, smaller-or-equal for the second, which keeps equal keys in source order (stable). Two HBM round trips per element
// in all, against six-plus radix passes over 64-bit (segment, key) pairs.
constexpr int MERGE_THREADS = 1024;

template <typename I>
__global__ __launch_bounds__(MERGE_THREADS) void merge_halves_kernel(const float* __restrict__ tmp_values,
                                                                     const int32_t* __restrict__ tmp_indices,
                                                                     float* __restrict__ values, I* __restrict__ indices,
                                                                     int64_t rows, int E, int h0, int descending) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mg_raw[];
    uint32_t* k = reinterpret_cast<uint32_t*>(mg_raw);  // [E] order images (complemented when descending: ascending here)
    const int h1 = E - h0;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* tv = tmp_values + row * E;
        const int32_t* ti = tmp_indices + row * E;
        __syncthreads();
        for (int i = threadIdx.x; i < E; i += MERGE_THREADS) {
            const uint32_t key = f32_key(tv[i]);
            k[i] = descending ? ~key : key;
        }
        __syncthreads();
        for (int i0 = threadIdx.x; i0 < E; i0 += MERGE_THREADS * 4) {
            int32_t src_idx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // unconditional, clamped: the four index loads are in flight together
                const int i = i0 + u * MERGE_THREADS;
                src_idx[u] = ti[i < E ? i : E - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * MERGE_THREADS;
                if (i >= E) continue;
                const uint32_t key = k[i];
                int lo, hi, pos;
                if (i < h0) {  // count elements of the second half that are strictly smaller
                    lo = 0; hi = h1;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (k[h0 + mid] < key) lo = mid + 1; else hi = mid; }
                    pos = i + lo;
                } else {       // count elements of the first half that are smaller or equal
                    lo = 0; hi = h0;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (k[mid] <= key) lo = mid + 1; else hi = mid; }
                    pos = (i - h0) + lo;
                }
                values[row * E + pos] = tv[i];
                indices[row * E + pos] = (I)src_idx[u];
            }
        }
    }
}

}  // namespace

// Largest row length the on-chip form takes.
extern "C" int64_t gnnops_sort_rows_max_len(void) { return 1024 * 22; }

template <typename I>
static int sort_rows_any(const float* input, float* values, I* indices, int64_t rows, int64_t E, int descending, hipStream_t stream) {
    GNNOPS_REQUIRE(rows >= 0 && E >= 0, GNNOPS_EINVAL, "sort_rows: negative size");
    GNNOPS_REQUIRE(E <= gnnops_sort_rows_max_len(), GNNOPS_EUNSUPPORTED, "sort_rows: row length %lld exceeds %lld",
                   (long long)E, (long long)gnnops_sort_rows_max_len());
    if (rows * E == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(input && values && indices, GNNOPS_EINVAL, "sort_rows: null pointer");
    const int e = (int)E;
    if (E <= 256 * 4) return launch<256, 4, I>(input, values, indices, rows, e, descending, stream);
    if (E <= 1024 * 4) return launch<1024, 4, I>(input, values, indices, rows, e, descending, stream);
    if (E <= 1024 * 8) return launch<1024, 8, I>(input, values, indices, rows, e, descending, stream);
    if (E <= 1024 * 16) return launch<1024, 16, I>(input, values, indices, rows, e, descending, stream);
    return launch<1024, 22, I>(input, values, indices, rows, e, descending, stream);
}

// input / values [rows, E] fp32, indices [rows, E] int64; sorted along E. E <= gnnops_sort_rows_max_len().
extern "C" int gnnops_sort_rows_f32(const float* input, float* values, int64_t* indices, int64_t rows, int64_t E,
                                    int descending, gnnops_stream_t s) {
    return sort_rows_any<int64_t>(input, values, indices, rows, E, descending, (hipStream_t)s);
}
// The same with the positions as int32 rows, for a caller that widens them itself in a later pass.
extern "C" int gnnops_sort_rows_f32_i32(const float* input, float* values, int32_t* indices, int64_t rows, int64_t E,
                                        int descending, gnnops_stream_t s) {
    return sort_rows_any<int32_t>(input, values, indices, rows, E, descending, (hipStream_t)s);
}

// Largest row length of the two-half form: the merge keeps one 4-byte order image per key of a row in LDS.
extern "C" int64_t gnnops_sort_rows2_max_len(void) { return 40000; }

template <typename I>
static int sort_rows2_any(const float* input, float* values, I* indices, float* tmp_values, int32_t* tmp_indices, int64_t rows, int64_t E,
                          int descending, hipStream_t stream) {
    GNNOPS_REQUIRE(rows >= 0 && E >= 0, GNNOPS_EINVAL, "sort_rows2: negative size");
    GNNOPS_REQUIRE(E > gnnops_sort_rows_max_len() && E <= gnnops_sort_rows2_max_len(), GNNOPS_EUNSUPPORTED,
                   "sort_rows2: row length %lld outside (%lld, %lld]", (long long)E, (long long)gnnops_sort_rows_max_len(),
                   (long long)gnnops_sort_rows2_max_len());
    if (rows == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(input && values && indices && tmp_values && tmp_indices, GNNOPS_EINVAL, "sort_rows2: null pointer");
    const int e = (int)E, h0 = (e + 1) / 2;
    int rc;
    if (h0 <= 1024 * 16) rc = launch<1024, 16, int32_t>(input, tmp_values, tmp_indices, rows, e, descending, stream, 2, h0);
    else rc = launch<1024, 22, int32_t>(input, tmp_values, tmp_indices, rows, e, descending, stream, 2, h0);
    if (rc != GNNOPS_OK) return rc;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&merge_halves_kernel<I>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return gnnops_check_launch("sort_rows2 attribute");
        configured = true;
    }
    hipLaunchKernelGGL(merge_halves_kernel<I>, dim3(gnnops_grid_cap(rows, 256 * 4)), dim3(MERGE_THREADS), (size_t)E * 4, stream,
                       tmp_values, tmp_indices, values, indices, rows, e, h0, descending);
    return gnnops_check_launch("sort_rows2");
}

// Rows longer than gnnops_sort_rows_max_len(), up to gnnops_sort_rows2_max_len(): the halves are sorted on chip into (tmp_values,
// tmp_indices) — each [rows, E], caller-provided — and merged into (values, indices). The half-sorted positions are kept as
// int32 inside tmp_indices whatever its declared width (the first 4 * rows * E bytes are used).
extern "C" int gnnops_sort_rows2_f32(const float* input, float* values, int64_t* indices, float* tmp_values,
                                     int64_t* tmp_indices, int64_t rows, int64_t E, int descending, gnnops_stream_t s) {
    return sort_rows2_any<int64_t>(input, values, indices, tmp_values, reinterpret_cast<int32_t*>(tmp_indices), rows, E, descending, (hipStream_t)s);
}
extern "C" int gnnops_sort_rows2_f32_i32(const float* input, float* values, int32_t* indices, float* tmp_values,
                                         int32_t* tmp_indices, int64_t rows, int64_t E, int descending, gnnops_stream_t s) {
    return sort_rows2_any<int32_t>(input, values, indices, tmp_values, tmp_indices, rows, E, descending, (hipStream_t)s);
}
