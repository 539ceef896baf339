// sort_engine_impl.h — stable LSD radix-sort pass for gfx950 (64-lane waves), key/value pairs.
//
// One pass = three launches:
//   hist_kernel     grid = tiles        per-tile digit histogram -> tile_hist[digit][tile]
//   scan_kernel     grid = 256 digits   exclusive scan of tile_hist[digit][*], digit totals
//   scatter_kernel  grid = tiles        stable rank of every key inside its tile (wave ballots +
//                                       per-wave LDS counters), reorder through LDS, then write each
//                                       digit's run to its global position (runs are contiguous, so
//                                       stores coalesce)
// Keys are processed as u32 digits of a u32 or u64 key; values are u32 (positions < 2^32).
// Tile = 4 waves x 32 rows x 64 lanes = 8192 keys: 64 KiB of LDS for the reorder buffers, two
// workgroups per CU.
#pragma once
#include "common.h"
#include "sort_engine.h"

namespace sortengine {


// --- key adapters: how a stored key yields the current 8-bit digit and how it is carried ---
struct KeyI64Low32 {  // first pass of the plan builder: int64 index -> u32 key
    using In = int64_t;
    using Carry = uint32_t;
    __device__ static inline Carry load(const In* p, int64_t i) { return (uint32_t)p[i]; }
    __device__ static inline uint32_t digit(Carry k, int shift) { return (k >> shift) & 255u; }
};
struct KeyU32 {
    using In = uint32_t;
    using Carry = uint32_t;
    __device__ static inline Carry load(const In* p, int64_t i) { return p[i]; }
    __device__ static inline uint32_t digit(Carry k, int shift) { return (k >> shift) & 255u; }
};
struct KeyU64 {
    using In = uint64_t;
    using Carry = uint64_t;
    __device__ static inline Carry load(const In* p, int64_t i) { return p[i]; }
    __device__ static inline uint32_t digit(Carry k, int shift) { return (uint32_t)(k >> shift) & 255u; }
};
// torch.sort on fp32: order-preserving map float -> u32 (negatives flipped entirely, positives get the
// sign bit); NaNs sort last like torch (they map above +inf because their payload is non-zero).
struct KeyF32 {
    using In = float;
    using Carry = uint32_t;
    __device__ static inline Carry load(const In* p, int64_t i) {
        uint32_t u = __float_as_uint(p[i]);
        if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // any NaN -> top key
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    __device__ static inline uint32_t digit(Carry k, int shift) { return (k >> shift) & 255u; }
};

__device__ inline uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// Exclusive scan of one value per thread over a 256-thread block. s_tmp: WAVES words of LDS.
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t* s_tmp, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan(v);
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        uint32_t t = s_tmp[w];
        if (w < wave) off += t;
        tot += t;
    }
    __syncthreads();
    if (total) *total = tot;
    return off + incl - v;
}

template <typename KA>
__global__ __launch_bounds__(THREADS) void hist_kernel(const typename KA::In* __restrict__ keys, int64_t n, int shift,
                                                       uint32_t* __restrict__ tile_hist, int num_tiles) {
    __shared__ uint32_t h[WAVES][RADIX];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < WAVES * RADIX; i += THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * TILE;
    const int64_t lim = (n - base < TILE) ? (n - base) : TILE;
#pragma unroll 8
    for (int i = tid; i < lim; i += THREADS) {
        uint32_t d = KA::digit(KA::load(keys, base + i), shift);
        atomicAdd(&h[wave][d], 1u);
    }
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t += h[w][tid];
    tile_hist[(size_t)tid * num_tiles + blockIdx.x] = t;
}

// One block per digit: in-place exclusive scan across tiles, total -> digit_total[digit].
__global__ __launch_bounds__(THREADS) void scan_kernel(uint32_t* __restrict__ tile_hist, int num_tiles,
                                                       uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t s_tmp[WAVES];
    uint32_t* row = tile_hist + (size_t)blockIdx.x * num_tiles;
    uint32_t carry = 0;
    constexpr int IPT = 8;
    for (int base = 0; base < num_tiles; base += THREADS * IPT) {
        uint32_t v[IPT];
        uint32_t sum = 0;
        const int i0 = base + threadIdx.x * IPT;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            v[j] = (i0 + j < num_tiles) ? row[i0 + j] : 0u;
            sum += v[j];
        }
        uint32_t tot;
        uint32_t off = carry + block_excl_scan(sum, s_tmp, &tot);
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            if (i0 + j < num_tiles) row[i0 + j] = off;
            off += v[j];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

template <typename KA, bool IMPLICIT_VALS, bool WRITE_KEYS>
__global__ __launch_bounds__(THREADS) void scatter_kernel(const typename KA::In* __restrict__ keys_in,
                                                          const uint32_t* __restrict__ vals_in,
                                                          typename KA::Carry* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                          const uint32_t* __restrict__ tile_hist_scanned,
                                                          const uint32_t* __restrict__ digit_total, int num_tiles) {
    using Carry = typename KA::Carry;
    __shared__ Carry s_keys[TILE];
    __shared__ uint32_t s_vals[TILE];
    __shared__ uint32_t s_whist[WAVES][RADIX];
    __shared__ uint32_t s_tile_start[RADIX];
    __shared__ uint32_t s_glob[RADIX];
    __shared__ uint32_t s_tmp[WAVES];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < WAVES * RADIX; i += THREADS) (&s_whist[0][0])[i] = 0;
    __syncthreads();

    const int64_t base = (int64_t)blockIdx.x * TILE;
    const int64_t wave_base = base + (int64_t)wave * ROUNDS * 64;
    const uint64_t lanes_below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    Carry key[ROUNDS];
    uint32_t rank[ROUNDS];
    volatile uint32_t* whist = &s_whist[wave][0];

    // Phase 1: stable rank of each key among equal digits of its wave (rows of 64 keys in memory order).
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wave_base + r * 64 + lane;
        const bool valid = i < n;
        Carry k = valid ? KA::load(keys_in, i) : (Carry)0;
        key[r] = k;
        const uint32_t d = KA::digit(k, shift);
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t below = __popcll(m & lanes_below);
        const uint32_t cnt = __popcll(m);
        const uint32_t prior = whist[d];
        if (valid && below == 0) whist[d] = prior + cnt;
        rank[r] = prior + below;
    }
    __syncthreads();

    // Phase 2: per-digit wave offsets, tile-local digit starts, global run bases.
    {
        const int d = tid;
        uint32_t off = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            uint32_t c = s_whist[w][d];
            s_whist[w][d] = off;
            off += c;
        }
        uint32_t tile_start = block_excl_scan(off, s_tmp, nullptr);
        uint32_t digit_base = block_excl_scan(digit_total[d], s_tmp, nullptr);
        s_tile_start[d] = tile_start;
        s_glob[d] = digit_base + tile_hist_scanned[(size_t)d * num_tiles + blockIdx.x] - tile_start;
    }
    __syncthreads();

    // Phase 3: place (key, value) at its tile-local sorted position.
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wave_base + r * 64 + lane;
        if (i < n) {
            const uint32_t d = KA::digit(key[r], shift);
            const uint32_t pos = s_tile_start[d] + s_whist[wave][d] + rank[r];
            s_keys[pos] = key[r];
            s_vals[pos] = IMPLICIT_VALS ? (uint32_t)i : vals_in[i];
        }
    }
    __syncthreads();

    // Phase 4: write runs. Position i of the tile-sorted buffer goes to s_glob[digit] + i.
    const int lim = (int)((n - base < TILE) ? (n - base) : TILE);
#pragma unroll 4
    for (int i = tid; i < lim; i += THREADS) {
        const Carry k = s_keys[i];
        const uint32_t d = KA::digit(k, shift);
        const uint32_t g = s_glob[d] + (uint32_t)i;
        if (WRITE_KEYS) keys_out[g] = k;
        vals_out[g] = s_vals[i];
    }
}

template <typename KA, bool IMPLICIT_VALS, bool WRITE_KEYS>
inline int run_pass(const typename KA::In* keys_in, const uint32_t* vals_in, typename KA::Carry* keys_out,
                    uint32_t* vals_out, int64_t n, int shift, uint32_t* tile_hist, uint32_t* digit_total,
                    int num_tiles, hipStream_t stream) {
    hipLaunchKernelGGL((hist_kernel<KA>), dim3(num_tiles), dim3(THREADS), 0, stream, keys_in, n, shift, tile_hist,
                       num_tiles);
    hipLaunchKernelGGL(scan_kernel, dim3(RADIX), dim3(THREADS), 0, stream, tile_hist, num_tiles, digit_total);
    hipLaunchKernelGGL((scatter_kernel<KA, IMPLICIT_VALS, WRITE_KEYS>), dim3(num_tiles), dim3(THREADS), 0, stream,
                       keys_in, vals_in, keys_out, vals_out, n, shift, tile_hist, digit_total, num_tiles);
    return gnnops_check_launch("radix pass");
}

}  // namespace sortengine
