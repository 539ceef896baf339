// sort_engine_impl.h — stable LSD radix-sort pass for gfx950 (64-lane waves), key/value pairs.
//
// One pass = three launches:
//   hist_kernel     grid = tiles        per-tile digit histogram -> tile_hist[digit][tile]
//   scan_kernel     grid = 256 digits   exclusive scan of tile_hist[digit][*], digit totals
//   scatter_kernel  grid = tiles        stable rank of every key inside its tile, reorder through LDS,
//                                       then write each digit's run to its global position (runs are
//                                       contiguous, so stores coalesce)
// Ranking (scatter_kernel phase 1): each wave owns 16 consecutive rows of 64 keys. Per row, eight
// ballots give every lane the mask of lanes holding the same digit; the lowest such lane adds the group
// size to the wave's LDS counter with ONE returning ds_add per distinct digit, and the others fetch the
// returned base with a bpermute. The 16 rows' atomics are issued back to back (LDS executes a wave's
// operations in order, so row r+1 sees row r's add) and consumed afterwards, so no row waits on LDS.
// Keys are u32 or u64 with an 8-bit digit taken at `shift`; values are u32 (positions < 2^32).
// Tile = 8 waves x 16 rows x 64 lanes = 8192 keys; 64 KiB of LDS for the (key, value) reorder image
// (u32 keys), two workgroups = 16 waves per CU.
#pragma once
#include "common.h"
#include "sort_engine.h"

namespace sortengine {

constexpr int HIST_THREADS = 256;
constexpr int HIST_WAVES = HIST_THREADS / 64;

// --- key adapters: how a stored key yields the current 8-bit digit and how it is carried ---
struct KeyI64Low32 {
    template <typename P> __device__ static inline P open(P p) { return p; }   // what load() reads from: the key array itself  // first pass of the plan builder: int64 index -> u32 key
    using In = int64_t;
    using Carry = uint32_t;
    // the WHOLE 8-byte element is loaded (the asm barrier keeps the compiler from narrowing the load to the low dword): a
    // wave then reads 512 contiguous bytes per instruction instead of 64 dwords at a stride of 8 bytes
    __device__ static inline Carry load(const In* p, int64_t i) {
        int64_t v = p[i];
        asm("" : "+v"(v));   // not volatile: the loads of a tile may still be issued together
        return (uint32_t)v;
    }
    __device__ static inline uint32_t digit(Carry k, int shift) { return (k >> shift) & 255u; }
};
struct KeyU32 {
    template <typename P> __device__ static inline P open(P p) { return p; }   // what load() reads from: the key array itself
    using In = uint32_t;
    using Carry = uint32_t;
    __device__ static inline Carry load(const In* p, int64_t i) { return p[i]; }
    __device__ static inline uint32_t digit(Carry k, int shift) { return (k >> shift) & 255u; }
};
struct KeyU64 {
    template <typename P> __device__ static inline P open(P p) { return p; }   // what load() reads from: the key array itself
    using In = uint64_t;
    using Carry = uint64_t;
    __device__ static inline Carry load(const In* p, int64_t i) { return p[i]; }
    __device__ static inline uint32_t digit(Carry k, int shift) { return (uint32_t)(k >> shift) & 255u; }
};
// 1-D scatter with the value carried beside its destination (scatter1d.hip): the key is built from two arrays.
template <typename T>
struct KeyDstVal {
    using In = DstValSrc<T>;
    using Carry = uint64_t;
    // The descriptor is read ONCE per kernel (open) into registers, and its two arrays are addressed as GLOBAL memory: a
    // pointer loaded from memory has no address space, so loads through it are flat loads that wait on both counters, and a
    // field read under the range test puts a scalar load and a wait inside every element's load — the first version of
    // this pass ran its 32 loads per lane one after the other (16.5 ms against 10.2 for the second pass, same bytes).
    struct Src {
        const __attribute__((address_space(1))) int64_t* idx;
        const __attribute__((address_space(1))) T* val;
        uint64_t n_dst;
        uint32_t sentinel;
    };
    __device__ static inline Src open(const In* p) {
        Src s;
        s.idx = (const __attribute__((address_space(1))) int64_t*)p->idx;
        s.val = (const __attribute__((address_space(1))) T*)p->val;
        s.n_dst = (uint64_t)p->n_dst;
        s.sentinel = p->sentinel;
        return s;
    }
    __device__ static inline Carry load(const Src& s, int64_t i) {
        const uint64_t d = (uint64_t)s.idx[i];
        float v;
        if constexpr (sizeof(T) == 4) {
            v = *reinterpret_cast<const __attribute__((address_space(1))) float*>(s.val + i);
        } else {   // 16-bit storage: the two bytes travel as an integer (class types do not copy out of an address space), then widen
            const uint16_t bits = *reinterpret_cast<const __attribute__((address_space(1))) uint16_t*>(s.val + i);
            T raw;
            __builtin_memcpy(&raw, &bits, 2);
            v = Elem<T>::load(&raw);
        }
        const uint32_t hi = d < s.n_dst ? (uint32_t)d : s.sentinel;
        return ((uint64_t)hi << 32) | (uint64_t)__float_as_uint(v);
    }
    __device__ static inline uint32_t digit(Carry k, int shift) { return (uint32_t)(k >> shift) & 255u; }
};
// torch.sort on fp32: order-preserving map float -> u32 (negatives flipped entirely, positives get the
// sign bit); NaNs sort last like torch (any NaN maps to the top key).
struct KeyF32 {
    template <typename P> __device__ static inline P open(P p) { return p; }   // what load() reads from: the key array itself
    using In = float;
    using Carry = uint32_t;
    __device__ static inline Carry load(const In* p, int64_t i) {
        uint32_t u = __float_as_uint(p[i]);
        if (u == 0x80000000u) u = 0u;  // -0.0 compares equal to +0.0
        if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    __device__ static inline uint32_t digit(Carry k, int shift) { return (k >> shift) & 255u; }
};

template <typename KA>
__global__ __launch_bounds__(HIST_THREADS) void hist_kernel(const typename KA::In* __restrict__ keys_arg, int64_t n,
                                                            int shift, uint32_t* __restrict__ tile_hist,
                                                            int num_tiles) {
    __shared__ uint32_t h[HIST_WAVES][RADIX];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < HIST_WAVES * RADIX; i += HIST_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const int tile = (int)xcd_contiguous(blockIdx.x, gridDim.x);   // neighbouring tiles share tile_hist lines: same L2
    const int64_t base = (int64_t)tile * TILE;
    const int lim = (int)((n - base < TILE) ? (n - base) : TILE);
    // four consecutive keys per lane and load (16 B for u32 keys): TILE and `base` are multiples of 4. A full tile issues ALL
    // its loads before the first histogram update: a load under a per-lane condition is waited for inside the branch
    // (one group of loads in flight at a time — the first version ran at 3.8 TB/s for that reason).
    constexpr int GROUPS = TILE / (HIST_THREADS * 4);
    const auto keys = KA::open(keys_arg);
    if (lim == TILE) {
        typename KA::Carry k[GROUPS][4];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) k[g][j] = KA::load(keys, base + (int64_t)(g * HIST_THREADS + tid) * 4 + j);
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(&h[wave][KA::digit(k[g][j], shift)], 1u);
    } else {
        for (int i = tid; i < lim; i += HIST_THREADS) atomicAdd(&h[wave][KA::digit(KA::load(keys, base + i), shift)], 1u);
    }
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < HIST_WAVES; ++w) t += h[w][tid];
    tile_hist[(size_t)tid * num_tiles + tile] = t;
}

// One block per digit: in-place exclusive scan across tiles, total -> digit_total[digit].
__global__ __launch_bounds__(HIST_THREADS) void scan_kernel(uint32_t* __restrict__ tile_hist, int num_tiles,
                                                            uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t s_tmp[HIST_WAVES];
    uint32_t* row = tile_hist + (size_t)blockIdx.x * num_tiles;
    uint32_t carry = 0;
    constexpr int IPT = 8;
    for (int base = 0; base < num_tiles; base += HIST_THREADS * IPT) {
        uint32_t v[IPT];
        uint32_t sum = 0;
        const int i0 = base + threadIdx.x * IPT;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            v[j] = (i0 + j < num_tiles) ? row[i0 + j] : 0u;
            sum += v[j];
        }
        uint32_t tot;
        uint32_t off = carry + block_excl_scan_u32<HIST_WAVES>(sum, s_tmp, &tot);
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
__global__ __launch_bounds__(THREADS, (sizeof(typename KA::Carry) == 4 ? 4 : 2)) void scatter_kernel(const typename KA::In* __restrict__ keys_arg,
                                                          const uint32_t* __restrict__ vals_in,
                                                          typename KA::Carry* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                          const uint32_t* __restrict__ tile_hist_scanned,
                                                          const uint32_t* __restrict__ digit_total, int num_tiles) {
    using Carry = typename KA::Carry;
    constexpr bool PACKED = sizeof(Carry) == 4;  // (value << 32 | key) in one 8-byte LDS slot
    __shared__ __attribute__((aligned(16))) unsigned char s_image[TILE * (sizeof(Carry) + 4)];
    __shared__ uint32_t s_whist[WAVES][RADIX];
    __shared__ uint32_t s_glob[RADIX];
    __shared__ uint32_t s_tmp[WAVES];
    uint64_t* s_pairs = reinterpret_cast<uint64_t*>(s_image);                          // PACKED
    Carry* s_keys = reinterpret_cast<Carry*>(s_image);                                 // !PACKED
    uint32_t* s_vals = reinterpret_cast<uint32_t*>(s_image + TILE * sizeof(Carry));    // !PACKED

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < WAVES * RADIX; i += THREADS) (&s_whist[0][0])[i] = 0;
    __syncthreads();

    // neighbouring tiles write neighbouring digit runs (and read neighbouring tile_hist slots): keep them on one XCD
    const int tile = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int64_t base = (int64_t)tile * TILE;
    const int64_t wave_base = base + (int64_t)wave * ROUNDS * 64;
    const uint64_t lanes_below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t* whist = &s_whist[wave][0];

    Carry key[ROUNDS];
    uint32_t val[ROUNDS];
    const auto keys_in = KA::open(keys_arg);
    if (base + TILE <= n) {  // a full tile (uniform): unconditional loads, all in flight together
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int64_t i = wave_base + r * 64 + lane;
            key[r] = KA::load(keys_in, i);
            if (!IMPLICIT_VALS) val[r] = vals_in[i];
        }
    } else {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int64_t i = wave_base + r * 64 + lane;
            key[r] = (i < n) ? KA::load(keys_in, i) : (Carry)0;
            if (!IMPLICIT_VALS) val[r] = (i < n) ? vals_in[i] : 0u;
        }
    }

    // Phase 1: stable rank of each key among equal digits of its wave (rows of 64 keys in memory order).
    // x[r] holds, in a group's lowest lane, the base returned by the LDS add; in the other lanes
    // `below | leader_lane << 8`. is_leader has bit r set where this lane is row r's group leader.
    uint32_t x[ROUNDS];
    uint32_t is_leader = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wave_base + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = KA::digit(key[r], shift);
        // peers = lanes whose digit equals mine: AND over the 8 digit bits of (ballot(bit) XNOR my bit)
        const uint64_t m = match_digit8(d, __ballot(valid));
        const uint32_t below = __popcll(m & lanes_below);
        if (valid && below == 0) {
            x[r] = atomicAdd(&whist[d], (uint32_t)__popcll(m));
            is_leader |= 1u << r;
        } else {
            x[r] = below | ((uint32_t)(__ffsll((unsigned long long)m) - 1) << 8);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep rows from being interleaved (register pressure)
    }
    uint32_t rank[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const bool lead = (is_leader >> r) & 1u;
        const int from = lead ? lane : (int)((x[r] >> 8) & 63u);
        const uint32_t p = __shfl(x[r], from);
        rank[r] = lead ? p : p + (x[r] & 255u);
    }
    __syncthreads();

    // Phase 2: per-digit wave offsets (+ tile-local digit start), global run bases.
    {
        const int d = tid & (RADIX - 1);
        const bool act = tid < RADIX;
        uint32_t cnts[WAVES];
        uint32_t tot = 0;
        if (act) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                cnts[w] = s_whist[w][d];
                tot += cnts[w];
            }
        }
        const uint32_t tile_start = block_excl_scan_u32<WAVES>(act ? tot : 0u, s_tmp, nullptr);
        const uint32_t digit_base = block_excl_scan_u32<WAVES>(act ? digit_total[d] : 0u, s_tmp, nullptr);
        if (act) {
            uint32_t off = tile_start;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                s_whist[w][d] = off;
                off += cnts[w];
            }
            s_glob[d] = digit_base + tile_hist_scanned[(size_t)d * num_tiles + tile] - tile_start;
        }
    }
    __syncthreads();

    // Phase 3: place (key, value) at its tile-local sorted position.
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wave_base + r * 64 + lane;
        if (i < n) {
            const uint32_t d = KA::digit(key[r], shift);
            const uint32_t pos = whist[d] + rank[r];
            const uint32_t v = IMPLICIT_VALS ? (uint32_t)i : val[r];
            if constexpr (PACKED) {
                s_pairs[pos] = ((uint64_t)v << 32) | (uint64_t)(uint32_t)key[r];
            } else {
                s_keys[pos] = key[r];
                s_vals[pos] = v;
            }
        }
    }
    __syncthreads();

    // Phase 4: write runs. Position i of the tile-sorted image goes to s_glob[digit] + i.
    const int lim = (int)((n - base < TILE) ? (n - base) : TILE);
#pragma unroll 4
    for (int i = tid; i < lim; i += THREADS) {
        Carry k;
        uint32_t v;
        if constexpr (PACKED) {
            const uint64_t p = s_pairs[i];
            k = (Carry)(uint32_t)p;
            v = (uint32_t)(p >> 32);
        } else {
            k = s_keys[i];
            v = s_vals[i];
        }
        const uint32_t g = s_glob[KA::digit(k, shift)] + (uint32_t)i;
        if (WRITE_KEYS) keys_out[g] = k;
        vals_out[g] = v;
    }
}

template <typename KA, bool IMPLICIT_VALS, bool WRITE_KEYS>
inline int run_pass(const typename KA::In* keys_in, const uint32_t* vals_in, typename KA::Carry* keys_out,
                    uint32_t* vals_out, int64_t n, int shift, uint32_t* tile_hist, uint32_t* digit_total,
                    int num_tiles, hipStream_t stream) {
    hipLaunchKernelGGL((hist_kernel<KA>), dim3(num_tiles), dim3(HIST_THREADS), 0, stream, keys_in, n, shift,
                       tile_hist, num_tiles);
    hipLaunchKernelGGL(scan_kernel, dim3(RADIX), dim3(HIST_THREADS), 0, stream, tile_hist, num_tiles, digit_total);
    hipLaunchKernelGGL((scatter_kernel<KA, IMPLICIT_VALS, WRITE_KEYS>), dim3(num_tiles), dim3(THREADS), 0, stream,
                       keys_in, vals_in, keys_out, vals_out, n, shift, tile_hist, digit_total, num_tiles);
    return gnnops_check_launch("radix pass");
}

}  // namespace sortengine
