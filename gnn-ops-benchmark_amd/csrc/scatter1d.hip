// scatter1d.hip — torch_scatter.scatter_min / scatter_max of a LONG 1-D tensor (src [E], index [E], K = 1) when the
// destinations do not fit on chip: the reference's ">= 95 % of an A100-40GB" shapes, data/scatter_min.csv:2 and
// data/scatter_max.csv:32 — 1 472 353 280 fp32 elements, index uniform over as many destinations
// (op_bm_scripts/benchmark_scatter_min.py:15-18, torch_scatter.scatter_min(src, idx, dim)).
//
// The plan path sorts (destination, position) pairs completely (four 8-bit passes at 31 bits) and then gathers
// src[position] at random: 4 useful bytes of every 64-B line (35 ms + 42 ms at that shape). Here the VALUE travels with
// its destination, and the sort stops at buckets of 32768 destinations — what one workgroup's LDS holds:
//
//   1. radix passes over the destination bits above the low 15 only (sort_engine; two passes at 31 bits), key =
//      (destination << 32) | fp32 bits of the value, payload = source position; the first pass builds its keys straight
//      from (index, src) (KeyDstVal): nothing is packed or copied beforehand;
//   2. bounds1d_kernel: one binary search per bucket boundary;
//   3. minmax1d_kernel: a workgroup (1024 threads, 128 KiB of LDS) takes a bucket —
//        a. table[d] = min / max over the bucket's pairs of the ORDERED u32 image of the value (LDS atomics; -0.0 counts as
//           +0.0, NaNs and the reduce's identity, +inf / -inf, never win: the conventions of scatter_elem.hip / segment.hip);
//        b. out[d] = that value (0 for a group nothing reached), stored;
//        c. the table becomes positions: every pair whose value EQUALS out[d] does an LDS atomic min of
//           (position << 1 | sign bit) — the smallest position among the winners, and whether that winner was a -0.0;
//        d. arg[d] = position (E for a group nothing reached); out[d] gets its sign back if the winner was -0.0.
//      Compares are exact, so the result is bit-identical to the sequential loop of the oracle: smallest position on ties.
//
// Sums / means / products (gnnops_scatter1d_sum) keep the sequential order of the oracle — LDS float atomics would not —
// so there the sort goes one pass further, to buckets of 256 destinations (bits 8 and up: three passes at 31 bits), and a
// 256-thread workgroup finishes a bucket as bucket.hip does for rows: a stable counting sort of its (destination & 255, value)
// pairs in LDS (ballot ranking, chunks of 4096 in source order), then thread d adds destination d's values one after the
// other, in source position order: bit-identical to the sequential loop. No position travels (means count their lists).
//
// HBM-bound, and no random access anywhere: per element ~100 B of streamed traffic (first pass 12 + 12 read, 12 written;
// second 8 + 12 read, 12 written; reduce 12 + 12 read) + 12 B written per destination.
#include "common.h"
#include "sort_engine.h"

namespace {

constexpr int LOW = 15, BUCKET = 1 << LOW;     // destinations per bucket = LDS table entries (4 B each: 128 KiB)
constexpr int RTHREADS = 1024;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int bits_of(int64_t v) {   // smallest b with (1 << b) > v
    int b = 0;
    while (b < 40 && ((int64_t)1 << b) <= v) ++b;
    return b;
}

struct Layout1d {
    size_t keys_a, keys_b, vals_a, vals_b, tile_hist, digit_total, bptr, desc, total;
};

inline Layout1d layout1d(int64_t E, int64_t N) {
    Layout1d l{};
    const size_t tiles = (size_t)gnnops_cdiv(E > 0 ? E : 1, sortengine::TILE);
    size_t o = 0;
    l.keys_a = o; o += align_up((size_t)E * 8, 256);
    l.keys_b = o; o += align_up((size_t)E * 8, 256);
    l.vals_a = o; o += align_up((size_t)E * 4, 256);
    l.vals_b = o; o += align_up((size_t)E * 4, 256);
    l.tile_hist = o; o += align_up(256 * tiles * 4, 256);
    l.digit_total = o; o += 256 * 4;
    l.bptr = o; o += align_up(((size_t)gnnops_cdiv(N > 0 ? N : 1, 256) + 2) * 8, 256);   // sized for the finer (256-row) buckets of the sums
    l.desc = o; o += 256;
    l.total = o;
    return l;
}

// the sentinel destination of ids outside [0, N): the first id past the last bucket (a bucket of its own, never reduced)
inline int64_t sentinel_of(int64_t N) { return gnnops_cdiv(N, BUCKET) * BUCKET; }
inline int passes_of(int64_t N, int low = LOW) { return (bits_of(sentinel_of(N)) - low + 7) / 8; }

template <typename T>
__global__ void set_desc_kernel(sortengine::DstValSrc<T>* d, const int64_t* idx, const T* val, int64_t n_dst, uint32_t sentinel) {
    d->idx = idx; d->val = val; d->n_dst = n_dst; d->sentinel = sentinel;
}

// bptr[b] = first position whose (key >> 32) >> low >= b; bptr[NB] = first position of the sentinel bucket.
__global__ void bounds1d_kernel(const uint64_t* __restrict__ keys, int64_t E, int64_t NB, int64_t* __restrict__ bptr, int low) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > NB) return;
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(keys[mid] >> (32 + low)) < b) lo = mid + 1; else hi = mid;
    }
    bptr[b] = lo;
}

// order-preserving u32 image of a float, -0.0 folded onto +0.0
__device__ inline uint32_t ordered_of(uint32_t u) {
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float float_of_ordered(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// HOLD keys per thread stay in REGISTERS between the value phase and the position phase (24 x 1024 = 24576 pairs — more spills: the 128 registers a lane of a 1024-thread workgroup has hold 48 for the keys — of the mean
// 32768-pair bucket of the reference's shape, E = N; what a bucket holds beyond that is streamed in both phases), so a bucket's keys are read from HBM once; the
// positions are read in the second phase only.
constexpr int HOLD = 24;

template <typename T, bool IS_MIN>
__global__ __launch_bounds__(RTHREADS) void minmax1d_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos,
                                                          const int64_t* __restrict__ bptr, T* __restrict__ out,
                                                          int64_t* __restrict__ arg_out, int64_t E, int64_t N, int64_t NB) {
    extern __shared__ uint32_t tab[];   // BUCKET entries
    const int tid = threadIdx.x;
    constexpr uint32_t EMPTY = IS_MIN ? 0xffffffffu : 0u;
    const uint32_t ident = IS_MIN ? 0x7f800000u : 0xff800000u;   // +inf / -inf: the reduce's identity never wins
    constexpr int U = 4;
    auto skip = [&](uint32_t bits) { return (bits & 0x7fffffffu) > 0x7f800000u || bits == ident; };   // NaN, or the identity
    auto value_step = [&](uint64_t k) {
        const uint32_t bits = (uint32_t)k;
        if (skip(bits)) return;
        const uint32_t d = (uint32_t)(k >> 32) & (BUCKET - 1);
        if (IS_MIN) atomicMin(&tab[d], ordered_of(bits)); else atomicMax(&tab[d], ordered_of(bits));
    };
    for (int64_t b = blockIdx.x; b < NB; b += gridDim.x) {
        const int64_t beg = bptr[b], end = bptr[b + 1];
        const int64_t d0 = b << LOW;
        const int nd = (int)((N - d0 < BUCKET) ? (N - d0) : BUCKET);
        const T* outr = out + d0;   // this bucket's values, read back in phase c (stored by this workgroup in phase b)
        auto position_step = [&](uint64_t k, uint32_t p) {
            const uint32_t bits = (uint32_t)k;
            if (skip(bits)) return;
            const uint32_t d = (uint32_t)(k >> 32) & (BUCKET - 1);
            const float won = Elem<T>::load(outr + d);
            if (__uint_as_float(bits) == won) atomicMin(&tab[d], (p << 1) | (bits >> 31));
        };
        for (int d = tid; d < BUCKET; d += RTHREADS) tab[d] = EMPTY;
        // the keys this thread holds: loads unconditional on clamped positions, all in flight while the table is initialised.
        // Positions RELATIVE to the bucket, 32-bit (E < 2^31): one uniform base + a 32-bit lane offset per load instead of 32
        // address pairs (which, beside the 64 registers of the keys themselves, spilled)
        const uint64_t* kb = keys + beg;
        const uint32_t* pb = pos + beg;
        const uint32_t cnt = (uint32_t)(end - beg);
        const uint32_t lastr = cnt ? cnt - 1 : 0;   // an empty bucket reads one element (inside the workspace), uses none
        uint64_t held[HOLD];
#pragma unroll
        for (int u = 0; u < HOLD; ++u) {
            const uint32_t r = (uint32_t)tid + (uint32_t)u * RTHREADS;
            held[u] = kb[r < cnt ? r : lastr];
        }
        __syncthreads();
        // a. the winning VALUE per destination
#pragma unroll
        for (int u = 0; u < HOLD; ++u)
            if ((uint32_t)tid + (uint32_t)u * RTHREADS < cnt) value_step(held[u]);
        for (uint32_t r0 = (uint32_t)HOLD * RTHREADS + tid; r0 < cnt; r0 += RTHREADS * U) {
            uint64_t k[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = r0 + u * RTHREADS;
                k[u] = kb[r < cnt ? r : lastr];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (r0 + u * RTHREADS < cnt) value_step(k[u]);
        }
        __syncthreads();
        // b. store the values; c. the table turns into positions
        for (int d = tid; d < BUCKET; d += RTHREADS) {
            const uint32_t o = tab[d];
            if (d < nd) Elem<T>::store(out + d0 + d, o == EMPTY ? 0.f : float_of_ordered(o));
        }
        __syncthreads();   // orders the stores above before this workgroup's loads of them below
        for (int d = tid; d < BUCKET; d += RTHREADS) tab[d] = 0xffffffffu;
        __syncthreads();
        {   // the held keys' positions, PC at a time (all HOLD of them beside the keys would not fit 128 registers)
            constexpr int PC = 8;
#pragma unroll
            for (int u0 = 0; u0 < HOLD; u0 += PC) {
                uint32_t hp[PC];
#pragma unroll
                for (int u = 0; u < PC; ++u) {
                    const uint32_t r = (uint32_t)tid + (uint32_t)(u0 + u) * RTHREADS;
                    hp[u] = pb[r < cnt ? r : lastr];
                }
#pragma unroll
                for (int u = 0; u < PC; ++u)
                    if ((uint32_t)tid + (uint32_t)(u0 + u) * RTHREADS < cnt) position_step(held[u0 + u], hp[u]);
            }
        }
        for (uint32_t r0 = (uint32_t)HOLD * RTHREADS + tid; r0 < cnt; r0 += RTHREADS * U) {
            uint64_t k[U];
            uint32_t p[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = r0 + u * RTHREADS;
                const uint32_t rc = r < cnt ? r : lastr;
                k[u] = kb[rc];
                p[u] = pb[rc];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (r0 + u * RTHREADS < cnt) position_step(k[u], p[u]);
        }
        __syncthreads();
        // d. positions out; a winner that was -0.0 gives its sign back
        for (int d = tid; d < nd; d += RTHREADS) {
            const uint32_t w = tab[d];
            arg_out[d0 + d] = (w == 0xffffffffu) ? E : (int64_t)(w >> 1);
            if (w != 0xffffffffu && (w & 1u) && Elem<T>::load(outr + d) == 0.f) Elem<T>::store(out + d0 + d, -0.f);
        }
        __syncthreads();   // the table is re-initialised for the next bucket
    }
}

// ---- sums / means / products: buckets of 256 destinations, finished in source order -------------------------------------
constexpr int SLOW = 8, SROWS = 1 << SLOW;      // destinations per bucket
constexpr int STHREADS = 256, SWAVES = STHREADS / 64, SROUNDS = 4, SCAP = STHREADS * SROUNDS;   // pairs sorted on chip at a time: a bucket of the
// reference shape holds ~256; four rounds keep the kernel at 64 registers, i.e. EIGHT workgroups per CU — the finish is a chain of
// barriers and dependent LDS steps per bucket (5.75 M buckets), and what hides that latency is other buckets in flight

// Stable counting sort, in LDS, of the n <= SCAP pairs at keys[cbeg .. cbeg + n) by (destination & 255): s_val gets the VALUE
// bits grouped by destination in their original order, s_rowptr[0..256] the group boundaries (bucket.hip's sort_chunk with
// the value instead of a position as payload). Ranking as in sort_engine_impl.h: per row of 64 pairs, eight ballots give every
// lane its equal-key lanes; the lowest of them does ONE returning LDS add. Returns the size of group `threadIdx.x`.
__device__ inline uint32_t sort_chunk64(const uint64_t* __restrict__ keys, int64_t cbeg, int n, uint32_t* s_val, uint32_t* s_whist,
                                        int32_t* s_rowptr, uint32_t* s_tmp) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint64_t lanes_below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t* whist = s_whist + wave * 256;   // zeroed by the caller behind a barrier (sum1d_kernel: one barrier fewer per chunk)
    const int rounds_n = (n + STHREADS - 1) / STHREADS;   // rows of 64 per wave
    const int wave_base = wave * rounds_n * 64;
    uint32_t dg[SROUNDS], vv[SROUNDS], rk[SROUNDS];
    uint32_t is_leader = 0;
#pragma unroll
    for (int r = 0; r < SROUNDS; ++r) {
        dg[r] = 0; vv[r] = 0; rk[r] = 0;
        if (r < rounds_n) {
            const int i = wave_base + r * 64 + lane;
            const bool valid = i < n;
            if (valid) {
                const uint64_t k = keys[cbeg + i];
                dg[r] = (uint32_t)(k >> 32) & (SROWS - 1);
                vv[r] = (uint32_t)k;
            }
            const uint32_t d = dg[r];
            const uint64_t m = match_digit8(d, __ballot(valid));   // valid lanes with my key
            const uint32_t below = __popcll(m & lanes_below);
            if (valid && below == 0) {
                rk[r] = atomicAdd(&whist[d], (uint32_t)__popcll(m));  // rank of the group inside this wave
                is_leader |= 1u << r;
            } else {
                rk[r] = below | ((uint32_t)(__ffsll((unsigned long long)m) - 1) << 16);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < SROUNDS; ++r) {
        if (r < rounds_n) {
            const bool lead = (is_leader >> r) & 1u;
            const int from = lead ? lane : (int)((rk[r] >> 16) & 63u);
            const uint32_t p = __shfl(rk[r], from);
            rk[r] = lead ? p : p + (rk[r] & 0xffffu);
        }
    }
    __syncthreads();
    uint32_t tot = 0;   // key offsets: exclusive over waves, then over keys (thread d owns key d)
#pragma unroll
    for (int w = 0; w < SWAVES; ++w) {
        const uint32_t c = s_whist[w * 256 + tid];
        s_whist[w * 256 + tid] = tot;
        tot += c;
    }
    // exclusive scan of `tot` over the 256 threads: wave scan, wave totals through s_tmp — ONE barrier (s_tmp is next written
    // a chunk later, behind the caller's barriers)
    const uint32_t incl = wave_incl_scan_u32(tot);
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t start = incl - tot;
#pragma unroll
    for (int w = 0; w < SWAVES; ++w)
        if (w < wave) start += s_tmp[w];
#pragma unroll
    for (int w = 0; w < SWAVES; ++w) s_whist[w * 256 + tid] += start;
    s_rowptr[tid] = (int32_t)start;
    if (tid == SROWS - 1) s_rowptr[SROWS] = (int32_t)(start + tot);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SROUNDS; ++r) {
        if (r < rounds_n) {
            const int i = wave_base + r * 64 + lane;
            if (i < n) s_val[whist[dg[r]] + rk[r]] = vv[r];
        }
    }
    return tot;
}

// R: GNNOPS_SUM or GNNOPS_MUL; is_mean divides the sum by max(count, 1).
template <typename T, int R>
__global__ __launch_bounds__(STHREADS, 8) void sum1d_kernel(const uint64_t* __restrict__ keys, const int64_t* __restrict__ bptr,
                                                         T* __restrict__ out, int64_t N, int64_t NB, int is_mean) {
    __shared__ uint32_t s_val[SCAP];
    __shared__ uint32_t s_whist[SWAVES * 256];
    __shared__ int32_t s_rowptr[SROWS + 1];
    __shared__ uint32_t s_tmp[SWAVES];
    const int tid = threadIdx.x;
    for (int i = tid; i < SWAVES * 256; i += STHREADS) s_whist[i] = 0;
    for (int64_t b = blockIdx.x; b < NB; b += gridDim.x) {
        const int64_t beg = bptr[b], end = bptr[b + 1];
        float acc = Red<R>::identity();
        uint32_t cnt = 0;
        for (int64_t cbeg = beg; cbeg < end; cbeg += SCAP) {
            const int n = (int)((end - cbeg < SCAP) ? (end - cbeg) : SCAP);
            __syncthreads();   // the previous chunk's / bucket's readers are done with s_val, s_rowptr; s_whist is zero
            cnt += sort_chunk64(keys, cbeg, n, s_val, s_whist, s_rowptr, s_tmp);
            __syncthreads();   // everyone has placed its values (the last readers of s_whist)
            for (int i = tid; i < SWAVES * 256; i += STHREADS) s_whist[i] = 0;   // for the next chunk, ordered by its leading barrier
            const int32_t jb = s_rowptr[tid], je = s_rowptr[tid + 1];
            for (int32_t j = jb; j < je; ++j) acc = Red<R>::apply(acc, __uint_as_float(s_val[j]));   // source order: chunks in order, stable inside
        }
        const int64_t d = (b << SLOW) + tid;
        if (d < N) {
            if (is_mean) acc = acc / (float)(cnt < 1 ? 1 : cnt);
            Elem<T>::store(out + d, acc);
        }
    }
}

template <typename T>
int run1d_sum(const void* src, const int64_t* index, void* out, int64_t E, int64_t N, int reduce, void* workspace, hipStream_t stream,
              int (*first_pass)(const sortengine::DstValSrc<T>*, uint64_t*, uint32_t*, int64_t, int, uint32_t*, uint32_t*, int,
                                hipStream_t)) {
    const Layout1d l = layout1d(E, N);
    char* w = (char*)workspace;
    uint64_t* kbuf[2] = {(uint64_t*)(w + l.keys_a), (uint64_t*)(w + l.keys_b)};
    uint32_t* vbuf[2] = {(uint32_t*)(w + l.vals_a), (uint32_t*)(w + l.vals_b)};
    uint32_t* tile_hist = (uint32_t*)(w + l.tile_hist);
    uint32_t* digit_total = (uint32_t*)(w + l.digit_total);
    int64_t* bptr = (int64_t*)(w + l.bptr);
    auto* desc = (sortengine::DstValSrc<T>*)(w + l.desc);
    const int tiles = (int)gnnops_cdiv(E, sortengine::TILE);
    const int64_t NB = gnnops_cdiv(N, SROWS);
    const int64_t sentinel = NB * SROWS;                       // ids outside [0, N): a bucket of their own behind the last one
    const int passes = (bits_of(sentinel) - SLOW + 7) / 8;
    hipLaunchKernelGGL(set_desc_kernel<T>, dim3(1), dim3(1), 0, stream, desc, index, (const T*)src, N, (uint32_t)sentinel);
    const uint64_t* kin = nullptr;
    const uint32_t* vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        const int shift = 32 + SLOW + 8 * p;
        const int rc = p == 0 ? first_pass(desc, kbuf[0], vbuf[0], E, shift, tile_hist, digit_total, tiles, stream)
                              : sortengine::pass_u64(kin, vin, kbuf[p & 1], vbuf[p & 1], E, shift, tile_hist, digit_total, tiles, stream);
        if (rc != GNNOPS_OK) return rc;
        kin = kbuf[p & 1]; vin = vbuf[p & 1];
    }
    hipLaunchKernelGGL(bounds1d_kernel, dim3((unsigned)gnnops_cdiv(NB + 1, 256)), dim3(256), 0, stream, kin, E, NB, bptr, SLOW);
    const int grid = gnnops_grid_cap(NB, 256 * 16);
    if (reduce == GNNOPS_MUL)
        hipLaunchKernelGGL((sum1d_kernel<T, GNNOPS_MUL>), dim3(grid), dim3(STHREADS), 0, stream, kin, bptr, (T*)out, N, NB, 0);
    else
        hipLaunchKernelGGL((sum1d_kernel<T, GNNOPS_SUM>), dim3(grid), dim3(STHREADS), 0, stream, kin, bptr, (T*)out, N, NB,
                           reduce == GNNOPS_MEAN ? 1 : 0);
    return gnnops_check_launch("scatter1d_sum");
}

template <typename T>
int run1d(const void* src, const int64_t* index, void* out, int64_t* arg_out, int64_t E, int64_t N, int reduce, void* workspace,
          hipStream_t stream, int (*first_pass)(const sortengine::DstValSrc<T>*, uint64_t*, uint32_t*, int64_t, int, uint32_t*, uint32_t*,
                                                int, hipStream_t)) {
    const Layout1d l = layout1d(E, N);
    char* w = (char*)workspace;
    uint64_t* kbuf[2] = {(uint64_t*)(w + l.keys_a), (uint64_t*)(w + l.keys_b)};
    uint32_t* vbuf[2] = {(uint32_t*)(w + l.vals_a), (uint32_t*)(w + l.vals_b)};
    uint32_t* tile_hist = (uint32_t*)(w + l.tile_hist);
    uint32_t* digit_total = (uint32_t*)(w + l.digit_total);
    int64_t* bptr = (int64_t*)(w + l.bptr);
    auto* desc = (sortengine::DstValSrc<T>*)(w + l.desc);
    const int tiles = (int)gnnops_cdiv(E, sortengine::TILE);
    const int passes = passes_of(N);
    hipLaunchKernelGGL(set_desc_kernel<T>, dim3(1), dim3(1), 0, stream, desc, index, (const T*)src, N, (uint32_t)sentinel_of(N));
    const uint64_t* kin = nullptr;
    const uint32_t* vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        const int shift = 32 + LOW + 8 * p;
        const int rc = p == 0 ? first_pass(desc, kbuf[0], vbuf[0], E, shift, tile_hist, digit_total, tiles, stream)
                              : sortengine::pass_u64(kin, vin, kbuf[p & 1], vbuf[p & 1], E, shift, tile_hist, digit_total, tiles, stream);
        if (rc != GNNOPS_OK) return rc;
        kin = kbuf[p & 1]; vin = vbuf[p & 1];
    }
    const int64_t NB = gnnops_cdiv(N, BUCKET);
    hipLaunchKernelGGL(bounds1d_kernel, dim3((unsigned)gnnops_cdiv(NB + 1, 256)), dim3(256), 0, stream, kin, E, NB, bptr, LOW);
    const int grid = (int)(NB < 256 ? NB : 256);
    const size_t lds = (size_t)BUCKET * 4;
    static bool configured[2] = {false, false};
    const bool is_min = reduce == GNNOPS_MIN;
    const void* fn = is_min ? reinterpret_cast<const void*>(&minmax1d_kernel<T, true>) : reinterpret_cast<const void*>(&minmax1d_kernel<T, false>);
    if (!configured[is_min]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return gnnops_check_launch("scatter1d attribute");
        configured[is_min] = true;
    }
    if (is_min)
        hipLaunchKernelGGL((minmax1d_kernel<T, true>), dim3(grid), dim3(RTHREADS), lds, stream, kin, vin, bptr, (T*)out, arg_out, E, N, NB);
    else
        hipLaunchKernelGGL((minmax1d_kernel<T, false>), dim3(grid), dim3(RTHREADS), lds, stream, kin, vin, bptr, (T*)out, arg_out, E, N, NB);
    return gnnops_check_launch("scatter1d_minmax");
}

}  // namespace

extern "C" size_t gnnops_scatter1d_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return 0;
    return layout1d(E, N).total;
}

// out[n] = min / max of src[e] over index[e] == n (0 where nothing arrives), arg_out[n] = smallest such e holding it (E where
// nothing arrives). GNNOPS_EUNSUPPORTED — take gnnops_segment_reduce — unless the shape is the one this form is for:
// BUCKET < N < 2^31 - BUCKET, 0 < E < 2^31.
extern "C" int gnnops_scatter1d_minmax(const void* src, const int64_t* index, void* out, int64_t* arg_out, int64_t E, int64_t N,
                                       int dtype, int reduce, void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    GNNOPS_REQUIRE(E >= 0 && N >= 0, GNNOPS_EINVAL, "scatter1d_minmax: negative size");
    GNNOPS_REQUIRE(reduce == GNNOPS_MIN || reduce == GNNOPS_MAX, GNNOPS_EUNSUPPORTED, "scatter1d_minmax: reduce %d (min / max only)", reduce);
    GNNOPS_REQUIRE(N > BUCKET && N < ((int64_t)1 << 31) - BUCKET && E > 0 && E < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED,
                   "scatter1d_minmax: shape outside this form (E=%lld N=%lld)", (long long)E, (long long)N);
    GNNOPS_REQUIRE(src && index && out && arg_out, GNNOPS_EINVAL, "scatter1d_minmax: null pointer");
    const Layout1d l = layout1d(E, N);
    GNNOPS_REQUIRE(workspace && workspace_bytes >= l.total && (uintptr_t)workspace % 256 == 0, GNNOPS_EWORKSPACE,
                   "scatter1d_minmax: workspace %zu < %zu (or not 256-B aligned)", workspace_bytes, l.total);
    hipStream_t stream = (hipStream_t)s;
    switch (dtype) {
        case GNNOPS_F32: return run1d<float>(src, index, out, arg_out, E, N, reduce, workspace, stream, sortengine::pass_first_dstval_f32);
        case GNNOPS_F16: return run1d<__half>(src, index, out, arg_out, E, N, reduce, workspace, stream, sortengine::pass_first_dstval_f16);
        case GNNOPS_BF16:
            return run1d<__hip_bfloat16>(src, index, out, arg_out, E, N, reduce, workspace, stream, sortengine::pass_first_dstval_bf16);
    }
    gnnops_set_error("scatter1d_minmax: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}

// out[n] = sum / mean / product of src[e] over index[e] == n, in source position order (0 — 1 for a product — where nothing
// arrives): bit-identical to the sequential loop. Same shape limits and workspace as gnnops_scatter1d_minmax.
extern "C" int gnnops_scatter1d_sum(const void* src, const int64_t* index, void* out, int64_t E, int64_t N, int dtype, int reduce,
                                    void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    GNNOPS_REQUIRE(E >= 0 && N >= 0, GNNOPS_EINVAL, "scatter1d_sum: negative size");
    GNNOPS_REQUIRE(reduce == GNNOPS_SUM || reduce == GNNOPS_MEAN || reduce == GNNOPS_MUL, GNNOPS_EUNSUPPORTED,
                   "scatter1d_sum: reduce %d (sum / mean / mul only)", reduce);
    GNNOPS_REQUIRE(N > BUCKET && N < ((int64_t)1 << 31) - BUCKET && E > 0 && E < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED,
                   "scatter1d_sum: shape outside this form (E=%lld N=%lld)", (long long)E, (long long)N);
    GNNOPS_REQUIRE(src && index && out, GNNOPS_EINVAL, "scatter1d_sum: null pointer");
    const Layout1d l = layout1d(E, N);
    GNNOPS_REQUIRE(workspace && workspace_bytes >= l.total && (uintptr_t)workspace % 256 == 0, GNNOPS_EWORKSPACE,
                   "scatter1d_sum: workspace %zu < %zu (or not 256-B aligned)", workspace_bytes, l.total);
    hipStream_t stream = (hipStream_t)s;
    switch (dtype) {
        case GNNOPS_F32: return run1d_sum<float>(src, index, out, E, N, reduce, workspace, stream, sortengine::pass_first_dstval_f32);
        case GNNOPS_F16: return run1d_sum<__half>(src, index, out, E, N, reduce, workspace, stream, sortengine::pass_first_dstval_f16);
        case GNNOPS_BF16:
            return run1d_sum<__hip_bfloat16>(src, index, out, E, N, reduce, workspace, stream, sortengine::pass_first_dstval_bf16);
    }
    gnnops_set_error("scatter1d_sum: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}
