// bucket.hip — one-shot row scatter (torch_scatter.scatter_{add,mean,min,max,mul} with a row index, Tensor.index_add_;
// reference call sites as in segment.hip) when there is no plan to reuse: the LAST radix pass of the plan build, the
// row-pointer kernel and the perm / rowptr round trip through HBM are folded into the reduction.
//
//   1. global stable LSD passes over the destination bits ABOVE the low 8 (sort_engine): edges end up grouped by
//      bucket = destination >> 8 (256 consecutive output rows), in source order inside a bucket;
//   2. bucket_bounds_kernel: one binary search per bucket boundary (no pass over the keys);
//   3. bucket_reduce_kernel: a workgroup takes a bucket, finishes the sort ON CHIP — a stable counting sort of its
//      (destination & 255, source position) pairs in LDS, ballot ranking as in sort_engine_impl.h — and then runs the
//      segment reduction of segment.hip with the permutation and row pointers read from LDS instead of HBM.
//
// Same arithmetic and order as seg_rows_kernel (contributions of a destination in ascending source position), so the
// result is bit-identical to the plan path and to the sequential oracle. Against plan build + segment reduce at config 2
// this drops one scatter pass, its histogram/scan, the rowptr kernel and ~0.25 GB of perm/rowptr reads.
//
// A bucket larger than CAP entries (skewed destinations) is processed in chunks of CAP in source order: chunk c > 0
// starts from the output rows (and arg rows) chunk c-1 stored — the same sequence of operations when the output type
// holds the running value exactly, i.e. fp32 for every reduce and any type for min / max; the host side (ops.py) keeps
// 16-bit sums / means / products on the plan path, where the fp32 accumulator is rounded once.
#include "common.h"
#include "hub.h"
#include "sort_engine.h"

namespace {

constexpr int BSHIFT = 8, BROWS = 1 << BSHIFT;  // destinations per bucket
constexpr int THREADS = 256, WAVES = THREADS / 64;
constexpr int ROUNDS = 16, CAP = THREADS * ROUNDS;  // entries sorted on chip at a time
constexpr int U = 8;                                // contribution rows in flight per lane group
constexpr bool NT = true;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int key_bits(int64_t N) {
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < N) ++bits;
    return bits;
}

// bptr[b] = first position whose key >> BSHIFT >= b (keys are sorted by that quantity); bptr[NB] = E.
__global__ void bucket_bounds_kernel(const uint32_t* __restrict__ keys, int64_t E, int64_t NB, int32_t* __restrict__ bptr) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > NB) return;
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(keys[mid] >> BSHIFT) < b) lo = mid + 1; else hi = mid;
    }
    bptr[b] = (int32_t)lo;
}

// 16-B load that does not trust this CU's L1 (rows a previous chunk of the same workgroup stored).
template <typename T>
__device__ inline u32x4 load16_coherent(const T* p) {
    const volatile uint32_t* q = reinterpret_cast<const volatile uint32_t*>(p);
    return u32x4{q[0], q[1], q[2], q[3]};
}

// Stable counting sort, in LDS, of the n <= CAP (key & 255, position) pairs at [cbeg, cbeg + n): s_perm gets the positions
// grouped by key in their original order, s_rowptr[0..256] the group boundaries. Returns the size of group `threadIdx.x`.
// Ranking as in sort_engine_impl.h: a wave owns consecutive rows of 64 pairs; eight ballots give every lane the mask of
// its equal-key lanes; the lowest of them does ONE returning LDS add for the group. The caller must have passed a
// barrier since the last readers of the LDS arrays; s_perm / s_rowptr are valid after the caller's next barrier.
__device__ inline uint32_t sort_chunk(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int32_t cbeg, int n,
                                      int32_t* s_perm, uint32_t* s_whist, int32_t* s_rowptr, uint32_t* s_tmp) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint64_t lanes_below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t* whist = s_whist + wave * 256;
    for (int i = tid; i < WAVES * 256; i += THREADS) s_whist[i] = 0;
    __syncthreads();
    const int rounds_n = (n + THREADS - 1) / THREADS;   // rows of 64 per wave
    const int wave_base = wave * rounds_n * 64;
    uint32_t dg[ROUNDS], vv[ROUNDS], rk[ROUNDS];
    uint32_t is_leader = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        dg[r] = 0; vv[r] = 0; rk[r] = 0;
        if (r < rounds_n) {
            const int i = wave_base + r * 64 + lane;
            const bool valid = i < n;
            if (valid) {
                dg[r] = keys[cbeg + i] & (BROWS - 1);
                vv[r] = vals[cbeg + i];
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
    for (int r = 0; r < ROUNDS; ++r) {
        if (r < rounds_n) {
            const bool lead = (is_leader >> r) & 1u;
            const int from = lead ? lane : (int)((rk[r] >> 16) & 63u);
            const uint32_t p = __shfl(rk[r], from);
            rk[r] = lead ? p : p + (rk[r] & 0xffffu);
        }
    }
    __syncthreads();
    // key offsets: exclusive over waves, then over keys (thread d owns key d)
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = s_whist[w * 256 + tid];
        s_whist[w * 256 + tid] = tot;
        tot += c;
    }
    const uint32_t start = block_excl_scan_u32<WAVES>(tot, s_tmp, nullptr);
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s_whist[w * 256 + tid] += start;
    s_rowptr[tid] = (int32_t)start;
    if (tid == BROWS - 1) s_rowptr[BROWS] = (int32_t)(start + tot);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r < rounds_n) {
            const int i = wave_base + r * 64 + lane;
            if (i < n) s_perm[whist[dg[r]] + rk[r]] = (int32_t)vv[r];
        }
    }
    return tot;
}

// Hub detection for a large bucket (cold path, kept out of line so that its registers do not add to the row walk's):
// counts the bucket's contributions per destination, sets aside up to MAX_PER_BUCKET hubs (hub.h) and, if there are any,
// compacts the bucket's OTHER entries — a handful scattered among the hubs' — stably into the spare half of the
// partition's ping-pong buffers at the bucket's own offset. Returns the end of that compacted list, or -1 (no hub).
__device__ __noinline__ int32_t hub_prepare(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int32_t bbeg,
                                            int32_t bend, int64_t bucket, int64_t N, hub::Ws hw,
                                            uint32_t* __restrict__ spare_keys, uint32_t* __restrict__ spare_vals,
                                            uint32_t* s_deg, uint8_t* s_hub, uint32_t* s_tmp) {
    const int tid = threadIdx.x;
    s_deg[tid] = 0;
    __syncthreads();
    // counted per wave: eight ballots give every lane its equal-key lanes and the lowest of them adds the group size —
    // a hub's bucket is mostly ONE key, and 10^6 same-address LDS atomics would take milliseconds
    for (int32_t i0 = bbeg + (tid & ~63); i0 < bend; i0 += THREADS) {
        const int32_t i = i0 + (tid & 63);
        const bool valid = i < bend;
        const uint32_t d = valid ? (keys[i] & (BROWS - 1)) : 0u;
        const uint64_t m = match_digit8(d, __ballot(valid));
        const uint64_t below = (tid & 63) ? (m & (~0ull >> (64 - (tid & 63)))) : 0ull;
        if (valid && below == 0) atomicAdd(&s_deg[d], (uint32_t)__popcll(m));
    }
    __syncthreads();
    const uint32_t is_hub = s_deg[tid] > (uint32_t)hub::T_HUB ? 1u : 0u;
    const uint32_t rank = block_excl_scan_u32<WAVES>(is_hub, s_tmp, nullptr);  // the smallest ids first: deterministic
    const bool aside = is_hub && rank < (uint32_t)hub::MAX_PER_BUCKET && bucket * BROWS + tid < N;
    if (aside) {
        s_hub[tid] = 1;
        hub::append(hw, (int)(bucket * BROWS + tid), bbeg, bend, (int)s_deg[tid]);
    }
    if (__syncthreads_count(aside) == 0) return -1;
    constexpr int IPT = 8;
    uint32_t running = 0;
    for (int32_t w0 = bbeg; w0 < bend; w0 += THREADS * IPT) {
        uint32_t kk[IPT], vv2[IPT];
        uint32_t c = 0, keep = 0;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int32_t i = w0 + tid * IPT + j;
            kk[j] = 0; vv2[j] = 0;
            if (i < bend) {
                kk[j] = keys[i];
                vv2[j] = vals[i];
                if (!s_hub[kk[j] & (BROWS - 1)]) { keep |= 1u << j; ++c; }
            }
        }
        uint32_t tot;
        uint32_t off = bbeg + running + block_excl_scan_u32<WAVES>(c, s_tmp, &tot);
#pragma unroll
        for (int j = 0; j < IPT; ++j)
            if ((keep >> j) & 1u) { spare_keys[off] = kk[j]; spare_vals[off] = vv2[j]; ++off; }
        running += tot;
    }
    __threadfence();
    __syncthreads();
    return bbeg + (int32_t)running;
}

// (THREADS, 4): four waves per SIMD = 128 VGPRs. Sums need 121-125 anyway; min / max would take 129-137 and lose a wave
// of occupancy for one register (fp32: fits without a spill; 16-bit min / max: two spilled registers).
template <typename T, int R>
__global__ __launch_bounds__(THREADS, 4) void bucket_reduce_kernel(const T* __restrict__ src, const uint32_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ vals,
                                                                const int32_t* __restrict__ bptr, T* __restrict__ out,
                                                                int64_t* __restrict__ arg_out, int64_t E, int64_t K,
                                                                int64_t N, int64_t NB, int gshift, int kchunks,
                                                                int init_from_out, int is_mean, hub::Ws hw, int hub_on,
                                                                uint32_t* __restrict__ spare_keys,
                                                                uint32_t* __restrict__ spare_vals) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr bool IS_ARG = (R == GNNOPS_MIN || R == GNNOPS_MAX);
    __shared__ uint32_t s_deg[BROWS];  // hub detection: contributions per destination of a large bucket
    __shared__ uint8_t s_hub[BROWS];   // 1: set aside for the hub pass (hub.h)
    __shared__ int32_t s_perm[CAP];
    __shared__ uint32_t s_whist[WAVES * 256];
    __shared__ int32_t s_rowptr[BROWS + 1];
    __shared__ uint32_t s_cnt[BROWS];  // contributions per destination over all chunks of the bucket
    __shared__ uint32_t s_tmp[WAVES];

    const int tid = threadIdx.x;
    const int G = 1 << gshift;
    const int gl = tid & (G - 1);
    const int gi = tid >> gshift, groups = THREADS >> gshift;

    for (int64_t bucket = blockIdx.x; bucket < NB; bucket += gridDim.x) {
        const int32_t bbeg = bptr[bucket], bend = bptr[bucket + 1];
        const uint32_t* wkeys = keys;   // what the windows below walk: the bucket, or its non-hub entries compacted
        const uint32_t* wvals = vals;
        int32_t wend = bend;
        __syncthreads();  // the previous bucket's readers are done with s_hub
        s_hub[tid] = 0;
        if (hub_on && bend - bbeg > hub::T_HUB) {  // only such a bucket can hold a hub
            const int32_t ce = hub_prepare(keys, vals, bbeg, bend, bucket, N, hw, spare_keys, spare_vals, s_deg, s_hub, s_tmp);
            if (ce >= 0) {
                wkeys = spare_keys;
                wvals = spare_vals;
                wend = ce;
            }
        }
        for (int32_t cbeg = bbeg;; cbeg += CAP) {
            const bool first = cbeg == bbeg;
            const bool last = cbeg + CAP >= wend;
            const int n = (wend - cbeg < CAP) ? (wend - cbeg) : CAP;
            __syncthreads();  // previous chunk's / bucket's readers are done with s_perm, s_rowptr, s_cnt
            if (first) s_cnt[tid] = 0;

            const uint32_t tot = sort_chunk(wkeys, wvals, cbeg, n, s_perm, s_whist, s_rowptr, s_tmp);
            s_cnt[tid] += tot;
            __syncthreads();

            // ---- segment reduction of the bucket's rows (segment.hip's loop, indices from LDS)
            const bool from_out = first ? (init_from_out != 0) : true;
            for (int item = gi; item < BROWS * kchunks; item += groups) {
                const int dloc = item & (BROWS - 1);
                const int chunk = item >> BSHIFT;
                const int64_t nrow = bucket * BROWS + dloc;
                const int64_t col = ((int64_t)chunk * G + gl) * VEC;
                if (nrow >= N || col >= K || s_hub[dloc]) continue;  // hubs: neither reduced nor stored here
                const int32_t beg = s_rowptr[dloc], end = s_rowptr[dloc + 1];
                // nothing to fold in and the row already holds the running value: leave it (unless this visit must still
                // write an arg row, zero-fill an empty min / max group or divide a mean)
                if (beg == end && from_out && !(IS_ARG && (arg_out || (last && !init_from_out))) && !(is_mean && last)) continue;
                const T* srcb = src + col;
                const int64_t oidx = nrow * K + col;

                float acc[VEC];
                int32_t arg[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) arg[v] = (int32_t)E;
                if (from_out) {
                    u32x4 r = first ? *reinterpret_cast<const u32x4*>(out + oidx) : load16_coherent(out + oidx);
                    Elem<T>::unpack(r, acc);
                    if constexpr (IS_ARG) {
                        if (!first && arg_out) {
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                arg[v] = (int32_t)reinterpret_cast<const volatile int64_t*>(arg_out + oidx)[v];
                        }
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = Red<R>::identity();
                }

                for (int32_t j = beg; j < end; j += U) {
                    int32_t e[U];
                    u32x4 rows[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) e[u] = (j + u < end) ? s_perm[j + u] : -1;
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (e[u] >= 0) rows[u] = load16<NT>(srcb + (int64_t)e[u] * K);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (e[u] >= 0) {
                            float f[VEC];
                            Elem<T>::unpack(rows[u], f);
#pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                if constexpr (IS_ARG) {
                                    if (Red<R>::better(f[v], acc[v])) { acc[v] = f[v]; arg[v] = e[u]; }
                                } else {
                                    acc[v] = Red<R>::apply(acc[v], f[v]);
                                }
                            }
                        }
                    }
                }

                if constexpr (IS_ARG) {
                    // torch_scatter: groups nothing reached become 0 (decided once the whole bucket has been seen)
                    if (last && !init_from_out && s_cnt[dloc] == 0) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
                    }
                    if (arg_out) {
                        int64_t a[VEC];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) a[v] = arg[v];
                        u32x4* ap = reinterpret_cast<u32x4*>(arg_out + oidx);
                        const u32x4* as = reinterpret_cast<const u32x4*>(a);
#pragma unroll
                        for (int q = 0; q < VEC / 2; ++q) ap[q] = as[q];
                    }
                } else if (R == GNNOPS_SUM) {
                    if (is_mean && last) {
                        const uint32_t cnt = s_cnt[dloc];
                        const float c = (float)(cnt < 1 ? 1 : cnt);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[v] = acc[v] / c;
                    }
                }
                store16<NT>(out + oidx, Elem<T>::pack(acc));
            }
            if (last) break;
            __threadfence();  // the next chunk re-reads the rows just stored
        }
    }
}

// index_select, push form, from the same partition (of `index` over the table's rows): a workgroup takes a bucket of 256
// table rows, finishes the sort on chip, loads each selected table row once and stores it to every output row that
// selects it (select_rows_push_kernel of gather.hip with rowptr / perm in LDS). Rows are opaque 16-B lanes.
template <int PU>
__global__ __launch_bounds__(THREADS) void bucket_push_kernel(const char* __restrict__ in, const uint32_t* __restrict__ keys,
                                                              const uint32_t* __restrict__ vals,
                                                              const int32_t* __restrict__ bptr, char* __restrict__ out,
                                                              int64_t N, int64_t NB, int64_t rowbytes, int gshift,
                                                              int chunks, hub::Ws hw, int hub_on,
                                                              uint32_t* __restrict__ spare_keys,
                                                              uint32_t* __restrict__ spare_vals) {
    __shared__ uint32_t s_deg[BROWS];
    __shared__ uint8_t s_hub[BROWS];   // 1: a hot row, its outputs are written by the hub pass (hub.h)
    __shared__ int32_t s_perm[CAP];
    __shared__ uint32_t s_whist[WAVES * 256];
    __shared__ int32_t s_rowptr[BROWS + 1];
    __shared__ uint32_t s_tmp[WAVES];
    const int tid = threadIdx.x;
    const int G = 1 << gshift;
    const int gl = tid & (G - 1);
    const int gi = tid >> gshift, groups = THREADS >> gshift;
    for (int64_t bucket = blockIdx.x; bucket < NB; bucket += gridDim.x) {
        const int32_t bbeg = bptr[bucket], bend = bptr[bucket + 1];
        const uint32_t* wkeys = keys;
        const uint32_t* wvals = vals;
        int32_t wend = bend;
        __syncthreads();
        s_hub[tid] = 0;
        if (hub_on && bend - bbeg > hub::T_HUB) {
            const int32_t ce = hub_prepare(keys, vals, bbeg, bend, bucket, N, hw, spare_keys, spare_vals, s_deg, s_hub, s_tmp);
            if (ce >= 0) {
                wkeys = spare_keys;
                wvals = spare_vals;
                wend = ce;
            }
        }
        for (int32_t cbeg = bbeg; cbeg < wend; cbeg += CAP) {   // an unselected bucket stores nothing
            const int n = (wend - cbeg < CAP) ? (wend - cbeg) : CAP;
            __syncthreads();
            sort_chunk(wkeys, wvals, cbeg, n, s_perm, s_whist, s_rowptr, s_tmp);
            __syncthreads();
            for (int item = gi; item < BROWS * chunks; item += groups) {
                const int dloc = item & (BROWS - 1);
                const int c = item >> BSHIFT;
                const int64_t nrow = bucket * BROWS + dloc;
                const int64_t colb = ((int64_t)c * G + gl) * 16;
                if (nrow >= N || colb >= rowbytes || s_hub[dloc]) continue;
                const int32_t beg = s_rowptr[dloc], end = s_rowptr[dloc + 1];
                if (beg == end) continue;
                const u32x4 v = load16<true>(in + nrow * rowbytes + colb);
                char* outb = out + colb;
                for (int32_t j = beg; j < end; j += PU) {
                    int32_t e[PU];
#pragma unroll
                    for (int u = 0; u < PU; ++u) e[u] = (j + u < end) ? s_perm[j + u] : -1;
#pragma unroll
                    for (int u = 0; u < PU; ++u)
                        if (e[u] >= 0) store16<true>(outb + (int64_t)e[u] * rowbytes, v);
                }
            }
        }
    }
}

template <typename T, int R>
int launch_bucket(const void* src, const uint32_t* keys, const uint32_t* vals, const int32_t* bptr, void* out,
                  int64_t* arg_out, int64_t E, int64_t K, int64_t N, int64_t NB, int init_from_out, int is_mean,
                  hipStream_t stream, void* hub_ws, size_t hub_ws_bytes, uint32_t* spare_keys, uint32_t* spare_vals) {
    constexpr int VEC = Elem<T>::VEC;
    const int64_t vecs = K / VEC;
    int gshift = 0;
    while ((1 << gshift) < vecs && gshift < 6) ++gshift;
    const int kchunks = (int)gnnops_cdiv(vecs, (int64_t)1 << gshift);
    const int grid = gnnops_grid_cap(NB, 256 * 16);
    constexpr bool want_arg = (R == GNNOPS_MIN || R == GNNOPS_MAX);
    hub::Ws hw{};
    int hub_on = 0;
    if (hub_ws && E > hub::T_HUB) {
        const hub::Layout hl = hub::layout(E, K, want_arg);
        if (hub_ws_bytes >= hl.total) {
            hw = hub::make_ws(hub_ws, hl, E, want_arg);
            if (gnnops_memset_async(hw.counters, 0, 8, stream) != hipSuccess) return gnnops_check_launch("hub memset");
            hub_on = 1;
        }
    }
    hipLaunchKernelGGL((bucket_reduce_kernel<T, R>), dim3(grid), dim3(THREADS), 0, stream, (const T*)src, keys, vals, bptr,
                       (T*)out, arg_out, E, K, N, NB, gshift, kchunks, init_from_out, is_mean, hw, hub_on, spare_keys,
                       spare_vals);
    if (hub_on)
        hub::launch_pass<T, R, true>((const T*)src, nullptr, keys, vals, (T*)out, arg_out, hw, E, K, gshift, kchunks,
                                     init_from_out, is_mean, stream);
    return gnnops_check_launch("scatter_rows_oneshot");
}

template <typename T>
int dispatch_bucket(int reduce, const void* src, const uint32_t* keys, const uint32_t* vals, const int32_t* bptr, void* out,
                    int64_t* arg_out, int64_t E, int64_t K, int64_t N, int64_t NB, int init_from_out, hipStream_t stream,
                    void* hw = nullptr, size_t hb = 0, uint32_t* sk = nullptr, uint32_t* sv = nullptr) {
    switch (reduce) {
        case GNNOPS_SUM: return launch_bucket<T, GNNOPS_SUM>(src, keys, vals, bptr, out, nullptr, E, K, N, NB, init_from_out, 0, stream, hw, hb, sk, sv);
        case GNNOPS_MEAN: return launch_bucket<T, GNNOPS_SUM>(src, keys, vals, bptr, out, nullptr, E, K, N, NB, init_from_out, 1, stream, hw, hb, sk, sv);
        case GNNOPS_MUL: return launch_bucket<T, GNNOPS_MUL>(src, keys, vals, bptr, out, nullptr, E, K, N, NB, init_from_out, 0, stream, hw, hb, sk, sv);
        case GNNOPS_MIN: return launch_bucket<T, GNNOPS_MIN>(src, keys, vals, bptr, out, arg_out, E, K, N, NB, init_from_out, 0, stream, hw, hb, sk, sv);
        case GNNOPS_MAX: return launch_bucket<T, GNNOPS_MAX>(src, keys, vals, bptr, out, arg_out, E, K, N, NB, init_from_out, 0, stream, hw, hb, sk, sv);
    }
    gnnops_set_error("scatter_rows_oneshot: unknown reduce %d", reduce);
    return GNNOPS_EINVAL;
}

struct Layout {
    size_t keys_a, keys_b, vals_a, vals_b, tile_hist, digit_total, bptr, total;
};
inline Layout layout(int64_t E, int64_t N) {
    Layout l{};
    const size_t tiles = (size_t)gnnops_cdiv(E > 0 ? E : 1, sortengine::TILE);
    size_t o = 0;
    l.keys_a = o; o += align_up((size_t)E * 4, 256);
    l.keys_b = o; o += align_up((size_t)E * 4, 256);
    l.vals_a = o; o += align_up((size_t)E * 4, 256);
    l.vals_b = o; o += align_up((size_t)E * 4, 256);
    l.tile_hist = o; o += align_up(256 * tiles * 4, 256);
    l.digit_total = o; o += 256 * 4;
    l.bptr = o; o += align_up(((size_t)gnnops_cdiv(N, BROWS) + 1) * 4, 256);
    l.total = o;
    return l;
}

// Keys are destinations in [0, N) plus one sentinel, the first multiple of 256 at or past N ("not mine": positions the
// windowed partition drops land in a bucket of their own behind the last real one). LSD passes cover bits [BSHIFT, bits)
// of the sentinel; their count also fixes which ping-pong buffer holds the partitioned (key, position) pairs.
inline int64_t sentinel_key(int64_t N) { return gnnops_cdiv(N, BROWS) * BROWS; }
inline int partition_passes(int64_t N) { return (key_bits(sentinel_key(N) + 1) - BSHIFT + 7) / 8; }

// key[e] = index[e] - lo if that lies in [0, N), else the sentinel.
__global__ void window_keys_kernel(const int64_t* __restrict__ index, uint32_t* __restrict__ keys, int64_t E, int64_t lo,
                                   int64_t N, uint32_t sentinel) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t d = (uint64_t)(index[e] - lo);
        keys[e] = d < (uint64_t)N ? (uint32_t)d : sentinel;
    }
}

inline bool oneshot_shape_ok(int64_t E, int64_t N) {
    return N > BROWS && E > 0 && E < ((int64_t)1 << 31) && N < ((int64_t)1 << 31);
}

}  // namespace

extern "C" size_t gnnops_bucket_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return 0;
    return layout(E, N).total;
}

namespace {
int partition_impl(const int64_t* index, int64_t E, int64_t lo, int64_t N, bool window, void* workspace,
                   size_t workspace_bytes, hipStream_t stream, const char* what) {
    GNNOPS_REQUIRE(E >= 0 && N >= 0, GNNOPS_EINVAL, "%s: negative size", what);
    GNNOPS_REQUIRE(oneshot_shape_ok(E, N), GNNOPS_EUNSUPPORTED, "%s: shape outside the bucketed form (E=%lld N=%lld)", what,
                   (long long)E, (long long)N);
    GNNOPS_REQUIRE(index, GNNOPS_EINVAL, "%s: null pointer", what);
    const Layout l = layout(E, N);
    GNNOPS_REQUIRE(workspace && workspace_bytes >= l.total, GNNOPS_EWORKSPACE, "%s: workspace %zu < %zu", what,
                   workspace_bytes, l.total);
    char* w = (char*)workspace;
    uint32_t* kbuf[2] = {(uint32_t*)(w + l.keys_a), (uint32_t*)(w + l.keys_b)};
    uint32_t* vbuf[2] = {(uint32_t*)(w + l.vals_a), (uint32_t*)(w + l.vals_b)};
    uint32_t* tile_hist = (uint32_t*)(w + l.tile_hist);
    uint32_t* digit_total = (uint32_t*)(w + l.digit_total);
    int32_t* bptr = (int32_t*)(w + l.bptr);
    const int tiles = (int)gnnops_cdiv(E, sortengine::TILE);
    const int passes = partition_passes(N);  // >= 1 because N > 256
    if (window) {  // keys_b is free until pass 1 writes it, and pass 0 is the only reader of the windowed keys
        const int grid = gnnops_grid_cap(gnnops_cdiv(E, 256), 256 * 16);
        hipLaunchKernelGGL(window_keys_kernel, dim3(grid), dim3(256), 0, stream, index, kbuf[1], E, lo, N,
                           (uint32_t)sentinel_key(N));
    }
    const uint32_t* kin = nullptr;
    const uint32_t* vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        uint32_t* kout = kbuf[p & 1];
        uint32_t* vout = vbuf[p & 1];
        int rc;
        if (p == 0)
            rc = window ? sortengine::pass_first_u32(kbuf[1], kout, vout, E, BSHIFT, tile_hist, digit_total, tiles, stream)
                        : sortengine::pass_first_i64(index, kout, vout, E, BSHIFT, tile_hist, digit_total, tiles, stream);
        else
            rc = sortengine::pass_u32(kin, vin, kout, vout, E, BSHIFT + 8 * p, tile_hist, digit_total, tiles, stream);
        if (rc != GNNOPS_OK) return rc;
        kin = kout; vin = vout;
    }
    const int64_t NB = gnnops_cdiv(N, BROWS);
    hipLaunchKernelGGL(bucket_bounds_kernel, dim3((unsigned)gnnops_cdiv(NB + 1, 256)), dim3(256), 0, stream, kin, E, NB, bptr);
    return gnnops_check_launch(what);
}
}  // namespace

// Stage 1: group the E positions by bucket = index >> 8, in position order inside a bucket, and find the bucket
// boundaries. The workspace then IS the partition: gnnops_bucket_reduce / gnnops_bucket_select may be called on it any
// number of times. Every index[e] must lie in [0, N).
extern "C" int gnnops_bucket_partition(const int64_t* index, int64_t E, int64_t N, void* workspace, size_t workspace_bytes,
                                       gnnops_stream_t s) {
    return partition_impl(index, E, 0, N, false, workspace, workspace_bytes, (hipStream_t)s, "bucket_partition");
}

// Windowed stage 1 (destination-partitioned scatter, gnnops/dist.py): positions with index[e] in [lo, lo + N) are
// partitioned under the local id index[e] - lo; all others are set aside, in position order, behind the last bucket:
// they are vals[bptr[NB] .. E) of the layout below and no reduce / select touches them.
extern "C" int gnnops_bucket_partition_window(const int64_t* index, int64_t E, int64_t lo, int64_t N, void* workspace,
                                              size_t workspace_bytes, gnnops_stream_t s) {
    return partition_impl(index, E, lo, N, true, workspace, workspace_bytes, (hipStream_t)s, "bucket_partition_window");
}

// Byte offsets, inside a partitioned workspace, of the (key u32[E], position u32[E]) pairs and of bptr int32[NB + 1]
// (NB = ceil(N / 256); bucket b holds pairs bptr[b] .. bptr[b+1]).
extern "C" int gnnops_bucket_layout(int64_t E, int64_t N, size_t* keys_offset, size_t* vals_offset, size_t* bptr_offset) {
    GNNOPS_REQUIRE(E >= 0 && N >= 0 && keys_offset && vals_offset && bptr_offset, GNNOPS_EINVAL, "bucket_layout: bad argument");
    const Layout l = layout(E, N);
    const int last = (partition_passes(N) - 1) & 1;
    *keys_offset = last ? l.keys_b : l.keys_a;
    *vals_offset = last ? l.vals_b : l.vals_a;
    *bptr_offset = l.bptr;
    return GNNOPS_OK;
}

// Stage 2: out[n, :] = reduce over { src[e, :] : index[e] == n } in ascending e, from a workspace gnnops_bucket_partition
// filled for the same (E, N). src [E, K], out [N, K]; arg_out [N, K] int64 or NULL (min / max only).
extern "C" int gnnops_bucket_reduce(const void* src, const void* workspace, void* out, int64_t* arg_out, int64_t E, int64_t K,
                                    int64_t N, int dtype, int reduce, int init_from_out, gnnops_stream_t s) {
    return gnnops_bucket_reduce_hubs(src, workspace, out, arg_out, E, K, N, dtype, reduce, init_from_out, nullptr, 0, s);
}

// The same with hubs (more than 8192 contributions to one destination) set aside and reduced piecewise: hub.h.
extern "C" int gnnops_bucket_reduce_hubs(const void* src, const void* workspace, void* out, int64_t* arg_out, int64_t E,
                                         int64_t K, int64_t N, int dtype, int reduce, int init_from_out, void* hub_workspace,
                                         size_t hub_workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(E >= 0 && K >= 0 && N >= 0, GNNOPS_EINVAL, "bucket_reduce: negative size");
    GNNOPS_REQUIRE(!(reduce == GNNOPS_MEAN && init_from_out), GNNOPS_EINVAL, "bucket_reduce: mean cannot start from out");
    GNNOPS_REQUIRE(dtype == GNNOPS_F32 || dtype == GNNOPS_F16 || dtype == GNNOPS_BF16, GNNOPS_EINVAL,
                   "bucket_reduce: unknown dtype %d", dtype);
    const int vec = dtype == GNNOPS_F32 ? 4 : 8;
    const bool rows_ok = K > 0 && K % vec == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)out % 16 == 0 &&
                         (arg_out == nullptr || (uintptr_t)arg_out % 16 == 0);
    GNNOPS_REQUIRE(rows_ok && oneshot_shape_ok(E, N), GNNOPS_EUNSUPPORTED,
                   "bucket_reduce: shape outside the bucketed form (E=%lld K=%lld N=%lld)", (long long)E, (long long)K,
                   (long long)N);
    GNNOPS_REQUIRE(src && workspace && out, GNNOPS_EINVAL, "bucket_reduce: null pointer");
    const Layout l = layout(E, N);
    const char* w = (const char*)workspace;
    const int last = (partition_passes(N) - 1) & 1;
    const uint32_t* keys = (const uint32_t*)(w + (last ? l.keys_b : l.keys_a));
    const uint32_t* vals = (const uint32_t*)(w + (last ? l.vals_b : l.vals_a));
    const int32_t* bptr = (const int32_t*)(w + l.bptr);
    const int64_t NB = gnnops_cdiv(N, BROWS);
    // the other half of the ping-pong buffers is scratch once the partition is built: the hub path compacts into it
    uint32_t* sk = (uint32_t*)(const_cast<char*>(w) + (last ? l.keys_a : l.keys_b));
    uint32_t* sv = (uint32_t*)(const_cast<char*>(w) + (last ? l.vals_a : l.vals_b));
    switch (dtype) {
        case GNNOPS_F32: return dispatch_bucket<float>(reduce, src, keys, vals, bptr, out, arg_out, E, K, N, NB, init_from_out, stream, hub_workspace, hub_workspace_bytes, sk, sv);
        case GNNOPS_F16: return dispatch_bucket<__half>(reduce, src, keys, vals, bptr, out, arg_out, E, K, N, NB, init_from_out, stream, hub_workspace, hub_workspace_bytes, sk, sv);
        default: return dispatch_bucket<__hip_bfloat16>(reduce, src, keys, vals, bptr, out, arg_out, E, K, N, NB, init_from_out, stream, hub_workspace, hub_workspace_bytes, sk, sv);
    }
}

// index_select(input [N, K], index [E]) -> out [E, K] from a workspace gnnops_bucket_partition filled for (index, E, N):
// every selected input row is read once. Rows are K * elem_bytes bytes, a multiple of 16, 16-B aligned.
extern "C" int gnnops_bucket_select(const void* input, const void* workspace, void* out, int64_t N, int64_t K, int64_t E,
                                    int elem_bytes, gnnops_stream_t s) {
    return gnnops_bucket_select_hubs(input, workspace, out, N, K, E, elem_bytes, nullptr, 0, s);
}

// The same with hot rows (selected by more than 8192 outputs) set aside and written by whole workgroups (hub.h);
// hub_workspace: gnnops_hub_workspace_bytes(E, 0, 0) bytes, or NULL.
extern "C" int gnnops_bucket_select_hubs(const void* input, const void* workspace, void* out, int64_t N, int64_t K, int64_t E,
                                         int elem_bytes, void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(N >= 0 && K >= 0 && E >= 0, GNNOPS_EINVAL, "bucket_select: negative size");
    const int64_t rowbytes = K * elem_bytes;
    GNNOPS_REQUIRE(rowbytes > 0 && rowbytes % 16 == 0 && (uintptr_t)input % 16 == 0 && (uintptr_t)out % 16 == 0 &&
                       oneshot_shape_ok(E, N),
                   GNNOPS_EUNSUPPORTED, "bucket_select: shape outside the bucketed form (E=%lld K=%lld N=%lld)", (long long)E,
                   (long long)K, (long long)N);
    GNNOPS_REQUIRE(input && workspace && out, GNNOPS_EINVAL, "bucket_select: null pointer");
    const Layout l = layout(E, N);
    const char* w = (const char*)workspace;
    const int last = (partition_passes(N) - 1) & 1;
    const uint32_t* keys = (const uint32_t*)(w + (last ? l.keys_b : l.keys_a));
    const uint32_t* vals = (const uint32_t*)(w + (last ? l.vals_b : l.vals_a));
    const int32_t* bptr = (const int32_t*)(w + l.bptr);
    const int64_t NB = gnnops_cdiv(N, BROWS);
    const int64_t lanes = rowbytes / 16;
    int gshift = 0;
    while ((1 << gshift) < lanes && gshift < 6) ++gshift;
    const int chunks = (int)gnnops_cdiv(lanes, (int64_t)1 << gshift);
    const int grid = gnnops_grid_cap(NB, 256 * 16);
    hub::Ws hw{};
    int hub_on = 0;
    if (hub_workspace && E > hub::T_HUB) {
        const hub::Layout hl = hub::layout(E, 0, false);
        if (hub_workspace_bytes >= hl.total) {
            hw = hub::make_ws(hub_workspace, hl, E, false);
            if (gnnops_memset_async(hw.counters, 0, 8, stream) != hipSuccess) return gnnops_check_launch("hub memset");
            hub_on = 1;
        }
    }
    uint32_t* sk = (uint32_t*)(const_cast<char*>(w) + (last ? l.keys_a : l.keys_b));   // spare half of the ping-pong buffers
    uint32_t* sv = (uint32_t*)(const_cast<char*>(w) + (last ? l.vals_a : l.vals_b));
    hipLaunchKernelGGL((bucket_push_kernel<8>), dim3(grid), dim3(THREADS), 0, stream, (const char*)input, keys, vals, bptr,
                       (char*)out, N, NB, rowbytes, gshift, chunks, hw, hub_on, sk, sv);
    if (hub_on)
        hub::launch_push_pass(true, (const char*)input, nullptr, keys, vals, (char*)out, hw, rowbytes, gshift, chunks, stream);
    return gnnops_check_launch("bucket_select");
}

// Both stages in one call — what a scatter with no plan to reuse runs. GNNOPS_EUNSUPPORTED (take the plan path instead)
// when the rows are not 16-B lane rows (K % (16 / elem) != 0 or unaligned pointers), N <= 256, E == 0 or E / N >= 2^31.
extern "C" int gnnops_scatter_rows_oneshot(const void* src, const int64_t* index, void* out, int64_t* arg_out, int64_t E,
                                           int64_t K, int64_t N, int dtype, int reduce, int init_from_out,
                                           void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    GNNOPS_REQUIRE(E >= 0 && K >= 0 && N >= 0, GNNOPS_EINVAL, "scatter_rows_oneshot: negative size");
    const int vec = dtype == GNNOPS_F32 ? 4 : 8;
    const bool rows_ok = K > 0 && K % vec == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)out % 16 == 0 &&
                         (arg_out == nullptr || (uintptr_t)arg_out % 16 == 0);
    GNNOPS_REQUIRE(rows_ok && oneshot_shape_ok(E, N), GNNOPS_EUNSUPPORTED,
                   "scatter_rows_oneshot: shape outside the one-shot form (E=%lld K=%lld N=%lld)", (long long)E,
                   (long long)K, (long long)N);
    const int rc = gnnops_bucket_partition(index, E, N, workspace, workspace_bytes, s);
    if (rc != GNNOPS_OK) return rc;
    return gnnops_bucket_reduce(src, workspace, out, arg_out, E, K, N, dtype, reduce, init_from_out, s);
}
