// hub.h — heavy destinations ("hubs") of the row reductions (segment.hip: plan form, bucket.hip: one-shot form).
//
// Everywhere else ONE lane group walks a destination's contributions in source order — that is what makes the result
// bit-identical to the sequential oracle — which serialises a destination with 10^5 contributions into 10^4 dependent
// batches of 8 rows (measured: +13 ms for one hub of 100 000 edges, +120 ms for 1 000 000; tools/time_hub.py). A uniform
// graph (the reference's generator) has no such destination; a power-law graph does. So:
//
//   * a destination with more than T_HUB contributions is set aside by the main kernel (one record per hub, appended
//     through an atomic counter; the main kernel neither reduces nor stores it);
//   * hub_partial_kernel: the hub's contributions are cut into pieces of PART, a workgroup per piece; inside a piece the
//     lane groups take contiguous parts and their partial results are combined IN ORDER;
//   * hub_combine_kernel: one lane group per hub folds the piece partials, again in order, into the output row (starting
//     from the row already there for index_add_ / `out=`), divides a mean, writes min / max positions.
//
// min / max (and their arg: smallest position) stay exact. Sums, means and products of a hub are re-associated — piece by
// piece instead of one by one — so they are deterministic but no longer bit-identical to the sequential loop: the price
// of not serialising. Everything below T_HUB is untouched.
//
// Bucket form: the partition groups a bucket's entries in source order but does not separate its 256 destinations, so a
// piece is PART entries of the BUCKET, filtered by destination on chip (stable compaction), and at most MAX_PER_BUCKET
// hubs of one bucket (those with the smallest ids) are set aside — the others take the sequential path.
#pragma once
#include "common.h"

namespace hub {

constexpr int T_HUB = 8192;
constexpr int PART = 4096;
constexpr int MAX_PER_BUCKET = 4;
constexpr int THREADS = 256;
constexpr int U = 8;  // rows in flight per lane group

struct Ws {
    int32_t* counters;    // [0] hubs, [1] pieces
    int32_t* hubs;        // [cap_h][4]: destination, beg, end, degree (beg / end: perm positions, or the bucket's entry range)
    int32_t* piece_base;  // [cap_h]: first piece of the hub
    int32_t* pieces;      // [cap_p][2]: hub, piece number inside the hub
    float* partial;       // [cap_p][K]
    int32_t* parg;        // [cap_p][K] (min / max) or nullptr
    int32_t cap_h, cap_p;
};

// Every hub has more than T_HUB contributions, so there are at most E / T_HUB of them. Plan form: a hub of length L has
// ceil(L / PART) pieces, in total at most E / PART + hubs. Bucket form: at most MAX_PER_BUCKET hubs per bucket, each with
// ceil(bucket length / PART) pieces, over buckets that hold more than T_HUB entries: at most
// MAX_PER_BUCKET * (E / PART + E / T_HUB). The capacities below cover both.
inline size_t cap_hubs(int64_t E) { return (size_t)(E / T_HUB) + 1; }
inline size_t cap_pieces(int64_t E) { return (size_t)MAX_PER_BUCKET * ((size_t)(E / PART) + cap_hubs(E)) + 1; }

struct Layout { size_t counters, hubs, piece_base, pieces, partial, parg, total; };
inline Layout layout(int64_t E, int64_t K, bool want_arg) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    Layout l{};
    size_t o = 0;
    l.counters = o; o += 256;
    l.hubs = o; o += up(cap_hubs(E) * 16);
    l.piece_base = o; o += up(cap_hubs(E) * 4);
    l.pieces = o; o += up(cap_pieces(E) * 8);
    l.partial = o; o += up(cap_pieces(E) * (size_t)K * 4);
    l.parg = o; o += want_arg ? up(cap_pieces(E) * (size_t)K * 4) : 0;
    l.total = o;
    return l;
}
inline Ws make_ws(void* base, const Layout& l, int64_t E, bool want_arg) {
    char* w = (char*)base;
    Ws s{};
    s.counters = (int32_t*)(w + l.counters);
    s.hubs = (int32_t*)(w + l.hubs);
    s.piece_base = (int32_t*)(w + l.piece_base);
    s.pieces = (int32_t*)(w + l.pieces);
    s.partial = (float*)(w + l.partial);
    s.parg = want_arg ? (int32_t*)(w + l.parg) : nullptr;
    s.cap_h = (int32_t)cap_hubs(E);
    s.cap_p = (int32_t)cap_pieces(E);
    return s;
}

// One thread: set a hub aside. The capacities cannot be exceeded (see above); should they be, the record is dropped and
// the counters saturate at the capacity in the readers (the hub would be lost, which the bound rules out).
__device__ inline void append(const Ws& w, int dst, int beg, int end, int deg) {
    const int np = (end - beg + PART - 1) / PART;
    const int h = atomicAdd(&w.counters[0], 1);
    if (h >= w.cap_h) return;
    const int pb = atomicAdd(&w.counters[1], np);
    w.hubs[4 * h + 0] = (pb + np <= w.cap_p) ? dst : -1;
    w.hubs[4 * h + 1] = beg;
    w.hubs[4 * h + 2] = end;
    w.hubs[4 * h + 3] = deg;
    w.piece_base[h] = pb;
    for (int p = 0; p < np && pb + p < w.cap_p; ++p) {
        w.pieces[2 * (pb + p) + 0] = (pb + np <= w.cap_p) ? h : -1;
        w.pieces[2 * (pb + p) + 1] = p;
    }
}

// A workgroup per piece: (1) the piece's source positions into LDS, in order — perm[beg + ...] (plan form) or the
// positions of the bucket entries that belong to the hub (bucket form, stable compaction); (2) the lane groups reduce
// contiguous parts of that list; (3) their partials are combined in group order into partial[piece].
template <typename T, int R, bool BUCKET>
__global__ __launch_bounds__(THREADS) void hub_partial_kernel(const T* __restrict__ src, const int32_t* __restrict__ perm,
                                                              const uint32_t* __restrict__ keys,
                                                              const uint32_t* __restrict__ vals, Ws w, int64_t E, int64_t K,
                                                              int gshift, int kchunks) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr bool IS_ARG = (R == GNNOPS_MIN || R == GNNOPS_MAX);
    __shared__ int32_t s_match[PART];
    __shared__ uint32_t s_tmp[THREADS / 64];
    __shared__ float s_part[THREADS * VEC];
    __shared__ int32_t s_parg[IS_ARG ? THREADS * VEC : 1];
    const int tid = threadIdx.x;
    const int G = 1 << gshift, gl = tid & (G - 1), gi = tid >> gshift, groups = THREADS >> gshift;
    int npieces = w.counters[1];
    if (npieces > w.cap_p) npieces = w.cap_p;
    for (int q = blockIdx.x; q < npieces; q += gridDim.x) {
        const int h = w.pieces[2 * q];
        if (h < 0) continue;
        const int pno = w.pieces[2 * q + 1];
        const int dst = w.hubs[4 * h], beg = w.hubs[4 * h + 1], end = w.hubs[4 * h + 2];
        const int pb = beg + pno * PART;
        const int n = (end - pb < PART) ? end - pb : PART;
        __syncthreads();  // the previous piece's readers are done with the LDS arrays
        int m;
        if constexpr (!BUCKET) {
            for (int i = tid; i < n; i += THREADS) s_match[i] = perm[pb + i];
            m = n;
        } else {
            constexpr int IPT = PART / THREADS;  // consecutive entries per thread: keeps the compaction stable
            const uint32_t low = (uint32_t)dst & 255u;
            uint32_t pos[IPT];
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const int i = tid * IPT + j;
                pos[j] = 0xffffffffu;
                if (i < n && (keys[pb + i] & 255u) == low) { pos[j] = vals[pb + i]; ++c; }
            }
            uint32_t tot;
            uint32_t off = block_excl_scan_u32<THREADS / 64>(c, s_tmp, &tot);
#pragma unroll
            for (int j = 0; j < IPT; ++j)
                if (pos[j] != 0xffffffffu) s_match[off++] = (int32_t)pos[j];
            m = (int)tot;
        }
        __syncthreads();
        const int part = (m + groups - 1) / groups;
        const int jb = gi * part, je = (jb + part < m) ? jb + part : m;
        for (int chunk = 0; chunk < kchunks; ++chunk) {
            const int64_t col = ((int64_t)chunk * G + gl) * VEC;
            float acc[VEC];
            int32_t arg[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) { acc[v] = Red<R>::identity(); arg[v] = (int32_t)E; }
            if (col < K) {
                const T* srcb = src + col;
                for (int j = jb; j < je; j += U) {
                    int32_t e[U];
                    u32x4 rows[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) e[u] = (j + u < je) ? s_match[j + u] : -1;
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (e[u] >= 0) rows[u] = load16<true>(srcb + (int64_t)e[u] * K);
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
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                s_part[tid * VEC + v] = acc[v];
                if constexpr (IS_ARG) s_parg[tid * VEC + v] = arg[v];
            }
            __syncthreads();
            if (gi == 0 && col < K) {  // fold the groups' partials in group (= source) order
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float a = s_part[gl * VEC + v];
                    int32_t ar = IS_ARG ? s_parg[gl * VEC + v] : 0;
                    for (int g = 1; g < groups; ++g) {
                        const float f = s_part[(g * G + gl) * VEC + v];
                        if constexpr (IS_ARG) {
                            if (Red<R>::better(f, a)) { a = f; ar = s_parg[(g * G + gl) * VEC + v]; }
                        } else {
                            a = Red<R>::apply(a, f);
                        }
                    }
                    w.partial[(int64_t)q * K + col + v] = a;
                    if constexpr (IS_ARG) {
                        if (w.parg) w.parg[(int64_t)q * K + col + v] = ar;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// One lane group per (hub, column chunk): the piece partials folded in order into the output row.
template <typename T, int R>
__global__ __launch_bounds__(THREADS) void hub_combine_kernel(T* __restrict__ out, int64_t* __restrict__ arg_out, Ws w,
                                                              int64_t E, int64_t K, int gshift, int kchunks,
                                                              int init_from_out, int is_mean) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr bool IS_ARG = (R == GNNOPS_MIN || R == GNNOPS_MAX);
    const int G = 1 << gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> gshift;
    const int gl = (int)(gtid & (G - 1));
    int nh = w.counters[0];
    if (nh > w.cap_h) nh = w.cap_h;
    for (int64_t item = gtid >> gshift; item < (int64_t)nh * kchunks; item += ngroups) {
        const int h = (int)(item / kchunks);
        const int chunk = (int)(item - (int64_t)h * kchunks);
        const int dst = w.hubs[4 * h];
        const int64_t col = ((int64_t)chunk * G + gl) * VEC;
        if (dst < 0 || col >= K) continue;
        const int np = (w.hubs[4 * h + 2] - w.hubs[4 * h + 1] + PART - 1) / PART;
        const int pb = w.piece_base[h];
        const int64_t oidx = (int64_t)dst * K + col;
        float acc[VEC];
        int32_t arg[VEC];
        if (init_from_out) {
            Elem<T>::unpack(*reinterpret_cast<const u32x4*>(out + oidx), acc);
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = Red<R>::identity();
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) arg[v] = (int32_t)E;
        for (int p = 0; p < np; ++p) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float f = w.partial[(int64_t)(pb + p) * K + col + v];
                if constexpr (IS_ARG) {
                    if (Red<R>::better(f, acc[v])) { acc[v] = f; arg[v] = w.parg[(int64_t)(pb + p) * K + col + v]; }
                } else {
                    acc[v] = Red<R>::apply(acc[v], f);
                }
            }
        }
        if constexpr (IS_ARG) {
            if (arg_out) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) arg_out[oidx + v] = arg[v];
            }
        } else if (R == GNNOPS_SUM) {
            if (is_mean) {
                const float c = (float)w.hubs[4 * h + 3];
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = acc[v] / c;
            }
        }
        store16<true>(out + oidx, Elem<T>::pack(acc));
    }
}

// SpMM (spmm.hip): an output row with more than T_HUB nonzeros. Same pieces; a contribution is value[e] * mat[col[e], :]
// (two roundings, as in spmm_rows_kernel), the piece partials are folded by hub_combine_kernel<T, GNNOPS_SUM>.
template <typename T>
__global__ __launch_bounds__(THREADS) void hub_partial_spmm_kernel(const int32_t* __restrict__ perm,
                                                                   const int64_t* __restrict__ col, const T* __restrict__ value,
                                                                   const T* __restrict__ mat, Ws w, int64_t D, int gshift,
                                                                   int kchunks) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ int32_t s_match[PART];
    __shared__ float s_part[THREADS * VEC];
    const int tid = threadIdx.x;
    const int G = 1 << gshift, gl = tid & (G - 1), gi = tid >> gshift, groups = THREADS >> gshift;
    int npieces = w.counters[1];
    if (npieces > w.cap_p) npieces = w.cap_p;
    for (int q = blockIdx.x; q < npieces; q += gridDim.x) {
        const int h = w.pieces[2 * q];
        if (h < 0) continue;
        const int pno = w.pieces[2 * q + 1];
        const int beg = w.hubs[4 * h + 1], end = w.hubs[4 * h + 2];
        const int pb = beg + pno * PART;
        const int n = (end - pb < PART) ? end - pb : PART;
        __syncthreads();
        for (int i = tid; i < n; i += THREADS) s_match[i] = perm ? perm[pb + i] : pb + i;
        __syncthreads();
        const int part = (n + groups - 1) / groups;
        const int jb = gi * part, je = (jb + part < n) ? jb + part : n;
        for (int chunk = 0; chunk < kchunks; ++chunk) {
            const int64_t c0 = ((int64_t)chunk * G + gl) * VEC;
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
            if (c0 < D) {
                for (int j = jb; j < je; j += U) {
                    int64_t c[U];
                    float wt[U];
                    u32x4 rows[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        c[u] = -1;
                        if (j + u < je) {
                            const int32_t e = s_match[j + u];
                            c[u] = col[e];
                            wt[u] = value ? Elem<T>::load(value + e) : 1.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (c[u] >= 0) rows[u] = load16<false>(mat + c[u] * D + c0);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (c[u] >= 0) {
                            float f[VEC];
                            Elem<T>::unpack(rows[u], f);
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[v] = __fadd_rn(acc[v], __fmul_rn(wt[u], f[v]));
                        }
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) s_part[tid * VEC + v] = acc[v];
            __syncthreads();
            if (gi == 0 && c0 < D) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float a = s_part[gl * VEC + v];
                    for (int g = 1; g < groups; ++g) a = __fadd_rn(a, s_part[(g * G + gl) * VEC + v]);
                    w.partial[(int64_t)q * D + c0 + v] = a;
                }
            }
            __syncthreads();
        }
    }
}

template <typename T>
inline void launch_spmm_pass(const int32_t* perm, const int64_t* col, const T* value, const T* mat, T* out, const Ws& w,
                             int64_t nnz, int64_t D, int gshift, int kchunks, hipStream_t stream) {
    const int ga = w.cap_p < 2048 ? w.cap_p : 2048;
    hipLaunchKernelGGL((hub_partial_spmm_kernel<T>), dim3(ga), dim3(THREADS), 0, stream, perm, col, value, mat, w, D, gshift,
                       kchunks);
    const int64_t items = (int64_t)w.cap_h * kchunks;
    const int gb = gnnops_grid_cap(gnnops_cdiv(items, THREADS >> gshift), 1024);
    hipLaunchKernelGGL((hub_combine_kernel<T, GNNOPS_SUM>), dim3(gb), dim3(THREADS), 0, stream, out, (int64_t*)nullptr, w, nnz,
                       D, gshift, kchunks, 0, 0);
}

// Push-form index_select (gather.hip / bucket.hip): a HOT table row — selected by more than T_HUB outputs — is stored
// by one lane group, eight stores in flight, for all of its outputs (56 ms for a row selected 10^6 times, against 1.3 ms
// for the whole op without it). Set aside in the same way, its outputs are written by a workgroup per piece instead:
// every lane group loads the row once and stores it to a contiguous part of the piece's output positions.
template <bool BUCKET>
__global__ __launch_bounds__(THREADS) void hub_push_kernel(const char* __restrict__ in, const int32_t* __restrict__ perm,
                                                           const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                           char* __restrict__ out, Ws w, int64_t rowbytes, int gshift,
                                                           int chunks) {
    __shared__ int32_t s_match[PART];
    __shared__ uint32_t s_tmp[THREADS / 64];
    const int tid = threadIdx.x;
    const int G = 1 << gshift, gl = tid & (G - 1), gi = tid >> gshift, groups = THREADS >> gshift;
    int npieces = w.counters[1];
    if (npieces > w.cap_p) npieces = w.cap_p;
    for (int q = blockIdx.x; q < npieces; q += gridDim.x) {
        const int h = w.pieces[2 * q];
        if (h < 0) continue;
        const int pno = w.pieces[2 * q + 1];
        const int dst = w.hubs[4 * h], beg = w.hubs[4 * h + 1], end = w.hubs[4 * h + 2];
        const int pb = beg + pno * PART;
        const int n = (end - pb < PART) ? end - pb : PART;
        __syncthreads();
        int m;
        if constexpr (!BUCKET) {
            for (int i = tid; i < n; i += THREADS) s_match[i] = perm[pb + i];
            m = n;
        } else {
            constexpr int IPT = PART / THREADS;
            const uint32_t low = (uint32_t)dst & 255u;
            uint32_t pos[IPT];
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const int i = tid * IPT + j;
                pos[j] = 0xffffffffu;
                if (i < n && (keys[pb + i] & 255u) == low) { pos[j] = vals[pb + i]; ++c; }
            }
            uint32_t tot;
            uint32_t off = block_excl_scan_u32<THREADS / 64>(c, s_tmp, &tot);
#pragma unroll
            for (int j = 0; j < IPT; ++j)
                if (pos[j] != 0xffffffffu) s_match[off++] = (int32_t)pos[j];
            m = (int)tot;
        }
        __syncthreads();
        for (int c = 0; c < chunks; ++c) {
            const int64_t colb = ((int64_t)c * G + gl) * 16;
            if (colb >= rowbytes) continue;
            const u32x4 v = load16<false>(in + (int64_t)dst * rowbytes + colb);
            char* outb = out + colb;
            for (int j0 = gi; j0 < m; j0 += groups * U) {   // groups interleave: neighbouring positions, neighbouring groups
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * groups;
                    if (j < m) store16<true>(outb + (int64_t)s_match[j] * rowbytes, v);
                }
            }
        }
    }
}

inline void launch_push_pass(bool bucket, const char* in, const int32_t* perm, const uint32_t* keys, const uint32_t* vals,
                             char* out, const Ws& w, int64_t rowbytes, int gshift, int chunks, hipStream_t stream) {
    const int ga = w.cap_p < 2048 ? w.cap_p : 2048;
    if (bucket)
        hipLaunchKernelGGL((hub_push_kernel<true>), dim3(ga), dim3(THREADS), 0, stream, in, perm, keys, vals, out, w, rowbytes,
                           gshift, chunks);
    else
        hipLaunchKernelGGL((hub_push_kernel<false>), dim3(ga), dim3(THREADS), 0, stream, in, perm, keys, vals, out, w, rowbytes,
                           gshift, chunks);
}

// The two launches that follow a main kernel which set hubs aside (they exit at once when there are none).
template <typename T, int R, bool BUCKET>
inline void launch_pass(const T* src, const int32_t* perm, const uint32_t* keys, const uint32_t* vals, T* out,
                        int64_t* arg_out, const Ws& w, int64_t E, int64_t K, int gshift, int kchunks, int init_from_out,
                        int is_mean, hipStream_t stream) {
    const int ga = w.cap_p < 2048 ? w.cap_p : 2048;
    hipLaunchKernelGGL((hub_partial_kernel<T, R, BUCKET>), dim3(ga), dim3(THREADS), 0, stream, src, perm, keys, vals, w, E, K,
                       gshift, kchunks);
    const int64_t items = (int64_t)w.cap_h * kchunks;
    const int gb = gnnops_grid_cap(gnnops_cdiv(items, THREADS >> gshift), 1024);
    hipLaunchKernelGGL((hub_combine_kernel<T, R>), dim3(gb), dim3(THREADS), 0, stream, out, arg_out, w, E, K, gshift, kchunks,
                       init_from_out, is_mean);
}

}  // namespace hub
