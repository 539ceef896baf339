// segment.hip — per-destination reduction over a plan (rowptr, perm): the kernel behind
// torch_scatter.scatter_{add,mean,min,max,mul} with a row index, and Tensor.index_add_.
// Reference call sites: op_bm_scripts/benchmark_scatter_add.py:15-19, benchmark_scatter_mean.py:15-18,
// benchmark_scatter_min.py:15-18, benchmark_scatter_max.py:15-18, benchmark_native_index_add_.py:13-16.
//
// HBM-bound. Each destination row is produced by ONE lane group that walks its contributions in
// ascending source position (the plan is stable), so the fp32 result is the same sequence of adds a
// sequential CPU loop performs, and min/max ties resolve to the smallest position. Every source row is
// read once with 16-B lane accesses along the feature dimension; every output row is written once with
// plain stores (no atomics, no zero-fill pass).
//
// Algorithmic bytes per destination row: deg*K*s (src) + deg*8 (index) + K*s (out)
// [+ K*8 arg_out]. Extra real traffic: rowptr 4 B/row and perm 4 B/edge instead of the 8-B index.
#include "common.h"
#include <stdlib.h>
#include "hub.h"

namespace {

constexpr int U = 8;       // contribution rows in flight per lane group
// Source rows and output rows are touched exactly once: nontemporal accesses keep them from evicting
// rowptr / perm lines (measured -5 % kernel time at config 2; tools/time_seg.py).
constexpr bool NT = true;
constexpr int SEG_BLOCK = 256;   // consecutive items (output rows) a workgroup of seg_rows_kernel takes at a time

// Row form: the row is K elements, K % VEC == 0, 16-B aligned. A group of G = 2^gshift lanes owns one
// (b, n, chunk) item; chunk c covers elements [c*G*VEC, (c+1)*G*VEC).
template <typename T, int R>
__global__ __launch_bounds__(256) void seg_rows_kernel(const T* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ perm, T* __restrict__ out,
                                                       int64_t* __restrict__ arg_out, int64_t B, int64_t E, int64_t K,
                                                       int64_t N, int gshift, int kchunks, int init_from_out,
                                                       int is_mean, hub::Ws hw, int hub_on, int blk_map) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr bool IS_ARG = (R == GNNOPS_MIN || R == GNNOPS_MAX);
    const int G = 1 << gshift;
    const int gl = threadIdx.x & (G - 1);
    const int gi = threadIdx.x >> gshift, groups = 256 >> gshift;
    const int64_t items = B * (int64_t)kchunks * N;
    const int64_t nblocks = (items + SEG_BLOCK - 1) / SEG_BLOCK;

    // blk_map: a workgroup visits SEG_BLOCK CONSECUTIVE items (= consecutive output rows) at a time, its lane groups
    // interleaved over them: what it stores is one contiguous run (128 KiB for 512-B rows), written front to back. The other
    // map grid-strides the items (each workgroup round 4 KiB here, the next one a whole grid away) — the pattern that loses
    // 10-25 % in a plain copy / fill (tools/micro/store_sweep.hip); which one a launch takes: launch_seg below.
    const int64_t outer_n = blk_map ? nblocks : (int64_t)gridDim.x;   // grid-stride: every workgroup runs the outer loop once
    for (int64_t ob = blockIdx.x; ob < outer_n; ob += gridDim.x) {
    const int64_t iend = blk_map ? ((ob + 1) * SEG_BLOCK < items ? (ob + 1) * SEG_BLOCK : items) : items;
    const int64_t istep = blk_map ? groups : ((int64_t)gridDim.x * 256) >> gshift;
    for (int64_t item = blk_map ? ob * SEG_BLOCK + gi : ((int64_t)blockIdx.x * 256 + threadIdx.x) >> gshift; item < iend; item += istep) {
        const int64_t n = item % N;
        const int64_t bc = item / N;
        const int chunk = (int)(bc % kchunks);
        const int64_t b = bc / kchunks;
        const int64_t col = ((int64_t)chunk * G + gl) * VEC;
        if (col >= K) continue;
        const int32_t beg = rowptr[n], end = rowptr[n + 1];
        if (beg == end && init_from_out && !(IS_ARG && arg_out)) continue;  // nothing to fold in: the out row stays as it is
        if (hub_on && end - beg > hub::T_HUB) {  // a hub (hub.h): set aside for the hub pass, neither reduced nor stored here
            if (gl == 0 && chunk == 0) hub::append(hw, (int)n, beg, end, end - beg);
            continue;
        }
        const T* srcb = src + (b * E) * K + col;
        const int64_t oidx = (b * N + n) * K + col;

        float acc[VEC];
        int32_t arg[VEC];
        if (init_from_out) {
            u32x4 r = *reinterpret_cast<const u32x4*>(out + oidx);
            Elem<T>::unpack(r, acc);
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = Red<R>::identity();
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) arg[v] = (int32_t)E;

        for (int32_t j = beg; j < end; j += U) {
            int32_t e[U];
            u32x4 rows[U];
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = (j + u < end) ? (perm ? perm[j + u] : j + u) : -1;
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
            if (!init_from_out) {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (arg[v] == (int32_t)E) acc[v] = 0.f;  // torch_scatter: groups nothing reached become 0
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
            if (is_mean) {
                const int32_t cnt = end - beg;
                const float c = (float)(cnt < 1 ? 1 : cnt);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = acc[v] / c;
            }
        }
        store16<NT>(out + oidx, Elem<T>::pack(acc));
    }
    }
}

// Generic form: one thread per output element (b, n, k); any K, any alignment. Coalesced along k.
template <typename T, int R>
__global__ __launch_bounds__(256) void seg_elems_kernel(const T* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ perm, T* __restrict__ out,
                                                        int64_t* __restrict__ arg_out, int64_t B, int64_t E, int64_t K,
                                                        int64_t N, int init_from_out, int is_mean) {
    constexpr bool IS_ARG = (R == GNNOPS_MIN || R == GNNOPS_MAX);
    const int64_t total = B * N * K;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = o % K;
        const int64_t bn = o / K;
        const int64_t n = bn % N;
        const int64_t b = bn / N;
        const int32_t beg = rowptr[n], end = rowptr[n + 1];
        const T* srcb = src + (b * E) * K + k;
        float acc = init_from_out ? Elem<T>::load(out + o) : Red<R>::identity();
        int32_t arg = (int32_t)E;
        for (int32_t j = beg; j < end; ++j) {
            const int32_t e = perm ? perm[j] : j;
            const float f = Elem<T>::load(srcb + (int64_t)e * K);
            if constexpr (IS_ARG) {
                if (Red<R>::better(f, acc)) { acc = f; arg = e; }
            } else {
                acc = Red<R>::apply(acc, f);
            }
        }
        if constexpr (IS_ARG) {
            if (!init_from_out && arg == (int32_t)E) acc = 0.f;
            if (arg_out) arg_out[o] = arg;
        } else if (R == GNNOPS_SUM) {
            if (is_mean) {
                const int32_t cnt = end - beg;
                acc = acc / (float)(cnt < 1 ? 1 : cnt);
            }
        }
        Elem<T>::store(out + o, acc);
    }
}

// K == 1 with a batch in front (the reference's index_add_ case: (L, L) matrices, dim = 1, one index for every row —
// benchmark_native_index_add_.py:13-16,62): out[b, n] = reduce over e in segment n of src[b, e]. The generic element
// kernel gathers 2-byte values from HBM; here a workgroup parks TB whole rows src[b, :] in LDS with coalesced loads and
// every thread walks the segment of its destination once for all TB rows, reading the values from LDS. No atomics:
// contributions are combined in ascending position, as everywhere else.
constexpr int K1_THREADS = 512, K1_MAX_THREADS = 1024;
constexpr int K1_MAX_TB = 4;
constexpr int K1_U = 4;  // destinations per thread per sweep
constexpr size_t K1_LDS_BYTES = 156 * 1024;  // largest row taken
constexpr size_t K1_LDS_TARGET = 40 * 1024;  // LDS per workgroup aimed at

template <typename T, int R>
__global__ __launch_bounds__(K1_MAX_THREADS) void seg_k1_kernel(const T* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ perm, T* __restrict__ out,
                                                            int64_t* __restrict__ arg_out, int64_t B, int64_t E, int64_t N,
                                                            int TB, int init_from_out, int is_mean) {
    constexpr bool IS_ARG = (R == GNNOPS_MIN || R == GNNOPS_MAX);
    extern __shared__ __attribute__((aligned(16))) unsigned char k1_raw[];
    T* rows = reinterpret_cast<T*>(k1_raw);  // [TB][E]
    const int NT = (int)blockDim.x;  // 512, or 1024 for rows that leave room for one workgroup per CU only
    const int64_t b0 = (int64_t)blockIdx.x * TB;
    const int tb = (int)((B - b0 < TB) ? (B - b0) : TB);
    const T* sb = src + b0 * E;
    const int64_t nelem = (int64_t)tb * E;  // the tb rows are contiguous in src
    if ((((uintptr_t)sb) & 15) == 0) {
        constexpr int PER = 16 / (int)sizeof(T);
        const int64_t nvec = nelem / PER;
        const u32x4* sv = reinterpret_cast<const u32x4*>(sb);
        u32x4* dv = reinterpret_cast<u32x4*>(rows);
        int64_t i = threadIdx.x;
        for (; i + 3 * NT < nvec; i += 4 * NT) {  // four 16-B loads in flight per thread
            const u32x4 a = sv[i], b = sv[i + NT], c = sv[i + 2 * NT], d = sv[i + 3 * NT];
            dv[i] = a; dv[i + NT] = b; dv[i + 2 * NT] = c; dv[i + 3 * NT] = d;
        }
        for (; i < nvec; i += NT) dv[i] = sv[i];
        for (int64_t j = nvec * PER + threadIdx.x; j < nelem; j += NT) rows[j] = sb[j];
    } else {
        for (int64_t i = threadIdx.x; i < nelem; i += NT) rows[i] = sb[i];
    }
    __syncthreads();

    // K1_U destinations per thread at a time, so that the row-pointer, out and perm loads of all of them are in flight
    // together (loads are unconditional on clamped indices: a load under a branch is waited for inside the branch)
    for (int64_t n0 = threadIdx.x; n0 < N; n0 += (int64_t)NT * K1_U) {
        int32_t beg[K1_U], end[K1_U];
        int64_t nn[K1_U];
#pragma unroll
        for (int u = 0; u < K1_U; ++u) {
            const int64_t n = n0 + (int64_t)u * NT;
            nn[u] = n < N ? n : N - 1;
            beg[u] = rowptr[nn[u]];
            end[u] = rowptr[nn[u] + 1];
        }
        float acc[K1_U][K1_MAX_TB];
        int32_t arg[K1_U][K1_MAX_TB];
#pragma unroll
        for (int u = 0; u < K1_U; ++u)
#pragma unroll
            for (int t = 0; t < K1_MAX_TB; ++t) {
                arg[u][t] = (int32_t)E;
                acc[u][t] = Red<R>::identity();
                if (init_from_out && t < tb) acc[u][t] = Elem<T>::load(out + (b0 + t) * N + nn[u]);
            }
        int32_t maxlen = 0;
#pragma unroll
        for (int u = 0; u < K1_U; ++u) {
            if (n0 + (int64_t)u * NT >= N) end[u] = beg[u];  // past the end: nothing to do, nothing stored
            maxlen = (end[u] - beg[u] > maxlen) ? end[u] - beg[u] : maxlen;
        }
        for (int32_t sidx = 0; sidx < maxlen; ++sidx) {
            int32_t e[K1_U];
#pragma unroll
            for (int u = 0; u < K1_U; ++u) {
                const int32_t j = beg[u] + sidx;
                const int32_t jc = j < end[u] ? j : beg[u];   // beg[u] < E whenever the segment is not empty
                e[u] = (j < end[u]) ? (perm ? perm[jc] : jc) : -1;
            }
#pragma unroll
            for (int u = 0; u < K1_U; ++u) {
                if (e[u] < 0) continue;
#pragma unroll
                for (int t = 0; t < K1_MAX_TB; ++t) {
                    if (t < tb) {
                        const float f = Elem<T>::load(rows + (int64_t)t * E + e[u]);
                        if constexpr (IS_ARG) {
                            if (Red<R>::better(f, acc[u][t])) { acc[u][t] = f; arg[u][t] = e[u]; }
                        } else {
                            acc[u][t] = Red<R>::apply(acc[u][t], f);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < K1_U; ++u) {
            const int64_t n = n0 + (int64_t)u * NT;
            if (n >= N) continue;
            const int32_t len = end[u] - beg[u];
            if (len == 0 && init_from_out && !(IS_ARG && arg_out)) continue;  // the out row stays as it is
#pragma unroll
            for (int t = 0; t < K1_MAX_TB; ++t) {
                if (t < tb) {
                    float v = acc[u][t];
                    const int64_t o = (b0 + t) * N + n;
                    if constexpr (IS_ARG) {
                        if (!init_from_out && arg[u][t] == (int32_t)E) v = 0.f;
                        if (arg_out) arg_out[o] = arg[u][t];
                    } else if (R == GNNOPS_SUM) {
                        if (is_mean) v = v / (float)(len < 1 ? 1 : len);
                    }
                    Elem<T>::store(out + o, v);
                }
            }
        }
    }
}

template <typename T, int R>
int launch_seg(const void* src, const int32_t* rowptr, const int32_t* perm, void* out, int64_t* arg_out, int64_t B,
               int64_t E, int64_t K, int64_t N, int init_from_out, int is_mean, hipStream_t stream, void* hub_ws = nullptr,
               size_t hub_ws_bytes = 0) {
    constexpr int VEC = Elem<T>::VEC;
    const bool aligned = ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                         (arg_out == nullptr || (uintptr_t)arg_out % 16 == 0);
    // (small N or E: too little work per workgroup — the element kernel's thread-per-output mapping is the better one)
    if (K == 1 && B > 1 && E >= 512 && N >= K1_THREADS / 2 && (size_t)E * sizeof(T) <= K1_LDS_BYTES) {
        // rows per workgroup: as many as fit 40 KiB, so that four 512-thread workgroups share a CU (measured at the
        // reference's (10000, 10000) fp16 index_add_: 0.28 ms with two rows / 40 KB, 0.40 ms with one or with three)
        int tb = (int)(K1_LDS_TARGET / ((size_t)E * sizeof(T)));
        if (tb < 1) tb = 1;
        if (tb > K1_MAX_TB) tb = K1_MAX_TB;
        if (tb > B) tb = (int)B;
        static bool configured = false;
        if (!configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_k1_kernel<T, R>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)K1_LDS_BYTES) != hipSuccess)
                return gnnops_check_launch("seg_k1 attribute");
            configured = true;
        }
        const size_t lds = (size_t)tb * E * sizeof(T);
        hipLaunchKernelGGL((seg_k1_kernel<T, R>), dim3((unsigned)gnnops_cdiv(B, tb)),
                           dim3(lds > 80 * 1024 ? K1_MAX_THREADS : K1_THREADS), lds, stream, (const T*)src, rowptr, perm,
                           (T*)out, arg_out, B, E, N, tb, init_from_out, is_mean);
        return gnnops_check_launch("segment_reduce");
    }
    if (K % VEC == 0 && aligned) {
        const int64_t vecs = K / VEC;  // 16-B lanes per row
        int gshift = 0;
        while ((1 << gshift) < vecs && gshift < 6) ++gshift;
        const int G = 1 << gshift;
        const int kchunks = (int)gnnops_cdiv(vecs, G);
        const int64_t items = B * kchunks * N;
        // GNNOPS_SEG_MAP=gs: the grid-strided map of rounds 1-2 (A/B on one box: tools/ab_store_map.py)
        const char* mapenv = getenv("GNNOPS_SEG_MAP");
        const int blk_map = !(mapenv && mapenv[0] == 'g');
        int grid = blk_map ? gnnops_grid_cap(gnnops_cdiv(items, SEG_BLOCK), 256 * 16)
                           : gnnops_grid_cap(gnnops_cdiv(items, 256 >> gshift), 256 * 64);
        // hubs (hub.h) are set aside when the caller brought the workspace for them: plan form, one matrix
        constexpr bool IS_ARG_R = (R == GNNOPS_MIN || R == GNNOPS_MAX);
        const bool want_arg = IS_ARG_R;
        hub::Ws hw{};
        int hub_on = 0;
        if (hub_ws && perm && B == 1 && E > hub::T_HUB) {
            const hub::Layout hl = hub::layout(E, K, want_arg);
            if (hub_ws_bytes >= hl.total) {
                hw = hub::make_ws(hub_ws, hl, E, want_arg);
                if (gnnops_memset_async(hw.counters, 0, 8, stream) != hipSuccess) return gnnops_check_launch("hub memset");
                hub_on = 1;
            }
        }
        hipLaunchKernelGGL((seg_rows_kernel<T, R>), dim3(grid), dim3(256), 0, stream, (const T*)src, rowptr, perm,
                           (T*)out, arg_out, B, E, K, N, gshift, kchunks, init_from_out, is_mean, hw, hub_on, blk_map);
        if (hub_on)
            hub::launch_pass<T, R, false>((const T*)src, perm, nullptr, nullptr, (T*)out, arg_out, hw, E, K, gshift, kchunks,
                                          init_from_out, is_mean, stream);
    } else {
        int grid = gnnops_grid_cap(gnnops_cdiv(B * N * K, 256), 256 * 32);
        hipLaunchKernelGGL((seg_elems_kernel<T, R>), dim3(grid), dim3(256), 0, stream, (const T*)src, rowptr, perm,
                           (T*)out, arg_out, B, E, K, N, init_from_out, is_mean);
    }
    return gnnops_check_launch("segment_reduce");
}

template <typename T>
int dispatch_reduce(int reduce, const void* src, const int32_t* rowptr, const int32_t* perm, void* out,
                    int64_t* arg_out, int64_t B, int64_t E, int64_t K, int64_t N, int init_from_out,
                    hipStream_t stream, void* hw = nullptr, size_t hb = 0) {
    switch (reduce) {
        case GNNOPS_SUM: return launch_seg<T, GNNOPS_SUM>(src, rowptr, perm, out, nullptr, B, E, K, N, init_from_out, 0, stream, hw, hb);
        case GNNOPS_MEAN: return launch_seg<T, GNNOPS_SUM>(src, rowptr, perm, out, nullptr, B, E, K, N, init_from_out, 1, stream, hw, hb);
        case GNNOPS_MUL: return launch_seg<T, GNNOPS_MUL>(src, rowptr, perm, out, nullptr, B, E, K, N, init_from_out, 0, stream, hw, hb);
        case GNNOPS_MIN: return launch_seg<T, GNNOPS_MIN>(src, rowptr, perm, out, arg_out, B, E, K, N, init_from_out, 0, stream, hw, hb);
        case GNNOPS_MAX: return launch_seg<T, GNNOPS_MAX>(src, rowptr, perm, out, arg_out, B, E, K, N, init_from_out, 0, stream, hw, hb);
    }
    gnnops_set_error("segment_reduce: unknown reduce %d", reduce);
    return GNNOPS_EINVAL;
}

}  // namespace

extern "C" size_t gnnops_hub_workspace_bytes(int64_t E, int64_t K, int reduce) {
    if (E <= hub::T_HUB || K < 0) return 0;  // K == 0: the lists only (push-form index_select)
    return hub::layout(E, K, reduce == GNNOPS_MIN || reduce == GNNOPS_MAX).total;
}

extern "C" int gnnops_segment_reduce(const void* src, const int32_t* rowptr, const int32_t* perm, void* out,
                                     int64_t* arg_out, int64_t B, int64_t E, int64_t K, int64_t N, int dtype,
                                     int reduce, int init_from_out, gnnops_stream_t s) {
    return gnnops_segment_reduce_hubs(src, rowptr, perm, out, arg_out, B, E, K, N, dtype, reduce, init_from_out, nullptr, 0, s);
}

extern "C" int gnnops_segment_reduce_hubs(const void* src, const int32_t* rowptr, const int32_t* perm, void* out,
                                          int64_t* arg_out, int64_t B, int64_t E, int64_t K, int64_t N, int dtype,
                                          int reduce, int init_from_out, void* hub_workspace, size_t hub_workspace_bytes,
                                          gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && E >= 0 && K >= 0 && N >= 0, GNNOPS_EINVAL, "segment_reduce: negative size");
    GNNOPS_REQUIRE(E < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "segment_reduce: E must be < 2^31");
    GNNOPS_REQUIRE(!(reduce == GNNOPS_MEAN && init_from_out), GNNOPS_EINVAL,
                   "segment_reduce: mean cannot start from out");
    if (B * N * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && out && (E == 0 || src), GNNOPS_EINVAL, "segment_reduce: null pointer");
    switch (dtype) {
        case GNNOPS_F32: return dispatch_reduce<float>(reduce, src, rowptr, perm, out, arg_out, B, E, K, N, init_from_out, stream, hub_workspace, hub_workspace_bytes);
        case GNNOPS_F16: return dispatch_reduce<__half>(reduce, src, rowptr, perm, out, arg_out, B, E, K, N, init_from_out, stream, hub_workspace, hub_workspace_bytes);
        case GNNOPS_BF16: return dispatch_reduce<__hip_bfloat16>(reduce, src, rowptr, perm, out, arg_out, B, E, K, N, init_from_out, stream, hub_workspace, hub_workspace_bytes);
    }
    gnnops_set_error("segment_reduce: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}
