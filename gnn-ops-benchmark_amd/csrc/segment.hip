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

namespace {

constexpr int U = 8;       // contribution rows in flight per lane group
// Source rows and output rows are touched exactly once: nontemporal accesses keep them from evicting
// rowptr / perm lines (measured -5 % kernel time at config 2; tools/time_seg.py).
constexpr bool NT = true;

// Row form: the row is K elements, K % VEC == 0, 16-B aligned. A group of G = 2^gshift lanes owns one
// (b, n, chunk) item; chunk c covers elements [c*G*VEC, (c+1)*G*VEC).
template <typename T, int R>
__global__ __launch_bounds__(256) void seg_rows_kernel(const T* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ perm, T* __restrict__ out,
                                                       int64_t* __restrict__ arg_out, int64_t B, int64_t E, int64_t K,
                                                       int64_t N, int gshift, int kchunks, int init_from_out,
                                                       int is_mean) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr bool IS_ARG = (R == GNNOPS_MIN || R == GNNOPS_MAX);
    const int G = 1 << gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> gshift;
    const int gl = (int)(gtid & (G - 1));
    const int64_t items = B * (int64_t)kchunks * N;

    for (int64_t item = gtid >> gshift; item < items; item += ngroups) {
        const int64_t n = item % N;
        const int64_t bc = item / N;
        const int chunk = (int)(bc % kchunks);
        const int64_t b = bc / kchunks;
        const int64_t col = ((int64_t)chunk * G + gl) * VEC;
        if (col >= K) continue;
        const int32_t beg = rowptr[n], end = rowptr[n + 1];
        if (beg == end && init_from_out && !(IS_ARG && arg_out)) continue;  // nothing to fold in: the out row stays as it is
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

template <typename T, int R>
int launch_seg(const void* src, const int32_t* rowptr, const int32_t* perm, void* out, int64_t* arg_out, int64_t B,
               int64_t E, int64_t K, int64_t N, int init_from_out, int is_mean, hipStream_t stream) {
    constexpr int VEC = Elem<T>::VEC;
    const bool aligned = ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                         (arg_out == nullptr || (uintptr_t)arg_out % 16 == 0);
    if (K % VEC == 0 && aligned) {
        const int64_t vecs = K / VEC;  // 16-B lanes per row
        int gshift = 0;
        while ((1 << gshift) < vecs && gshift < 6) ++gshift;
        const int G = 1 << gshift;
        const int kchunks = (int)gnnops_cdiv(vecs, G);
        const int64_t items = B * kchunks * N;
        const int64_t groups_per_block = 256 >> gshift;
        int grid = gnnops_grid_cap(gnnops_cdiv(items, groups_per_block), 256 * 64);
        hipLaunchKernelGGL((seg_rows_kernel<T, R>), dim3(grid), dim3(256), 0, stream, (const T*)src, rowptr, perm,
                           (T*)out, arg_out, B, E, K, N, gshift, kchunks, init_from_out, is_mean);
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
                    hipStream_t stream) {
    switch (reduce) {
        case GNNOPS_SUM: return launch_seg<T, GNNOPS_SUM>(src, rowptr, perm, out, nullptr, B, E, K, N, init_from_out, 0, stream);
        case GNNOPS_MEAN: return launch_seg<T, GNNOPS_SUM>(src, rowptr, perm, out, nullptr, B, E, K, N, init_from_out, 1, stream);
        case GNNOPS_MUL: return launch_seg<T, GNNOPS_MUL>(src, rowptr, perm, out, nullptr, B, E, K, N, init_from_out, 0, stream);
        case GNNOPS_MIN: return launch_seg<T, GNNOPS_MIN>(src, rowptr, perm, out, arg_out, B, E, K, N, init_from_out, 0, stream);
        case GNNOPS_MAX: return launch_seg<T, GNNOPS_MAX>(src, rowptr, perm, out, arg_out, B, E, K, N, init_from_out, 0, stream);
    }
    gnnops_set_error("segment_reduce: unknown reduce %d", reduce);
    return GNNOPS_EINVAL;
}

}  // namespace

extern "C" int gnnops_segment_reduce(const void* src, const int32_t* rowptr, const int32_t* perm, void* out,
                                     int64_t* arg_out, int64_t B, int64_t E, int64_t K, int64_t N, int dtype,
                                     int reduce, int init_from_out, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && E >= 0 && K >= 0 && N >= 0, GNNOPS_EINVAL, "segment_reduce: negative size");
    GNNOPS_REQUIRE(E < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "segment_reduce: E must be < 2^31");
    GNNOPS_REQUIRE(!(reduce == GNNOPS_MEAN && init_from_out), GNNOPS_EINVAL,
                   "segment_reduce: mean cannot start from out");
    if (B * N * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && out && (E == 0 || src), GNNOPS_EINVAL, "segment_reduce: null pointer");
    switch (dtype) {
        case GNNOPS_F32: return dispatch_reduce<float>(reduce, src, rowptr, perm, out, arg_out, B, E, K, N, init_from_out, stream);
        case GNNOPS_F16: return dispatch_reduce<__half>(reduce, src, rowptr, perm, out, arg_out, B, E, K, N, init_from_out, stream);
        case GNNOPS_BF16: return dispatch_reduce<__hip_bfloat16>(reduce, src, rowptr, perm, out, arg_out, B, E, K, N, init_from_out, stream);
    }
    gnnops_set_error("segment_reduce: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}
