// fused.hip — index_select(index_add(input, dim, index, other), dim, index).sum(dim) in one pass
// (reference: op_bm_scripts/benchmark_fused_index_add_reduce.py:12-20; `other = input.clone()`,
// index length = input.shape[dim]). The reference materialises a full clone (index_add), a full gather
// and then reduces — 5x the input in reserved memory (mem_prof_data/fused_index_add_reduce.csv:200).
//
// With cnt[n] = #{j : index[j] == n} (from the plan of index):
//   out[b,k] = sum_j tmp[b,index[j],k] = sum_n cnt[n] * tmp[b,n,k],
//   tmp[b,n,k] = round_T( input[b,n,k] + sum_{e in segment n, in order} other[b,e,k] )   (index_add in fp32, one rounding)
// Nothing of size [B,N,K] is written. The result is returned in fp32 (the reference's fp16 sum overflows
// at its own sizes). HBM-bound: input and other are each read once.
//   K >= 2 (dim 0 of a matrix): one thread per (b,k) column and row slice, coalesced along k, partials
//     combined in slice order by a second launch (deterministic).
//   K == 1 (dim 1 of a matrix): one wave per row b, lanes stride over n (coalesced), wave reduction.
#include "common.h"

namespace {

constexpr int SLICES = 64;

template <typename T>
__device__ inline float round_to(float v) {
    T t;
    Elem<T>::store(&t, v);
    return Elem<T>::load(&t);
}

template <typename T>
__global__ __launch_bounds__(256) void fused_cols_kernel(const T* __restrict__ input, const T* __restrict__ other,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ perm, float* __restrict__ partial,
                                                         int64_t B, int64_t N, int64_t E, int64_t K) {
    const int64_t bk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (bk >= B * K) return;
    const int64_t b = bk / K, k = bk % K;
    const int slice = blockIdx.y;
    const int64_t n_lo = N * slice / SLICES, n_hi = N * (slice + 1) / SLICES;
    float acc = 0.f;
    for (int64_t n = n_lo; n < n_hi; ++n) {
        const int32_t beg = rowptr[n], end = rowptr[n + 1];
        if (beg == end) continue;  // cnt = 0: the row is never selected
        float t = Elem<T>::load(input + (b * N + n) * K + k);
        for (int32_t j = beg; j < end; ++j) t += Elem<T>::load(other + (b * E + perm[j]) * K + k);
        acc += (float)(end - beg) * round_to<T>(t);
    }
    partial[(int64_t)slice * B * K + bk] = acc;
}

__global__ void combine_slices_kernel(const float* __restrict__ partial, float* __restrict__ out, int64_t BK) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BK) return;
    float acc = 0.f;
    for (int s = 0; s < SLICES; ++s) acc += partial[(int64_t)s * BK + i];
    out[i] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void fused_rows_kernel(const T* __restrict__ input, const T* __restrict__ other,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ perm, float* __restrict__ out,
                                                         int64_t B, int64_t N, int64_t E) {
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= B) return;
    const int lane = lane_id();
    float acc = 0.f;
    for (int64_t n = lane; n < N; n += 64) {
        const int32_t beg = rowptr[n], end = rowptr[n + 1];
        if (beg == end) continue;
        float t = Elem<T>::load(input + b * N + n);
        for (int32_t j = beg; j < end; ++j) t += Elem<T>::load(other + b * E + perm[j]);
        acc += (float)(end - beg) * round_to<T>(t);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[b] = acc;
}

// K == 1, rows of `other` that fit in LDS (the reference's (L, L) fp16 shapes): a workgroup parks TB rows other[b, :] on
// chip with coalesced 16-B loads, then every thread walks destinations n (four at a time, so their row-pointer, input and
// perm loads are in flight together) and reads the contributions from LDS instead of gathering 2-byte values from HBM.
constexpr int FR_THREADS = 512, FR_MAX_TB = 4, FR_U = 4;
constexpr size_t FR_LDS_TARGET = 40 * 1024, FR_LDS_MAX = 64 * 1024;

template <typename T>
__global__ __launch_bounds__(FR_THREADS) void fused_rows_lds_kernel(const T* __restrict__ input, const T* __restrict__ other,
                                                                    const int32_t* __restrict__ rowptr,
                                                                    const int32_t* __restrict__ perm, float* __restrict__ out,
                                                                    int64_t B, int64_t N, int64_t E, int TB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fr_raw[];
    T* rows = reinterpret_cast<T*>(fr_raw);  // [TB][E]
    __shared__ float s_part[FR_THREADS / 64][FR_MAX_TB];
    const int64_t b0 = (int64_t)blockIdx.x * TB;
    const int tb = (int)((B - b0 < TB) ? (B - b0) : TB);
    const T* sb = other + b0 * E;
    const int64_t nelem = (int64_t)tb * E;
    if ((((uintptr_t)sb) & 15) == 0) {
        constexpr int PER = 16 / (int)sizeof(T);
        const int64_t nvec = nelem / PER;
        const u32x4* sv = reinterpret_cast<const u32x4*>(sb);
        u32x4* dv = reinterpret_cast<u32x4*>(rows);
        int64_t i = threadIdx.x;
        for (; i + 3 * FR_THREADS < nvec; i += 4 * FR_THREADS) {
            const u32x4 a = sv[i], b = sv[i + FR_THREADS], c = sv[i + 2 * FR_THREADS], d = sv[i + 3 * FR_THREADS];
            dv[i] = a; dv[i + FR_THREADS] = b; dv[i + 2 * FR_THREADS] = c; dv[i + 3 * FR_THREADS] = d;
        }
        for (; i < nvec; i += FR_THREADS) dv[i] = sv[i];
        for (int64_t j = nvec * PER + threadIdx.x; j < nelem; j += FR_THREADS) rows[j] = sb[j];
    } else {
        for (int64_t i = threadIdx.x; i < nelem; i += FR_THREADS) rows[i] = sb[i];
    }
    __syncthreads();

    float acc[FR_MAX_TB];
#pragma unroll
    for (int t = 0; t < FR_MAX_TB; ++t) acc[t] = 0.f;
    for (int64_t n0 = threadIdx.x; n0 < N; n0 += (int64_t)FR_THREADS * FR_U) {
        int32_t beg[FR_U], end[FR_U];
        float tv[FR_U][FR_MAX_TB];
#pragma unroll
        for (int u = 0; u < FR_U; ++u) {
            const int64_t n = n0 + (int64_t)u * FR_THREADS;
            const int64_t nc = n < N ? n : N - 1;
            beg[u] = rowptr[nc];
            end[u] = rowptr[nc + 1];
#pragma unroll
            for (int t = 0; t < FR_MAX_TB; ++t) tv[u][t] = (t < tb) ? Elem<T>::load(input + (b0 + t) * N + nc) : 0.f;
        }
        int32_t maxlen = 0;
#pragma unroll
        for (int u = 0; u < FR_U; ++u) {
            if (n0 + (int64_t)u * FR_THREADS >= N) end[u] = beg[u];
            maxlen = (end[u] - beg[u] > maxlen) ? end[u] - beg[u] : maxlen;
        }
        for (int32_t sidx = 0; sidx < maxlen; ++sidx) {
            int32_t e[FR_U];
#pragma unroll
            for (int u = 0; u < FR_U; ++u) {
                const int32_t j = beg[u] + sidx;
                e[u] = (j < end[u]) ? perm[j < end[u] ? j : beg[u]] : -1;
            }
#pragma unroll
            for (int u = 0; u < FR_U; ++u) {
                if (e[u] < 0) continue;
#pragma unroll
                for (int t = 0; t < FR_MAX_TB; ++t)
                    if (t < tb) tv[u][t] += Elem<T>::load(rows + (int64_t)t * E + e[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < FR_U; ++u) {
            const int32_t cnt = end[u] - beg[u];
            if (cnt == 0) continue;  // the row is never selected
#pragma unroll
            for (int t = 0; t < FR_MAX_TB; ++t)
                if (t < tb) acc[t] += (float)cnt * round_to<T>(tv[u][t]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < FR_MAX_TB; ++t) {
        float a = acc[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) s_part[wave][t] = a;
    }
    __syncthreads();
    if (threadIdx.x < tb) {
        float a = 0.f;
        for (int w = 0; w < FR_THREADS / 64; ++w) a += s_part[w][threadIdx.x];
        out[b0 + threadIdx.x] = a;
    }
}

template <typename T>
int launch(const void* input, const void* other, const int32_t* rowptr, const int32_t* perm, float* out, int64_t B,
           int64_t N, int64_t E, int64_t K, float* partial, hipStream_t stream) {
    if (K == 1 && E >= 512 && N >= FR_THREADS / 2 && (size_t)E * sizeof(T) <= FR_LDS_MAX) {
        int tb = (int)(FR_LDS_TARGET / ((size_t)E * sizeof(T)));
        if (tb < 1) tb = 1;
        if (tb > FR_MAX_TB) tb = FR_MAX_TB;
        if (tb > B) tb = (int)B;
        hipLaunchKernelGGL((fused_rows_lds_kernel<T>), dim3((unsigned)gnnops_cdiv(B, tb)), dim3(FR_THREADS),
                           (size_t)tb * E * sizeof(T), stream, (const T*)input, (const T*)other, rowptr, perm, out, B, N, E, tb);
    } else if (K == 1) {
        const int grid = (int)gnnops_cdiv(B * 64, 256);
        hipLaunchKernelGGL((fused_rows_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)input, (const T*)other,
                           rowptr, perm, out, B, N, E);
    } else {
        dim3 grid((unsigned)gnnops_cdiv(B * K, 256), SLICES);
        hipLaunchKernelGGL((fused_cols_kernel<T>), grid, dim3(256), 0, stream, (const T*)input, (const T*)other, rowptr,
                           perm, partial, B, N, E, K);
        hipLaunchKernelGGL(combine_slices_kernel, dim3((unsigned)gnnops_cdiv(B * K, 256)), dim3(256), 0, stream, partial,
                           out, B * K);
    }
    return gnnops_check_launch("fused_index_add_select_sum");
}

}  // namespace

extern "C" size_t gnnops_fused_index_add_select_sum_workspace_bytes(int64_t B, int64_t K) {
    if (B < 0 || K < 0) return 0;
    return (size_t)SLICES * (size_t)(B * K) * sizeof(float) + 256;
}

extern "C" int gnnops_fused_index_add_select_sum(const void* input, const void* other, const int32_t* rowptr,
                                                 const int32_t* perm, float* out_f32, int64_t B, int64_t N, int64_t E,
                                                 int64_t K, int dtype, void* workspace, size_t workspace_bytes,
                                                 gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && N >= 0 && E >= 0 && K >= 0, GNNOPS_EINVAL, "fused_index_add_select_sum: negative size");
    if (B * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(out_f32 && rowptr && (N == 0 || input) && (E == 0 || (other && perm)), GNNOPS_EINVAL,
                   "fused_index_add_select_sum: null pointer");
    GNNOPS_REQUIRE(K == 1 || (workspace && workspace_bytes >= gnnops_fused_index_add_select_sum_workspace_bytes(B, K)),
                   GNNOPS_EWORKSPACE, "fused_index_add_select_sum: workspace too small");
    GNNOPS_REQUIRE(gnnops_cdiv(B * K, 256) < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "fused: too many columns");
    float* partial = (float*)workspace;
    switch (dtype) {
        case GNNOPS_F32: return launch<float>(input, other, rowptr, perm, out_f32, B, N, E, K, partial, stream);
        case GNNOPS_F16: return launch<__half>(input, other, rowptr, perm, out_f32, B, N, E, K, partial, stream);
        case GNNOPS_BF16: return launch<__hip_bfloat16>(input, other, rowptr, perm, out_f32, B, N, E, K, partial, stream);
    }
    gnnops_set_error("fused_index_add_select_sum: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}
