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

template <typename T>
int launch(const void* input, const void* other, const int32_t* rowptr, const int32_t* perm, float* out, int64_t B,
           int64_t N, int64_t E, int64_t K, float* partial, hipStream_t stream) {
    if (K == 1) {
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
