// backward.hip — the two device pieces the backward passes of the segment / sparse ops need besides the forward kernels
// (autograd for what PyG's aggregation calls on the reference's OpProfiler training loop,
// graph_benchmark/profile/OpProfiler.py:259-292):
//
//   gnnops_rowptr_expand   index[e] = the segment of a CSR pointer that holds position e (torch_scatter.gather_csr's
//                          addressing; also the gather that is segment_csr's backward) — one binary search per position
//   gnnops_sddmm           out[k] = <a[ra[k], :], b[rb[k], :]> per nonzero: d(value) of torch_sparse.spmm
//                          (benchmark_sparse_spmm.py:12-14 in training), fp32 accumulation, one rounding
#include "common.h"

namespace {

__global__ void rowptr_expand_kernel(const int32_t* __restrict__ rowptr, int64_t N, int64_t E, int64_t* __restrict__ index) {
    const int32_t first = rowptr[0], last = rowptr[N];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < first || e >= last) {  // no segment holds this position
            index[e] = N;
            continue;
        }
        int64_t lo = 0, hi = N;        // largest n with rowptr[n] <= e (empty segments are skipped by the upper bound)
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)rowptr[mid] <= e) lo = mid; else hi = mid;
        }
        index[e] = lo;
    }
}

// One lane group of 2^gshift lanes per nonzero; every lane takes 16-B pieces of both rows.
template <typename T>
__global__ __launch_bounds__(256) void sddmm_kernel(const int64_t* __restrict__ ra, const int64_t* __restrict__ rb,
                                                    const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                    int64_t nnz, int64_t D, int gshift) {
    constexpr int VEC = Elem<T>::VEC;
    const int G = 1 << gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> gshift;
    const int gl = (int)(gtid & (G - 1));
    const bool vec = D % VEC == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0;
    for (int64_t k = gtid >> gshift; k < nnz; k += ngroups) {
        const T* pa = a + ra[k] * D;
        const T* pb = b + rb[k] * D;
        float acc = 0.f;
        if (vec) {
            for (int64_t c = (int64_t)gl * VEC; c < D; c += (int64_t)G * VEC) {
                float fa[VEC], fb[VEC];
                Elem<T>::unpack(*reinterpret_cast<const u32x4*>(pa + c), fa);
                Elem<T>::unpack(*reinterpret_cast<const u32x4*>(pb + c), fb);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc = __fadd_rn(acc, __fmul_rn(fa[v], fb[v]));
            }
        } else {
            for (int64_t c = gl; c < D; c += G) acc = __fadd_rn(acc, __fmul_rn(Elem<T>::load(pa + c), Elem<T>::load(pb + c)));
        }
        for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (gl == 0) Elem<T>::store(out + k, acc);
    }
}

template <typename T>
int launch_sddmm(const int64_t* ra, const int64_t* rb, const void* a, const void* b, void* out, int64_t nnz, int64_t D,
                 hipStream_t stream) {
    const int64_t pieces = gnnops_cdiv(D, Elem<T>::VEC);
    int gshift = 0;
    while ((1 << gshift) < pieces && gshift < 6) ++gshift;
    const int grid = gnnops_grid_cap(gnnops_cdiv(nnz, 256 >> gshift), 256 * 32);
    hipLaunchKernelGGL((sddmm_kernel<T>), dim3(grid), dim3(256), 0, stream, ra, rb, (const T*)a, (const T*)b, (T*)out, nnz, D,
                       gshift);
    return gnnops_check_launch("sddmm");
}

}  // namespace

extern "C" int gnnops_rowptr_expand(const int32_t* rowptr, int64_t N, int64_t E, int64_t* index, gnnops_stream_t s) {
    GNNOPS_REQUIRE(N >= 0 && E >= 0, GNNOPS_EINVAL, "rowptr_expand: negative size");
    if (E == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && index, GNNOPS_EINVAL, "rowptr_expand: null pointer");
    const int grid = gnnops_grid_cap(gnnops_cdiv(E, 256), 256 * 16);
    hipLaunchKernelGGL(rowptr_expand_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, rowptr, N, E, index);
    return gnnops_check_launch("rowptr_expand");
}

extern "C" int gnnops_sddmm(const int64_t* rows_a, const int64_t* rows_b, const void* a, const void* b, void* out, int64_t nnz,
                            int64_t D, int dtype, gnnops_stream_t s) {
    GNNOPS_REQUIRE(nnz >= 0 && D >= 0, GNNOPS_EINVAL, "sddmm: negative size");
    if (nnz == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rows_a && rows_b && out && (D == 0 || (a && b)), GNNOPS_EINVAL, "sddmm: null pointer");
    switch (dtype) {
        case GNNOPS_F32: return launch_sddmm<float>(rows_a, rows_b, a, b, out, nnz, D, (hipStream_t)s);
        case GNNOPS_F16: return launch_sddmm<__half>(rows_a, rows_b, a, b, out, nnz, D, (hipStream_t)s);
        case GNNOPS_BF16: return launch_sddmm<__hip_bfloat16>(rows_a, rows_b, a, b, out, nnz, D, (hipStream_t)s);
    }
    gnnops_set_error("sddmm: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}

// ---- destination-partitioned scatter across GPUs (gnnops/dist.py): how many of this rank's edges go to each owner ----
// counts[g] = #{e : g * per <= index[e] < (g + 1) * per}, g < G <= 64 (contiguous slabs of `per` destination rows).
// One streaming read of the index; per wave one ballot per owner, one LDS add per wave and owner, one global add per
// workgroup and owner.
namespace {
__global__ __launch_bounds__(256) void owner_counts_kernel(const int64_t* __restrict__ index, int64_t E, int64_t per, int G,
                                                           unsigned long long* __restrict__ counts) {
    __shared__ unsigned int s_cnt[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = (E + stride - 1) / stride;   // every wave runs the same number of rounds: ballots stay full
    for (int64_t r = 0; r < rounds; ++r) {
        const int64_t e = r * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        int owner = -1;
        if (e < E) {
            const int64_t d = index[e];
            if (d >= 0 && d < (int64_t)G * per) {   // an id outside [0, G * per) is counted for NO owner: the counts then do not
                owner = 0;                           // add up to E and the caller (dist.py) raises instead of misrouting the edge
                for (int g = 1; g < G; ++g) owner += (d >= (int64_t)g * per) ? 1 : 0;
            }
        }
        for (int g = 0; g < G; ++g) {
            const unsigned long long m = __ballot(owner == g);
            if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cnt[g], (unsigned int)__popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x < G && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}
}  // namespace

extern "C" int gnnops_owner_counts(const int64_t* index, int64_t E, int64_t rows_per_owner, int owners, int64_t* counts,
                                   gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(E >= 0 && rows_per_owner > 0 && owners >= 1 && owners <= 64, GNNOPS_EINVAL,
                   "owner_counts: bad argument (E=%lld per=%lld owners=%d)", (long long)E, (long long)rows_per_owner, owners);
    GNNOPS_REQUIRE(counts && (E == 0 || index), GNNOPS_EINVAL, "owner_counts: null pointer");
    if (gnnops_memset_async(counts, 0, sizeof(int64_t) * owners, stream) != hipSuccess) return gnnops_check_launch("owner_counts memset");
    if (E == 0) return GNNOPS_OK;
    const int grid = gnnops_grid_cap(gnnops_cdiv(E, 256 * 8), 256 * 8);
    hipLaunchKernelGGL(owner_counts_kernel, dim3(grid), dim3(256), 0, stream, index, E, rows_per_owner, owners,
                       (unsigned long long*)counts);
    return gnnops_check_launch("owner_counts");
}
