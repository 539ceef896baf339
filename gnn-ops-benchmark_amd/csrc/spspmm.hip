// spspmm.hip — sparse x sparse: torch_sparse.spspmm(indexA, valueA, indexB, valueB, m, k, n) and
// torch.sparse.mm(COO, COO) (reference: op_bm_scripts/benchmark_sparse_spspmm.py:12-14,94-95; the reference's
// CUDA path is expand-sort-compress too, ops_to_kernels.md:12).
//
// Expand - sort - compress on the device:
//   count    products per nonzero a of A = length of row colA[a] of B (CSR view of B from the plan builder);
//            block sums -> exclusive scan -> total P (the caller sizes the expansion from it)
//   expand   product p of nonzero a: row = rowA[a], col = colB[e], val = round(valA[a] * valB[e]), e walking
//            B's row in stored order; p enumerates (a, e) lexicographically
//   compress gnnops_coalesce on the expansion: stable 64-bit radix sort by (row, col) packed as row << bits(n) | col, duplicates summed in
//            expansion order (fp32) -> coalesced COO, bit-reproducible
// Index-heavy and HBM-bound; nothing here is a dense contraction.
#include "common.h"

namespace {

constexpr int T = 256;

__device__ inline uint32_t products_of(const int64_t* colA, const int32_t* rowptrB, int64_t a) {
    const int64_t kk = colA[a];
    return (uint32_t)(rowptrB[kk + 1] - rowptrB[kk]);
}

__global__ __launch_bounds__(T) void count_kernel(const int64_t* __restrict__ colA, int64_t nnzA,
                                                  const int32_t* __restrict__ rowptrB, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_tmp[T / 64];
    const int64_t a = (int64_t)blockIdx.x * T + threadIdx.x;
    const uint32_t c = a < nnzA ? products_of(colA, rowptrB, a) : 0u;
    uint32_t tot;
    block_excl_scan_u32<T / 64>(c, s_tmp, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(T) void scan_sums_kernel(uint32_t* __restrict__ block_sums, int nb, int64_t* __restrict__ d_total) {
    __shared__ uint32_t s_tmp[T / 64];
    uint64_t carry = 0;
    for (int base = 0; base < nb; base += T) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nb ? block_sums[i] : 0u;
        uint32_t tot;
        const uint32_t off = (uint32_t)carry + block_excl_scan_u32<T / 64>(v, s_tmp, &tot);
        if (i < nb) block_sums[i] = off;
        carry += tot;
    }
    if (threadIdx.x == 0) *d_total = (int64_t)carry;  // > 2^32 - 1 is rejected by the host wrapper
}

// Load-balanced expansion: the T nonzeros of a workgroup own a contiguous run of products; their start offsets sit in
// LDS and the threads sweep the RUN (thread t takes products t, t + T, ...), finding the owning nonzero by binary search,
// so the three output streams are written coalesced whatever the row lengths of B (a thread per nonzero writes its
// products T-strided: 0.51 ms for 8.8M products at the reference shape, 5x the time of everything it moves).
template <typename V>
__global__ __launch_bounds__(T) void expand_kernel(const int64_t* __restrict__ rowA, const int64_t* __restrict__ colA,
                                                   const V* __restrict__ valA, int64_t nnzA,
                                                   const int32_t* __restrict__ rowptrB, const int32_t* __restrict__ permB,
                                                   const int64_t* __restrict__ colB, const V* __restrict__ valB,
                                                   const uint32_t* __restrict__ block_off, int64_t* __restrict__ out_row,
                                                   int64_t* __restrict__ out_col, V* __restrict__ out_val) {
    __shared__ uint32_t s_tmp[T / 64];
    __shared__ uint32_t s_start[T + 1];   // first product of nonzero t inside the workgroup's run
    __shared__ int32_t s_bbeg[T];         // where its row of B starts
    __shared__ int64_t s_row[T];
    __shared__ float s_va[T];
    const int64_t a = (int64_t)blockIdx.x * T + threadIdx.x;
    uint32_t c = 0;
    if (a < nnzA) {
        const int64_t kk = colA[a];
        s_bbeg[threadIdx.x] = rowptrB[kk];
        c = (uint32_t)(rowptrB[kk + 1] - rowptrB[kk]);
        s_row[threadIdx.x] = rowA[a];
        s_va[threadIdx.x] = Elem<V>::load(valA + a);
    }
    uint32_t tot;
    const uint32_t off = block_excl_scan_u32<T / 64>(c, s_tmp, &tot);
    s_start[threadIdx.x] = off;
    if (threadIdx.x == 0) s_start[T] = tot;
    __syncthreads();
    const uint32_t base = block_off[blockIdx.x];
    for (uint32_t q = threadIdx.x; q < tot; q += T) {
        int lo = 0, hi = T;  // largest t with s_start[t] <= q (empty nonzeros share a start: the last one owns q)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_start[mid] <= q) lo = mid; else hi = mid;
        }
        const int32_t e = permB[s_bbeg[lo] + (int32_t)(q - s_start[lo])];
        const uint32_t p = base + q;
        out_row[p] = s_row[lo];
        out_col[p] = colB[e];
        Elem<V>::store(out_val + p, s_va[lo] * Elem<V>::load(valB + e));
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" size_t gnnops_spspmm_workspace_bytes(int64_t nnzA) {
    if (nnzA < 0) return 0;
    return align_up((size_t)gnnops_cdiv(nnzA > 0 ? nnzA : 1, T) * 4, 256) + 256;
}

// Phase 1: *d_total = number of products; block offsets are left in `workspace` for phase 2.
extern "C" int gnnops_spspmm_count(const int64_t* colA, int64_t nnzA, const int32_t* rowptrB, int64_t* d_total,
                                   void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(nnzA >= 0 && d_total, GNNOPS_EINVAL, "spspmm_count: bad arguments");
    GNNOPS_REQUIRE(workspace && workspace_bytes >= gnnops_spspmm_workspace_bytes(nnzA), GNNOPS_EWORKSPACE,
                   "spspmm_count: workspace too small");
    if (nnzA == 0) {
        if (gnnops_memset_async(d_total, 0, sizeof(int64_t), stream) != hipSuccess) return gnnops_check_launch("spspmm memset");
        return GNNOPS_OK;
    }
    GNNOPS_REQUIRE(colA && rowptrB, GNNOPS_EINVAL, "spspmm_count: null pointer");
    const int nb = (int)gnnops_cdiv(nnzA, T);
    uint32_t* block_sums = (uint32_t*)workspace;
    hipLaunchKernelGGL(count_kernel, dim3(nb), dim3(T), 0, stream, colA, nnzA, rowptrB, block_sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(T), 0, stream, block_sums, nb, d_total);
    return gnnops_check_launch("spspmm_count");
}

// Phase 2: write the P products (row, col, val) in (a, e) order; `workspace` is the one phase 1 filled.
extern "C" int gnnops_spspmm_expand(const int64_t* rowA, const int64_t* colA, const void* valA, int64_t nnzA,
                                    const int32_t* rowptrB, const int32_t* permB, const int64_t* colB, const void* valB,
                                    int64_t* out_row, int64_t* out_col, void* out_val, int dtype, const void* workspace,
                                    gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(nnzA >= 0, GNNOPS_EINVAL, "spspmm_expand: negative size");
    if (nnzA == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowA && colA && valA && rowptrB && permB && colB && valB && out_row && out_col && out_val && workspace,
                   GNNOPS_EINVAL, "spspmm_expand: null pointer");
    const int nb = (int)gnnops_cdiv(nnzA, T);
    const uint32_t* block_off = (const uint32_t*)workspace;
#define EXPAND(V)                                                                                                     \
    hipLaunchKernelGGL((expand_kernel<V>), dim3(nb), dim3(T), 0, stream, rowA, colA, (const V*)valA, nnzA, rowptrB,   \
                       permB, colB, (const V*)valB, block_off, out_row, out_col, (V*)out_val)
    switch (dtype) {
        case GNNOPS_F32: EXPAND(float); break;
        case GNNOPS_F16: EXPAND(__half); break;
        case GNNOPS_BF16: EXPAND(__hip_bfloat16); break;
        default: gnnops_set_error("spspmm_expand: unknown dtype %d", dtype); return GNNOPS_EINVAL;
    }
#undef EXPAND
    return gnnops_check_launch("spspmm_expand");
}
