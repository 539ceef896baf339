// spmm.hip — sparse (CSR / plan-ordered COO) x dense: torch_sparse.spmm(index, value, m, n, matrix) and
// torch.sparse.mm(COO, dense) (reference: op_bm_scripts/benchmark_sparse_spmm.py:12-14; BASELINE
// config 3: CSR 2M x 2M, nnz 40M, D=256 bf16).
//
// out[i, :] = sum over the nonzeros e of row i, in order, of value[e] * mat[col[e], :]
// (torch_sparse: scatter_add(matrix.index_select(-2, col) * value.unsqueeze(-1), row, dim=-2)).
//
// Row-split: one lane group per output row (and 1-KiB column chunk), 8 gathered rows of `mat` in flight,
// fp32 accumulation, one rounding on store. HBM/Infinity-Cache bound (2*nnz*D flops on ~nnz*D*s gathered
// bytes: arithmetic intensity ~1 flop/B against a ridge of ~300), so this is VALU FMA work on 16-B lane
// loads, not an MFMA tile: no two sparse rows share a dense operand tile at 20 nnz/row.
//
// Arithmetic (matches the oracle bit for bit): fp32 inputs: p = round(v*x), acc = round(acc+p) — the two
// roundings torch's `matrix[col] * value` followed by scatter_add perform; no FMA contraction.
// 16-bit inputs: v*x is exact in fp32, acc in fp32, rounded once to the storage type.
#include "common.h"

namespace {

constexpr int U = 8;

template <typename T>
__global__ __launch_bounds__(256) void spmm_rows_kernel(const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ perm,
                                                        const int64_t* __restrict__ col, const T* __restrict__ value,
                                                        const T* __restrict__ mat, T* __restrict__ out, int64_t M,
                                                        int64_t D, int gshift, int kchunks) {
    constexpr int VEC = Elem<T>::VEC;
    const int G = 1 << gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> gshift;
    const int gl = (int)(gtid & (G - 1));
    const int64_t items = (int64_t)kchunks * M;
    for (int64_t item = gtid >> gshift; item < items; item += ngroups) {
        const int64_t i = item % M;
        const int chunk = (int)(item / M);
        const int64_t c0 = ((int64_t)chunk * G + gl) * VEC;
        if (c0 >= D) continue;
        const int32_t beg = rowptr[i], end = rowptr[i + 1];
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        for (int32_t j = beg; j < end; j += U) {
            int64_t c[U];
            float w[U];
            u32x4 rows[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = -1;
                if (j + u < end) {
                    const int32_t e = perm ? perm[j + u] : (j + u);
                    c[u] = col[e];
                    w[u] = value ? Elem<T>::load(value + e) : 1.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c[u] >= 0) rows[u] = *reinterpret_cast<const u32x4*>(mat + c[u] * D + c0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c[u] >= 0) {
                    float f[VEC];
                    Elem<T>::unpack(rows[u], f);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = __fadd_rn(acc[v], __fmul_rn(w[u], f[v]));
                }
            }
        }
        store16<true>(out + i * D + c0, Elem<T>::pack(acc));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void spmm_elems_kernel(const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ perm,
                                                         const int64_t* __restrict__ col, const T* __restrict__ value,
                                                         const T* __restrict__ mat, T* __restrict__ out, int64_t M,
                                                         int64_t D) {
    const int64_t total = M * D;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = o / D, k = o % D;
        float acc = 0.f;
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            const int32_t e = perm ? perm[j] : j;
            const float w = value ? Elem<T>::load(value + e) : 1.f;
            acc = __fadd_rn(acc, __fmul_rn(w, Elem<T>::load(mat + col[e] * D + k)));
        }
        Elem<T>::store(out + o, acc);
    }
}

template <typename T>
int launch(const int32_t* rowptr, const int32_t* perm, const int64_t* col, const void* value, const void* mat,
           void* out, int64_t M, int64_t D, hipStream_t stream) {
    constexpr int VEC = Elem<T>::VEC;
    if (D % VEC == 0 && (uintptr_t)mat % 16 == 0 && (uintptr_t)out % 16 == 0) {
        const int64_t vecs = D / VEC;
        int gshift = 0;
        while ((1 << gshift) < vecs && gshift < 6) ++gshift;
        const int kchunks = (int)gnnops_cdiv(vecs, (int64_t)1 << gshift);
        const int grid = gnnops_grid_cap(gnnops_cdiv((int64_t)kchunks * M, 256 >> gshift), 256 * 64);
        hipLaunchKernelGGL((spmm_rows_kernel<T>), dim3(grid), dim3(256), 0, stream, rowptr, perm, col, (const T*)value,
                           (const T*)mat, (T*)out, M, D, gshift, kchunks);
    } else {
        const int grid = gnnops_grid_cap(gnnops_cdiv(M * D, 256), 256 * 32);
        hipLaunchKernelGGL((spmm_elems_kernel<T>), dim3(grid), dim3(256), 0, stream, rowptr, perm, col, (const T*)value,
                           (const T*)mat, (T*)out, M, D);
    }
    return gnnops_check_launch("spmm");
}

}  // namespace

extern "C" int gnnops_spmm(const int32_t* rowptr, const int32_t* perm, const int64_t* col, const void* value,
                           const void* mat, void* out, int64_t M, int64_t D, int64_t nnz, int dtype,
                           gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(M >= 0 && D >= 0 && nnz >= 0, GNNOPS_EINVAL, "spmm: negative size");
    GNNOPS_REQUIRE(nnz < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "spmm: nnz must be < 2^31");
    if (M * D == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && out && (nnz == 0 || (col && mat)), GNNOPS_EINVAL, "spmm: null pointer");
    switch (dtype) {
        case GNNOPS_F32: return launch<float>(rowptr, perm, col, value, mat, out, M, D, stream);
        case GNNOPS_F16: return launch<__half>(rowptr, perm, col, value, mat, out, M, D, stream);
        case GNNOPS_BF16: return launch<__hip_bfloat16>(rowptr, perm, col, value, mat, out, M, D, stream);
    }
    gnnops_set_error("spmm: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}
