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
#include "hub.h"
#include <stdlib.h>

namespace {

constexpr int U = 8;

template <typename T, bool NT>
__global__ __launch_bounds__(256) void spmm_rows_kernel(const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ perm,
                                                        const int64_t* __restrict__ col, const T* __restrict__ value,
                                                        const T* __restrict__ mat, T* __restrict__ out, int64_t M,
                                                        int64_t D, int gshift, int kchunks, hub::Ws hw, int hub_on) {
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
        if (hub_on && end - beg > hub::T_HUB) {  // a hub row (hub.h): multiplied out piecewise by the hub pass
            if (gl == 0 && chunk == 0) hub::append(hw, (int)i, beg, end, end - beg);
            continue;
        }
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        for (int32_t j = beg; j < end; j += U) {
            int64_t c[U];
            T wraw[U];      // the value as STORED: converting it where it is loaded would put a wait for the load right there, and
            u32x4 rows[U];  // the U (column id, value) pairs of a step would be fetched one memory latency after the other
            // three phases, each with all its loads in flight together: positions (plan order only), then (column id, value)
            // pairs, then the gathered rows. Interleaved per nonzero, every position -> column -> row chain waited on its own.
            int32_t e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = (j + u < end) ? (perm ? perm[j + u] : j + u) : -1;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = -1;
                if (e[u] >= 0) {
                    c[u] = col[e[u]];
                    if (value) wraw[u] = value[e[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c[u] >= 0) rows[u] = load16<NT>(mat + c[u] * D + c0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c[u] >= 0) {
                    float f[VEC];
                    Elem<T>::unpack(rows[u], f);
                    const float w = value ? Elem<T>::load(&wraw[u]) : 1.f;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = __fadd_rn(acc[v], __fmul_rn(w, f[v]));
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

// Wide dense operands whose rows are not 16-B multiples (the reference's (L, L) fp32 shapes, L = 7071): a workgroup takes
// (output row i, a run of 256 * UC columns); the row's column ids and values are the same for every lane (scalar loads),
// each lane keeps UC accumulators and sweeps the nonzeros two at a time, so 2 * UC coalesced element loads are in flight.
// Same arithmetic and order as the row kernel; no per-element division (spmm_elems_kernel pays one per output element).
template <typename T>
__global__ __launch_bounds__(256) void spmm_widerows_kernel(const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ perm,
                                                            const int64_t* __restrict__ col, const T* __restrict__ value,
                                                            const T* __restrict__ mat, T* __restrict__ out, int64_t M,
                                                            int64_t D, int chunks) {
    constexpr int UC = 4;
    const int64_t items = M * chunks;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t i = item / chunks;
        const int64_t c0 = (item - i * chunks) * (256 * UC) + threadIdx.x;
        const int32_t beg = rowptr[i], end = rowptr[i + 1];
        float acc[UC];
#pragma unroll
        for (int u = 0; u < UC; ++u) acc[u] = 0.f;
        for (int32_t j = beg; j < end; j += 2) {
            const bool two = j + 1 < end;
            const int32_t e0 = perm ? perm[j] : j, e1 = two ? (perm ? perm[j + 1] : j + 1) : e0;
            const int64_t r0 = col[e0], r1 = col[e1];
            const float w0 = value ? Elem<T>::load(value + e0) : 1.f;
            const float w1 = two ? (value ? Elem<T>::load(value + e1) : 1.f) : 0.f;
            float x0[UC], x1[UC];
#pragma unroll
            for (int u = 0; u < UC; ++u) {
                const int64_t k = c0 + 256 * u;
                const int64_t kc = k < D ? k : D - 1;
                x0[u] = Elem<T>::load(mat + r0 * D + kc);
                x1[u] = Elem<T>::load(mat + r1 * D + kc);
            }
#pragma unroll
            for (int u = 0; u < UC; ++u) {
                acc[u] = __fadd_rn(acc[u], __fmul_rn(w0, x0[u]));
                if (two) acc[u] = __fadd_rn(acc[u], __fmul_rn(w1, x1[u]));
            }
        }
#pragma unroll
        for (int u = 0; u < UC; ++u) {
            const int64_t k = c0 + 256 * u;
            if (k < D) Elem<T>::store(out + i * D + k, acc[u]);
        }
    }
}

// CSR materialisation of a plan-ordered COO operand: out[j] = in[perm[j]] for the column ids (8 B) or the values.
template <typename U>
__global__ void permute_kernel(const U* __restrict__ in, const int32_t* __restrict__ perm, U* __restrict__ out, int64_t n) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        out[j] = in[perm[j]];
}

template <typename T>
int launch(const int32_t* rowptr, const int32_t* perm, const int64_t* col, const void* value, const void* mat,
           void* out, int64_t M, int64_t D, int64_t mat_rows, hipStream_t stream, int64_t nnz, void* hub_ws,
           size_t hub_ws_bytes) {
    constexpr int VEC = Elem<T>::VEC;
    if (D % VEC == 0 && (uintptr_t)mat % 16 == 0 && (uintptr_t)out % 16 == 0) {
        const int64_t vecs = D / VEC;
        int gshift = 0;
        while ((1 << gshift) < vecs && gshift < 6) ++gshift;
        // A/B switch for tools/time_spmm.py: narrower lane groups = column slices of the dense operand, swept slice by slice
        // (items are chunk-major), so that the gathered slice of `mat` can stay in the Infinity Cache
        if (const char* gs = getenv("GNNOPS_SPMM_GSHIFT")) {
            const int g = atoi(gs);
            if (g >= 0 && g < gshift) gshift = g;
        }
        const int kchunks = (int)gnnops_cdiv(vecs, (int64_t)1 << gshift);
        const int grid = gnnops_grid_cap(gnnops_cdiv((int64_t)kchunks * M, 256 >> gshift), 256 * 64);
        // a dense operand far beyond the 256 MiB Infinity Cache is streamed nontemporally (its rows would only evict
        // the CSR arrays); a smaller one keeps normal caching so repeated rows hit on-die
        const bool nt = (size_t)mat_rows * (size_t)D * sizeof(T) > ((size_t)2 << 30);
        hub::Ws hw{};
        int hub_on = 0;
        if (hub_ws && nnz > hub::T_HUB) {
            const hub::Layout hl = hub::layout(nnz, D, false);
            if (hub_ws_bytes >= hl.total) {
                hw = hub::make_ws(hub_ws, hl, nnz, false);
                if (gnnops_memset_async(hw.counters, 0, 8, stream) != hipSuccess) return gnnops_check_launch("hub memset");
                hub_on = 1;
            }
        }
        if (nt)
            hipLaunchKernelGGL((spmm_rows_kernel<T, true>), dim3(grid), dim3(256), 0, stream, rowptr, perm, col,
                               (const T*)value, (const T*)mat, (T*)out, M, D, gshift, kchunks, hw, hub_on);
        else
            hipLaunchKernelGGL((spmm_rows_kernel<T, false>), dim3(grid), dim3(256), 0, stream, rowptr, perm, col,
                               (const T*)value, (const T*)mat, (T*)out, M, D, gshift, kchunks, hw, hub_on);
        if (hub_on)
            hub::launch_spmm_pass<T>(perm, col, (const T*)value, (const T*)mat, (T*)out, hw, nnz, D, gshift, kchunks, stream);
    } else if (D >= 256) {
        const int chunks = (int)gnnops_cdiv(D, 256 * 4);
        const int grid = gnnops_grid_cap(M * chunks, 256 * 32);
        hipLaunchKernelGGL((spmm_widerows_kernel<T>), dim3(grid), dim3(256), 0, stream, rowptr, perm, col, (const T*)value,
                           (const T*)mat, (T*)out, M, D, chunks);
    } else {
        const int grid = gnnops_grid_cap(gnnops_cdiv(M * D, 256), 256 * 32);
        hipLaunchKernelGGL((spmm_elems_kernel<T>), dim3(grid), dim3(256), 0, stream, rowptr, perm, col, (const T*)value,
                           (const T*)mat, (T*)out, M, D);
    }
    return gnnops_check_launch("spmm");
}

}  // namespace

extern "C" int gnnops_spmm(const int32_t* rowptr, const int32_t* perm, const int64_t* col, const void* value,
                           const void* mat, void* out, int64_t M, int64_t D, int64_t nnz, int64_t mat_rows, int dtype,
                           gnnops_stream_t s) {
    return gnnops_spmm_hubs(rowptr, perm, col, value, mat, out, M, D, nnz, mat_rows, dtype, nullptr, 0, s);
}

// The same with rows of more than 8192 nonzeros set aside and multiplied out piecewise by whole workgroups (hub.h):
// hub_workspace = gnnops_hub_workspace_bytes(nnz, D, 0) bytes, or NULL.
extern "C" int gnnops_spmm_hubs(const int32_t* rowptr, const int32_t* perm, const int64_t* col, const void* value,
                                const void* mat, void* out, int64_t M, int64_t D, int64_t nnz, int64_t mat_rows, int dtype,
                                void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(M >= 0 && D >= 0 && nnz >= 0, GNNOPS_EINVAL, "spmm: negative size");
    GNNOPS_REQUIRE(nnz < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "spmm: nnz must be < 2^31");
    if (M * D == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && out && (nnz == 0 || (col && mat)), GNNOPS_EINVAL, "spmm: null pointer");
    switch (dtype) {
        case GNNOPS_F32: return launch<float>(rowptr, perm, col, value, mat, out, M, D, mat_rows, stream, nnz, hub_workspace, hub_workspace_bytes);
        case GNNOPS_F16: return launch<__half>(rowptr, perm, col, value, mat, out, M, D, mat_rows, stream, nnz, hub_workspace, hub_workspace_bytes);
        case GNNOPS_BF16: return launch<__hip_bfloat16>(rowptr, perm, col, value, mat, out, M, D, mat_rows, stream, nnz, hub_workspace, hub_workspace_bytes);
    }
    gnnops_set_error("spmm: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}

// out[j] = in[perm[j]], elements of 2, 4 or 8 bytes: turns (plan, COO columns / values) into CSR arrays once, so the
// row-split kernel streams them instead of chasing perm -> col for every nonzero.
extern "C" int gnnops_permute(const void* in, const int32_t* perm, void* out, int64_t n, int elem_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(n >= 0, GNNOPS_EINVAL, "permute: negative size");
    GNNOPS_REQUIRE(elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8, GNNOPS_EUNSUPPORTED, "permute: elem_bytes %d", elem_bytes);
    if (n == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(in && perm && out, GNNOPS_EINVAL, "permute: null pointer");
    const int grid = gnnops_grid_cap(gnnops_cdiv(n, 256), 256 * 32);
    if (elem_bytes == 8)
        hipLaunchKernelGGL(permute_kernel<uint64_t>, dim3(grid), dim3(256), 0, stream, (const uint64_t*)in, perm, (uint64_t*)out, n);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL(permute_kernel<uint32_t>, dim3(grid), dim3(256), 0, stream, (const uint32_t*)in, perm, (uint32_t*)out, n);
    else
        hipLaunchKernelGGL(permute_kernel<uint16_t>, dim3(grid), dim3(256), 0, stream, (const uint16_t*)in, perm, (uint16_t*)out, n);
    return gnnops_check_launch("permute");
}
