// composite.hip — per-group softmax / log_softmax / logsumexp / std over a plan (or a CSR rowptr):
// torch_scatter.composite.{scatter_softmax, scatter_log_softmax, scatter_logsumexp, scatter_std} — the ops PyG's
// attention layers (GATv2 softmax, graph_benchmark/models/ptg_models.py:238-258) and PNA aggregators
// (ptg_models.py:62-78) put on the reference's OpProfiler path (ops.txt:44-50).
//
// Row form: lane group per destination with 16-B lane accesses; groups of <= 8 rows are read once and held in
// registers across the passes (max / mean, shifted sum, and — softmax only — one store per source row). Element
// form (any K / alignment): one thread per output column, coalesced along k, rows re-read per pass.
// fp32 arithmetic, sequential over the group in plan order, one rounding on store.
//   softmax      out[b,e,k] = exp(x - max_n) / sum_n exp(x - max_n)
//   log_softmax  out[b,e,k] = (x - max_n) - log(sum_n + eps)
//   logsumexp    out[b,n,k] = max_n + log(sum_n + eps)          (empty group: max := 0, sum = 0)
//   std          out[b,n,k] = sqrt( sum_n (x - mean_n)^2 / (cnt' + 1e-6) ),  cnt' = unbiased ? max(cnt-1,1) : max(cnt,1)
#include "common.h"

namespace {

enum { MODE_SOFTMAX = 0, MODE_LOG_SOFTMAX = 1, MODE_LOGSUMEXP = 2, MODE_STD = 3 };

template <typename T, int MODE>
__global__ __launch_bounds__(256) void seg_composite_kernel(const T* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ perm, T* __restrict__ out,
                                                            int64_t B, int64_t E, int64_t K, int64_t N, float param) {
    const int64_t total = B * N * K;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = o % K;
        const int64_t bn = o / K;
        const int64_t n = bn % N;
        const int64_t b = bn / N;
        const int32_t beg = rowptr[n], end = rowptr[n + 1];
        const T* srcb = src + (b * E) * K + k;
        auto row = [&](int32_t j) -> int64_t { return (int64_t)(perm ? perm[j] : j) * K; };
        if constexpr (MODE == MODE_STD) {
            float sum = 0.f;
            for (int32_t j = beg; j < end; ++j) sum += Elem<T>::load(srcb + row(j));
            const int32_t cnt = end - beg;
            const float mean = sum / (float)(cnt < 1 ? 1 : cnt);
            float var = 0.f;
            for (int32_t j = beg; j < end; ++j) {
                const float d = Elem<T>::load(srcb + row(j)) - mean;
                var += d * d;
            }
            int32_t c = (param != 0.f) ? cnt - 1 : cnt;  // param != 0: unbiased
            if (c < 1) c = 1;
            Elem<T>::store(out + o, sqrtf(var / ((float)c + 1e-6f)));
        } else {
            float m = -__builtin_huge_valf();
            for (int32_t j = beg; j < end; ++j) {
                const float x = Elem<T>::load(srcb + row(j));
                m = x > m ? x : m;
            }
            if (beg == end) m = 0.f;  // torch_scatter: scatter_max leaves empty groups at 0
            float s = 0.f;
            for (int32_t j = beg; j < end; ++j) {
                float r = Elem<T>::load(srcb + row(j)) - m;
                if (r != r) r = -__builtin_huge_valf();  // (-inf) - (-inf): treated as -inf, as upstream does
                s += expf(r);
            }
            if constexpr (MODE == MODE_LOGSUMEXP) {
                Elem<T>::store(out + o, m + logf(s + param));
            } else {
                const float lg = logf(s + param);
                for (int32_t j = beg; j < end; ++j) {
                    const int64_t r = row(j);
                    float x = Elem<T>::load(srcb + r) - m;
                    if (x != x) x = -__builtin_huge_valf();
                    const float y = (MODE == MODE_SOFTMAX) ? expf(x) / s : x - lg;
                    Elem<T>::store(out + (b * E) * K + k + r, y);
                }
            }
        }
    }
}

// Row form (K % VEC == 0, 16-B aligned): one lane group per destination and 1-KiB column chunk, 16-B lane
// accesses, U source rows in flight. A group of at most U rows — the common case at GNN degrees — is read ONCE and
// kept in registers across the two or three logical passes; longer groups re-read their rows chunk by chunk.
constexpr int U = 8;

template <typename T, int MODE>
__global__ __launch_bounds__(256) void seg_composite_rows_kernel(const T* __restrict__ src,
                                                                 const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ perm, T* __restrict__ out,
                                                                 int64_t B, int64_t E, int64_t K, int64_t N, int gshift,
                                                                 int kchunks, float param) {
    constexpr int VEC = Elem<T>::VEC;
    const int G = 1 << gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> gshift;
    const int gl = (int)(gtid & (G - 1));
    const int64_t items = B * (int64_t)kchunks * N;
    const float NEG_INF = -__builtin_huge_valf();
    for (int64_t item = gtid >> gshift; item < items; item += ngroups) {
        const int64_t n = item % N;
        const int64_t bc = item / N;
        const int chunk = (int)(bc % kchunks);
        const int64_t b = bc / kchunks;
        const int64_t col = ((int64_t)chunk * G + gl) * VEC;
        if (col >= K) continue;
        const int32_t beg = rowptr[n], end = rowptr[n + 1];
        const bool inreg = (end - beg) <= U;
        const T* srcb = src + (b * E) * K + col;
        T* outb = out + (b * E) * K + col;
        int32_t e[U];
        u32x4 rows[U];
        auto load_chunk = [&](int32_t j) {
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = (j + u < end) ? (perm ? perm[j + u] : j + u) : -1;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (e[u] >= 0) rows[u] = *reinterpret_cast<const u32x4*>(srcb + (int64_t)e[u] * K);
        };
        float a1[VEC], a2[VEC];  // pass-1 statistic (max or mean) and pass-2 statistic (sum of exp or of squares)
#pragma unroll
        for (int v = 0; v < VEC; ++v) { a1[v] = (MODE == MODE_STD) ? 0.f : NEG_INF; a2[v] = 0.f; }

        for (int32_t j = beg; j < end; j += U) {
            load_chunk(j);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (e[u] < 0) continue;
                float f[VEC];
                Elem<T>::unpack(rows[u], f);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    if (MODE == MODE_STD) a1[v] += f[v];
                    else a1[v] = f[v] > a1[v] ? f[v] : a1[v];
                }
            }
        }
        const int32_t cnt = end - beg;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            if (MODE == MODE_STD) a1[v] = a1[v] / (float)(cnt < 1 ? 1 : cnt);
            else if (cnt == 0) a1[v] = 0.f;
        }
        for (int32_t j = beg; j < end; j += U) {
            if (!inreg) load_chunk(j);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (e[u] < 0) continue;
                float f[VEC];
                Elem<T>::unpack(rows[u], f);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float d = f[v] - a1[v];
                    if (MODE == MODE_STD) {
                        a2[v] += d * d;
                    } else {
                        if (d != d) d = NEG_INF;
                        a2[v] += expf(d);
                    }
                }
            }
        }
        if constexpr (MODE == MODE_STD || MODE == MODE_LOGSUMEXP) {
            float r[VEC];
            int32_t c = (MODE == MODE_STD && param != 0.f) ? cnt - 1 : cnt;
            if (c < 1) c = 1;
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                r[v] = (MODE == MODE_STD) ? sqrtf(a2[v] / ((float)c + 1e-6f)) : a1[v] + logf(a2[v] + param);
            store16<true>(out + (b * N + n) * K + col, Elem<T>::pack(r));
        } else {
            float lg[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) lg[v] = logf(a2[v] + param);
            for (int32_t j = beg; j < end; j += U) {
                if (!inreg) load_chunk(j);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (e[u] < 0) continue;
                    float f[VEC];
                    Elem<T>::unpack(rows[u], f);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float d = f[v] - a1[v];
                        if (d != d) d = NEG_INF;
                        f[v] = (MODE == MODE_SOFTMAX) ? expf(d) / a2[v] : d - lg[v];
                    }
                    store16<true>(outb + (int64_t)e[u] * K, Elem<T>::pack(f));
                }
            }
        }
    }
}

template <typename T>
int dispatch(int mode, const void* src, const int32_t* rowptr, const int32_t* perm, void* out, int64_t B, int64_t E,
             int64_t K, int64_t N, float param, hipStream_t stream) {
    constexpr int VEC = Elem<T>::VEC;
    if (K % VEC == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)out % 16 == 0) {
        const int64_t vecs = K / VEC;
        int gshift = 0;
        while ((1 << gshift) < vecs && gshift < 6) ++gshift;
        const int kchunks = (int)gnnops_cdiv(vecs, (int64_t)1 << gshift);
        const int rgrid = gnnops_grid_cap(gnnops_cdiv(B * kchunks * N, 256 >> gshift), 256 * 64);
#define LAUNCH_ROWS(M)                                                                                                  \
    hipLaunchKernelGGL((seg_composite_rows_kernel<T, M>), dim3(rgrid), dim3(256), 0, stream, (const T*)src, rowptr, perm,  \
                       (T*)out, B, E, K, N, gshift, kchunks, param)
        switch (mode) {
            case MODE_SOFTMAX: LAUNCH_ROWS(MODE_SOFTMAX); break;
            case MODE_LOG_SOFTMAX: LAUNCH_ROWS(MODE_LOG_SOFTMAX); break;
            case MODE_LOGSUMEXP: LAUNCH_ROWS(MODE_LOGSUMEXP); break;
            case MODE_STD: LAUNCH_ROWS(MODE_STD); break;
            default: gnnops_set_error("segment_composite: unknown mode %d", mode); return GNNOPS_EINVAL;
        }
#undef LAUNCH_ROWS
        return gnnops_check_launch("segment_composite");
    }
    const int grid = gnnops_grid_cap(gnnops_cdiv(B * N * K, 256), 256 * 32);
#define LAUNCH(M)                                                                                              \
    hipLaunchKernelGGL((seg_composite_kernel<T, M>), dim3(grid), dim3(256), 0, stream, (const T*)src, rowptr, perm, \
                       (T*)out, B, E, K, N, param)
    switch (mode) {
        case MODE_SOFTMAX: LAUNCH(MODE_SOFTMAX); break;
        case MODE_LOG_SOFTMAX: LAUNCH(MODE_LOG_SOFTMAX); break;
        case MODE_LOGSUMEXP: LAUNCH(MODE_LOGSUMEXP); break;
        case MODE_STD: LAUNCH(MODE_STD); break;
        default: gnnops_set_error("segment_composite: unknown mode %d", mode); return GNNOPS_EINVAL;
    }
#undef LAUNCH
    return gnnops_check_launch("segment_composite");
}

}  // namespace

extern "C" int gnnops_segment_composite(const void* src, const int32_t* rowptr, const int32_t* perm, void* out, int64_t B,
                                        int64_t E, int64_t K, int64_t N, int dtype, int mode, double param,
                                        gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && E >= 0 && K >= 0 && N >= 0, GNNOPS_EINVAL, "segment_composite: negative size");
    GNNOPS_REQUIRE(E < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "segment_composite: E must be < 2^31");
    if (B * N * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && out && (E == 0 || src), GNNOPS_EINVAL, "segment_composite: null pointer");
    switch (dtype) {
        case GNNOPS_F32: return dispatch<float>(mode, src, rowptr, perm, out, B, E, K, N, (float)param, stream);
        case GNNOPS_F16: return dispatch<__half>(mode, src, rowptr, perm, out, B, E, K, N, (float)param, stream);
        case GNNOPS_BF16: return dispatch<__hip_bfloat16>(mode, src, rowptr, perm, out, B, E, K, N, (float)param, stream);
    }
    gnnops_set_error("segment_composite: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}
