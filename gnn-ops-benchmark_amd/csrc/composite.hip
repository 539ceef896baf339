// composite.hip — per-group softmax / log_softmax / logsumexp / std over a plan (or a CSR rowptr):
// torch_scatter.composite.{scatter_softmax, scatter_log_softmax, scatter_logsumexp, scatter_std} — the ops PyG's
// attention layers (GATv2 softmax, graph_benchmark/models/ptg_models.py:238-258) and PNA aggregators
// (ptg_models.py:62-78) put on the reference's OpProfiler path (ops.txt:44-50).
//
// One thread per output column (b, n, k), coalesced along k; a group's rows are walked two or three times
// (max / mean, then the shifted sum, then — softmax only — one store per source row); the re-reads hit L1/L2
// for GNN-sized groups. fp32 arithmetic, sequential over the group in plan order, one rounding on store.
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

template <typename T>
int dispatch(int mode, const void* src, const int32_t* rowptr, const int32_t* perm, void* out, int64_t B, int64_t E,
             int64_t K, int64_t N, float param, hipStream_t stream) {
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
