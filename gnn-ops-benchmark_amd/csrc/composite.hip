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
#include "hub.h"

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
                                                                 int kchunks, float param, hub::Ws hw, int hub_on) {
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
        if (hub_on && end - beg > hub::T_HUB) {  // a hub (hub.h): three passes by one lane group would take milliseconds
            if (gl == 0 && chunk == 0) hub::append(hw, (int)n, beg, end, end - beg);
            continue;
        }
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

// ---- hubs (hub.h): a group with more than T_HUB members, piecewise -------------------------------------------------------
// pass 1  hub::hub_partial_kernel<T, MAX or SUM>: the max (mean for std) of every piece -> w.partial
// pass 2  chub_stat2_kernel: every workgroup folds the piece statistics of its hub (in order), then sums exp(x - max) or
//         (x - mean)^2 over its piece -> second partial buffer
// pass 3  chub_finish_kernel: folds both statistics and writes the outputs — per member (softmax / log_softmax, a workgroup
//         per piece) or per group (logsumexp / std, by the workgroup of piece 0)
// Same formulas as the row kernel; the sums over a hub are re-associated piece by piece.
template <int MODE>
__device__ inline float chub_stat1(const hub::Ws& w, int h, int64_t K, int64_t colv) {
    const int np = (w.hubs[4 * h + 2] - w.hubs[4 * h + 1] + hub::PART - 1) / hub::PART;
    const int pb = w.piece_base[h];
    float a = (MODE == MODE_STD) ? 0.f : -__builtin_huge_valf();
    for (int p = 0; p < np; ++p) {
        const float f = w.partial[(int64_t)(pb + p) * K + colv];
        if (MODE == MODE_STD) a += f; else a = f > a ? f : a;
    }
    if (MODE == MODE_STD) a = a / (float)w.hubs[4 * h + 3];
    return a;
}

template <typename T, int MODE>
__global__ __launch_bounds__(hub::THREADS) void chub_stat2_kernel(const T* __restrict__ src, const int32_t* __restrict__ perm,
                                                                  hub::Ws w, float* __restrict__ partial2, int64_t K,
                                                                  int gshift, int kchunks) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ int32_t s_match[hub::PART];
    __shared__ float s_part[hub::THREADS * VEC];
    const int tid = threadIdx.x;
    const int G = 1 << gshift, gl = tid & (G - 1), gi = tid >> gshift, groups = hub::THREADS >> gshift;
    const float NEG_INF = -__builtin_huge_valf();
    int npieces = w.counters[1];
    if (npieces > w.cap_p) npieces = w.cap_p;
    for (int q = blockIdx.x; q < npieces; q += gridDim.x) {
        const int h = w.pieces[2 * q];
        if (h < 0) continue;
        const int pno = w.pieces[2 * q + 1];
        const int beg = w.hubs[4 * h + 1], end = w.hubs[4 * h + 2];
        const int pb = beg + pno * hub::PART;
        const int n = (end - pb < hub::PART) ? end - pb : hub::PART;
        __syncthreads();
        for (int i = tid; i < n; i += hub::THREADS) s_match[i] = perm ? perm[pb + i] : pb + i;
        __syncthreads();
        const int part = (n + groups - 1) / groups;
        const int jb = gi * part, je = (jb + part < n) ? jb + part : n;
        for (int chunk = 0; chunk < kchunks; ++chunk) {
            const int64_t col = ((int64_t)chunk * G + gl) * VEC;
            float a1[VEC], acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) { acc[v] = 0.f; a1[v] = 0.f; }
            if (col < K) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) a1[v] = chub_stat1<MODE>(w, h, K, col + v);
                for (int j = jb; j < je; j += U) {
                    int32_t e[U];
                    u32x4 rows[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) e[u] = (j + u < je) ? s_match[j + u] : -1;
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (e[u] >= 0) rows[u] = *reinterpret_cast<const u32x4*>(src + (int64_t)e[u] * K + col);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (e[u] < 0) continue;
                        float f[VEC];
                        Elem<T>::unpack(rows[u], f);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            float d = f[v] - a1[v];
                            if (MODE == MODE_STD) {
                                acc[v] += d * d;
                            } else {
                                if (d != d) d = NEG_INF;
                                acc[v] += expf(d);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) s_part[tid * VEC + v] = acc[v];
            __syncthreads();
            if (gi == 0 && col < K) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float a = s_part[gl * VEC + v];
                    for (int g = 1; g < groups; ++g) a += s_part[(g * G + gl) * VEC + v];
                    partial2[(int64_t)q * K + col + v] = a;
                }
            }
            __syncthreads();
        }
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(hub::THREADS) void chub_finish_kernel(const T* __restrict__ src, const int32_t* __restrict__ perm,
                                                                   T* __restrict__ out, hub::Ws w,
                                                                   const float* __restrict__ partial2, int64_t K, int gshift,
                                                                   int kchunks, float param) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ int32_t s_match[hub::PART];
    const int tid = threadIdx.x;
    const int G = 1 << gshift, gl = tid & (G - 1), gi = tid >> gshift, groups = hub::THREADS >> gshift;
    const float NEG_INF = -__builtin_huge_valf();
    int npieces = w.counters[1];
    if (npieces > w.cap_p) npieces = w.cap_p;
    for (int q = blockIdx.x; q < npieces; q += gridDim.x) {
        const int h = w.pieces[2 * q];
        if (h < 0) continue;
        const int pno = w.pieces[2 * q + 1];
        const int dst = w.hubs[4 * h], beg = w.hubs[4 * h + 1], end = w.hubs[4 * h + 2], cnt = w.hubs[4 * h + 3];
        constexpr bool PER_MEMBER = (MODE == MODE_SOFTMAX || MODE == MODE_LOG_SOFTMAX);
        if (!PER_MEMBER && pno != 0) continue;   // one output row per group: piece 0's workgroup writes it
        const int pb = beg + pno * hub::PART;
        const int n = (end - pb < hub::PART) ? end - pb : hub::PART;
        const int np = (end - beg + hub::PART - 1) / hub::PART;
        const int pbase = w.piece_base[h];
        __syncthreads();
        if (PER_MEMBER) {
            for (int i = tid; i < n; i += hub::THREADS) s_match[i] = perm ? perm[pb + i] : pb + i;
        }
        __syncthreads();
        for (int chunk = 0; chunk < kchunks; ++chunk) {
            const int64_t col = ((int64_t)chunk * G + gl) * VEC;
            if (col >= K) continue;
            float a1[VEC], a2[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                a1[v] = chub_stat1<MODE>(w, h, K, col + v);
                float t = 0.f;
                for (int p = 0; p < np; ++p) t += partial2[(int64_t)(pbase + p) * K + col + v];
                a2[v] = t;
            }
            if constexpr (!PER_MEMBER) {
                if (gi == 0) {
                    float r[VEC];
                    int32_t cc = (MODE == MODE_STD && param != 0.f) ? cnt - 1 : cnt;
                    if (cc < 1) cc = 1;
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        r[v] = (MODE == MODE_STD) ? sqrtf(a2[v] / ((float)cc + 1e-6f)) : a1[v] + logf(a2[v] + param);
                    store16<true>(out + (int64_t)dst * K + col, Elem<T>::pack(r));
                }
            } else {
                float lg[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) lg[v] = logf(a2[v] + param);
                for (int j0 = gi; j0 < n; j0 += groups * U) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int j = j0 + u * groups;
                        if (j >= n) continue;
                        const int64_t e = s_match[j];
                        float f[VEC];
                        Elem<T>::unpack(*reinterpret_cast<const u32x4*>(src + e * K + col), f);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            float d = f[v] - a1[v];
                            if (d != d) d = NEG_INF;
                            f[v] = (MODE == MODE_SOFTMAX) ? expf(d) / a2[v] : d - lg[v];
                        }
                        store16<true>(out + e * K + col, Elem<T>::pack(f));
                    }
                }
            }
        }
    }
}

template <typename T, int MODE>
void launch_chub(const T* src, const int32_t* perm, T* out, const hub::Ws& w, int64_t E, int64_t K, int gshift, int kchunks,
                 float param, hipStream_t stream) {
    const int ga = w.cap_p < 2048 ? w.cap_p : 2048;
    float* partial2 = reinterpret_cast<float*>(w.parg);
    hub::Ws w1 = w;
    w1.parg = nullptr;  // pass 1 keeps no positions: the second buffer belongs to pass 2
    if (MODE == MODE_STD)
        hipLaunchKernelGGL((hub::hub_partial_kernel<T, GNNOPS_SUM, false>), dim3(ga), dim3(hub::THREADS), 0, stream, src, perm,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr, w1, E, K, gshift, kchunks);
    else
        hipLaunchKernelGGL((hub::hub_partial_kernel<T, GNNOPS_MAX, false>), dim3(ga), dim3(hub::THREADS), 0, stream, src, perm,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr, w1, E, K, gshift, kchunks);
    hipLaunchKernelGGL((chub_stat2_kernel<T, MODE>), dim3(ga), dim3(hub::THREADS), 0, stream, src, perm, w1, partial2, K, gshift,
                       kchunks);
    hipLaunchKernelGGL((chub_finish_kernel<T, MODE>), dim3(ga), dim3(hub::THREADS), 0, stream, src, perm, out, w1, partial2, K,
                       gshift, kchunks, param);
}

template <typename T>
int dispatch(int mode, const void* src, const int32_t* rowptr, const int32_t* perm, void* out, int64_t B, int64_t E,
             int64_t K, int64_t N, float param, hipStream_t stream, void* hub_ws, size_t hub_ws_bytes) {
    constexpr int VEC = Elem<T>::VEC;
    if (K % VEC == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)out % 16 == 0) {
        const int64_t vecs = K / VEC;
        int gshift = 0;
        while ((1 << gshift) < vecs && gshift < 6) ++gshift;
        const int kchunks = (int)gnnops_cdiv(vecs, (int64_t)1 << gshift);
        const int rgrid = gnnops_grid_cap(gnnops_cdiv(B * kchunks * N, 256 >> gshift), 256 * 64);
        hub::Ws hw{};
        int hub_on = 0;
        if (hub_ws && B == 1 && E > hub::T_HUB) {
            const hub::Layout hl = hub::layout(E, K, true);   // two partial buffers: the statistics of passes 1 and 2
            if (hub_ws_bytes >= hl.total) {
                hw = hub::make_ws(hub_ws, hl, E, true);
                if (gnnops_memset_async(hw.counters, 0, 8, stream) != hipSuccess) return gnnops_check_launch("hub memset");
                hub_on = 1;
            }
        }
#define LAUNCH_ROWS(M)                                                                                                  \
    hipLaunchKernelGGL((seg_composite_rows_kernel<T, M>), dim3(rgrid), dim3(256), 0, stream, (const T*)src, rowptr, perm,  \
                       (T*)out, B, E, K, N, gshift, kchunks, param, hw, hub_on);                                           \
    if (hub_on) launch_chub<T, M>((const T*)src, perm, (T*)out, hw, E, K, gshift, kchunks, param, stream)
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
    return gnnops_segment_composite_hubs(src, rowptr, perm, out, B, E, K, N, dtype, mode, param, nullptr, 0, s);
}

// The same with groups of more than 8192 members set aside and processed piecewise by whole workgroups (hub.h):
// hub_workspace = gnnops_hub_workspace_bytes(E, K, GNNOPS_MIN) bytes (two partial buffers), or NULL.
extern "C" int gnnops_segment_composite_hubs(const void* src, const int32_t* rowptr, const int32_t* perm, void* out, int64_t B,
                                             int64_t E, int64_t K, int64_t N, int dtype, int mode, double param,
                                             void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && E >= 0 && K >= 0 && N >= 0, GNNOPS_EINVAL, "segment_composite: negative size");
    GNNOPS_REQUIRE(E < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "segment_composite: E must be < 2^31");
    if (B * N * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && out && (E == 0 || src), GNNOPS_EINVAL, "segment_composite: null pointer");
    switch (dtype) {
        case GNNOPS_F32: return dispatch<float>(mode, src, rowptr, perm, out, B, E, K, N, (float)param, stream, hub_workspace, hub_workspace_bytes);
        case GNNOPS_F16: return dispatch<__half>(mode, src, rowptr, perm, out, B, E, K, N, (float)param, stream, hub_workspace, hub_workspace_bytes);
        case GNNOPS_BF16: return dispatch<__hip_bfloat16>(mode, src, rowptr, perm, out, B, E, K, N, (float)param, stream, hub_workspace, hub_workspace_bytes);
    }
    gnnops_set_error("segment_composite: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}
