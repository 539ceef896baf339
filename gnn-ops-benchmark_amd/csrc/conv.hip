// conv.hip — message + aggregate of one message-passing layer in ONE pass over the destination-sorted edge list
// (the single-layer forward passes the reference times, app_bm/benchmark_convs.py:146-246,
// app_bm/groq_script.py:91-109 = CGConv.forward / CGConv.message).
//
// A PyG layer's propagate() is  gather x_j (and x_i) per edge -> message(x_i, x_j, e_ij) -> scatter-reduce by destination:
// three [E, .] tensors written and read back. Every message the reference's five layers use is ELEMENTWISE in per-node
// projections once the linear maps are pulled out of the edge loop (z = [x_i, x_j, e] => z W = x_i W_i + x_j W_j + e W_e:
// two dense [N, D] x [D, .] products instead of a per-edge [E, 2D] x [2D, .] one — 5x fewer flops at 5 edges per node, and
// on MFMA through gemm.hip), so the edge loop that is left is HBM-bound:
//
//   out[i, a-th block] = AGGR_a over edges (j -> i) of  f( p[i, :], q[j, :], w[e, :] )
//
//   f = COPY    q                                      (GIN / SAGE neighbour sum / mean)
//       ADD     p + q (+ w)                            (PNAConv message with one pre-layer: Linear([x_i, x_j (, e)]))
//       CGCONV  sigmoid(p_f + q_f (+ w_f)) * softplus(p_s + q_s (+ w_s))    rows are [f part | s part], 2K wide
//       FILM    relu(gamma_i * q + beta_i)             p rows are [beta | gamma]  (FiLMConv.message)
//   AGGR = sum, mean, min, max, std — any ordered subset in the same pass (PNAConv: mean, min, max, std), each optionally
//   multiplied by PNA's degree scalers, written side by side into a row of pitch `ldo` (so the layer's torch.cat never runs).
//
// One lane group per (destination row, 16-B column chunk) exactly as segment.hip / spmm.hip: each gathered row is read once,
// each output row stored once, messages live in registers, fp32 arithmetic, ONE rounding on store. Algorithmic bytes per
// launch: E * (q row + 8 B column id (+ w row)) + N * (p row + out row) + 4 (N + 1).
#include "common.h"

namespace {

enum { F_COPY = 0, F_ADD = 1, F_CGCONV = 2, F_FILM = 3 };
enum { A_SUM = 0, A_MEAN = 1, A_MIN = 2, A_MAX = 3, A_STD = 4 };
enum { S_IDENTITY = 0, S_AMPLIFICATION = 1, S_ATTENUATION = 2, S_LINEAR = 3, S_INVERSE_LINEAR = 4 };

struct Args {
    const void *q, *p, *w, *add;
    const int32_t *rowptr, *perm;
    const int64_t* col;
    void* out;
    int64_t N, K, ldq, ldp, ldw, ldadd, ldo;
    int n_aggr, aggr[5], n_scal, scal[5];
    float avg_log, avg_lin;
    int gshift, kchunks;
};

template <int F> struct Parts;   // K-wide parts per row of q / p / w
template <> struct Parts<F_COPY> { static constexpr int Q = 1, P = 0, W = 0; };
template <> struct Parts<F_ADD> { static constexpr int Q = 1, P = 1, W = 1; };
template <> struct Parts<F_CGCONV> { static constexpr int Q = 2, P = 2, W = 2; };
template <> struct Parts<F_FILM> { static constexpr int Q = 1, P = 2, W = 0; };

template <typename T, int VEC, bool NT>
__device__ inline void load_vec(const T* p, float* f) {
    if constexpr (VEC == 1) {
        f[0] = Elem<T>::load(p);
    } else {
        Elem<T>::unpack(load16<NT>(p), f);
    }
}
template <typename T, int VEC>
__device__ inline void store_vec(T* p, const float* f) {
    if constexpr (VEC == 1) {
        Elem<T>::store(p, f[0]);
    } else {
        store16<true>(p, Elem<T>::pack(f));
    }
}

__device__ inline float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }
// torch.nn.functional.softplus (beta 1, threshold 20): x above the threshold, log1p(exp(x)) below
__device__ inline float softplus_f(float x) {
    if (x > 20.f) return x;
    const float t = __expf(-fabsf(x));
    const float l = t < 1e-3f ? t * (1.f - t * (0.5f - t * (1.f / 3.f))) : __logf(1.f + t);
    return fmaxf(x, 0.f) + l;
}

template <int F, int VEC>
__device__ inline void message(const float (*pv)[VEC], const float (*qv)[VEC], const float (*wv)[VEC], bool has_p, bool has_w,
                               float* m) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        if constexpr (F == F_COPY) {
            m[v] = qv[0][v];
        } else if constexpr (F == F_ADD) {
            float t = qv[0][v];
            if (has_p) t = pv[0][v] + t;
            if (has_w) t = t + wv[0][v];
            m[v] = t;
        } else if constexpr (F == F_CGCONV) {
            float f = qv[0][v], s = qv[1][v];
            if (has_p) { f = pv[0][v] + f; s = pv[1][v] + s; }
            if (has_w) { f = f + wv[0][v]; s = s + wv[1][v]; }
            m[v] = sigmoid_f(f) * softplus_f(s);
        } else {   // F_FILM: beta = part 0, gamma = part 1
            m[v] = fmaxf(pv[1][v] * qv[0][v] + pv[0][v], 0.f);
        }
    }
}

// MULTI = false: one running sum (sum or mean only); true: sum, sum of squares, min and max together.
template <typename T, int F, bool MULTI, int VEC, bool NT>
__global__ __launch_bounds__(256) void edge_reduce_kernel(const Args a) {
    constexpr int U = Parts<F>::Q == 2 ? 4 : 8;
    constexpr int NQ = Parts<F>::Q, NP = Parts<F>::P > 0 ? Parts<F>::P : 1, NW = Parts<F>::W > 0 ? Parts<F>::W : 1;
    const T* q = (const T*)a.q;
    const T* p = (const T*)a.p;
    const T* w = (const T*)a.w;
    const T* add = (const T*)a.add;
    T* out = (T*)a.out;
    const bool has_p = Parts<F>::P > 0 && p != nullptr, has_w = Parts<F>::W > 0 && w != nullptr;
    const int G = 1 << a.gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> a.gshift;
    const int gl = (int)(gtid & (G - 1));
    const int64_t items = (int64_t)a.kchunks * a.N;
    const int64_t K = a.K;

    for (int64_t item = gtid >> a.gshift; item < items; item += ngroups) {
        const int64_t n = item % a.N;
        const int chunk = (int)(item / a.N);
        const int64_t c0 = ((int64_t)chunk * G + gl) * VEC;
        if (c0 >= K) continue;
        const int32_t beg = a.rowptr[n], end = a.rowptr[n + 1];

        float pv[NP][VEC];
        if (has_p) {
#pragma unroll
            for (int r = 0; r < NP; ++r) load_vec<T, VEC, false>(p + n * a.ldp + r * K + c0, pv[r]);
        }
        float sum[VEC], sq[VEC], mn[VEC], mx[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { sum[v] = 0.f; sq[v] = 0.f; mn[v] = __builtin_huge_valf(); mx[v] = -__builtin_huge_valf(); }

        for (int32_t j = beg; j < end; j += U) {
            int64_t c[U];
            int32_t e[U];
            float qv[U][NQ][VEC], wv[U][NW][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = -1;
                if (j + u < end) {
                    c[u] = a.col[j + u];
                    e[u] = a.perm ? a.perm[j + u] : j + u;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c[u] >= 0) {
#pragma unroll
                    for (int r = 0; r < NQ; ++r) load_vec<T, VEC, NT>(q + c[u] * a.ldq + r * K + c0, qv[u][r]);
                    if (has_w) {
#pragma unroll
                        for (int r = 0; r < NW; ++r) load_vec<T, VEC, true>(w + (int64_t)e[u] * a.ldw + r * K + c0, wv[u][r]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c[u] >= 0) {
                    float m[VEC];
                    message<F, VEC>(pv, qv[u], wv[u], has_p, has_w, m);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        sum[v] += m[v];
                        if constexpr (MULTI) {
                            sq[v] += m[v] * m[v];
                            mn[v] = fminf(mn[v], m[v]);
                            mx[v] = fmaxf(mx[v], m[v]);
                        }
                    }
                }
            }
        }

        const int32_t cnt = end - beg;
        const float degc = (float)(cnt < 1 ? 1 : cnt);
        const float logd = __logf(degc + 1.f);
        T* orow = out + n * a.ldo + c0;
        for (int s = 0; s < a.n_scal; ++s) {
            float scale = 1.f;
            switch (a.scal[s]) {
                case S_AMPLIFICATION: scale = logd / a.avg_log; break;
                case S_ATTENUATION: scale = a.avg_log / logd; break;
                case S_LINEAR: scale = degc / a.avg_lin; break;
                case S_INVERSE_LINEAR: scale = a.avg_lin / degc; break;
                default: break;
            }
            for (int g = 0; g < a.n_aggr; ++g) {
                float o[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float r;
                    switch (a.aggr[g]) {
                        case A_SUM: r = sum[v]; break;
                        case A_MEAN: r = sum[v] / degc; break;
                        case A_MIN: r = cnt > 0 ? mn[v] : 0.f; break;
                        case A_MAX: r = cnt > 0 ? mx[v] : 0.f; break;
                        default: {   // A_STD: sqrt(relu(mean(m^2) - mean(m)^2) + 1e-5), PNAConv.aggregate
                            const float mean = sum[v] / degc;
                            r = __fsqrt_rn(fmaxf(sq[v] / degc - mean * mean, 0.f) + 1e-5f);
                        }
                    }
                    o[v] = r * scale;
                }
                if (add && s == 0 && g == 0) {
                    float t[VEC];
                    load_vec<T, VEC, false>(add + n * a.ldadd + c0, t);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] += t[v];
                }
                store_vec<T, VEC>(orow + ((int64_t)s * a.n_aggr + g) * K, o);
            }
        }
    }
}

template <typename T, int F, bool MULTI>
int launch(const Args& a0, bool vec_ok, hipStream_t stream) {
    Args a = a0;
    constexpr int VEC = Elem<T>::VEC;
    // gathered tables beyond ~2 GiB do not stay in the Infinity Cache between visits: stream them past it (as spmm.hip)
    const bool nt = (double)a.N * (double)a.ldq * sizeof(T) > 2.0 * 1024 * 1024 * 1024;
    const int64_t lanes = vec_ok ? a.K / VEC : a.K;
    int gshift = 0;
    while ((1 << gshift) < lanes && gshift < 6) ++gshift;
    a.gshift = gshift;
    a.kchunks = (int)gnnops_cdiv(lanes, (int64_t)1 << gshift);
    const int64_t items = (int64_t)a.kchunks * a.N;
    const int grid = gnnops_grid_cap(gnnops_cdiv(items, 256 >> gshift), 256 * 64);
    if (vec_ok) {
        if (nt) hipLaunchKernelGGL((edge_reduce_kernel<T, F, MULTI, VEC, true>), dim3(grid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((edge_reduce_kernel<T, F, MULTI, VEC, false>), dim3(grid), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL((edge_reduce_kernel<T, F, MULTI, 1, false>), dim3(grid), dim3(256), 0, stream, a);
    }
    return gnnops_check_launch("edge_reduce");
}

template <typename T>
int dispatch(int functor, bool multi, const Args& a, bool vec_ok, hipStream_t stream) {
#define GNNOPS_ER(F) return multi ? launch<T, F, true>(a, vec_ok, stream) : launch<T, F, false>(a, vec_ok, stream)
    switch (functor) {
        case F_COPY: GNNOPS_ER(F_COPY);
        case F_ADD: GNNOPS_ER(F_ADD);
        case F_CGCONV: GNNOPS_ER(F_CGCONV);
        case F_FILM: GNNOPS_ER(F_FILM);
    }
#undef GNNOPS_ER
    gnnops_set_error("edge_reduce: unknown functor %d", functor);
    return GNNOPS_EINVAL;
}

}  // namespace

extern "C" int gnnops_edge_reduce(int functor, const void* q, int64_t ldq, const void* p, int64_t ldp, const void* w, int64_t ldw,
                                  const void* add, int64_t ldadd, const int32_t* rowptr, const int32_t* perm, const int64_t* col,
                                  void* out, int64_t ldo, int64_t N, int64_t E, int64_t K, const int* aggr, int n_aggr,
                                  const int* scalers, int n_scalers, float avg_deg_log, float avg_deg_lin, int dtype,
                                  gnnops_stream_t s) {
    GNNOPS_REQUIRE(N >= 0 && E >= 0 && K >= 0, GNNOPS_EINVAL, "edge_reduce: negative size");
    GNNOPS_REQUIRE(N < ((int64_t)1 << 31) && E < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "edge_reduce: N and E must be < 2^31");
    GNNOPS_REQUIRE(functor >= F_COPY && functor <= F_FILM, GNNOPS_EINVAL, "edge_reduce: unknown functor %d", functor);
    GNNOPS_REQUIRE(n_aggr >= 1 && n_aggr <= 5 && aggr, GNNOPS_EINVAL, "edge_reduce: 1..5 aggregators");
    GNNOPS_REQUIRE(n_scalers >= 0 && n_scalers <= 5 && (n_scalers == 0 || scalers), GNNOPS_EINVAL, "edge_reduce: 0..5 scalers");
    if (N * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && out && (E == 0 || (q && col)), GNNOPS_EINVAL, "edge_reduce: null pointer");
    GNNOPS_REQUIRE(functor != F_FILM || p, GNNOPS_EINVAL, "edge_reduce: the FiLM message needs p = [beta | gamma]");
    Args a{};
    a.q = q; a.p = p; a.w = w; a.add = add; a.rowptr = rowptr; a.perm = perm; a.col = col; a.out = out;
    a.N = N; a.K = K; a.ldq = ldq; a.ldp = ldp; a.ldw = ldw; a.ldadd = ldadd; a.ldo = ldo;
    bool multi = false;
    a.n_aggr = n_aggr;
    for (int i = 0; i < n_aggr; ++i) {
        GNNOPS_REQUIRE(aggr[i] >= A_SUM && aggr[i] <= A_STD, GNNOPS_EINVAL, "edge_reduce: unknown aggregator %d", aggr[i]);
        a.aggr[i] = aggr[i];
        multi = multi || aggr[i] >= A_MIN;
    }
    a.n_scal = n_scalers > 0 ? n_scalers : 1;
    a.scal[0] = S_IDENTITY;
    for (int i = 0; i < n_scalers; ++i) {
        GNNOPS_REQUIRE(scalers[i] >= S_IDENTITY && scalers[i] <= S_INVERSE_LINEAR, GNNOPS_EINVAL, "edge_reduce: unknown scaler %d",
                       scalers[i]);
        a.scal[i] = scalers[i];
    }
    a.avg_log = avg_deg_log;
    a.avg_lin = avg_deg_lin;
    const int parts_q[4] = {1, 1, 2, 1}, parts_p[4] = {0, 1, 2, 2}, parts_w[4] = {0, 1, 2, 0};
    GNNOPS_REQUIRE(ldq >= parts_q[functor] * K && (!p || ldp >= parts_p[functor] * K) && (!w || ldw >= parts_w[functor] * K) &&
                       (!add || ldadd >= K) && ldo >= (int64_t)a.n_scal * n_aggr * K,
                   GNNOPS_EINVAL, "edge_reduce: a row pitch is shorter than the row");
    size_t es;
    int vec;
    switch (dtype) {
        case GNNOPS_F32: es = 4; vec = 4; break;
        case GNNOPS_F16: case GNNOPS_BF16: es = 2; vec = 8; break;
        default: gnnops_set_error("edge_reduce: unknown dtype %d", dtype); return GNNOPS_EINVAL;
    }
    auto ok16 = [&](const void* ptr, int64_t ld) { return !ptr || ((uintptr_t)ptr % 16 == 0 && (ld * es) % 16 == 0); };
    const bool vec_ok = K % vec == 0 && ok16(q, ldq) && ok16(p, ldp) && ok16(w, ldw) && ok16(add, ldadd) && ok16(out, ldo);
    hipStream_t stream = (hipStream_t)s;
    switch (dtype) {
        case GNNOPS_F32: return dispatch<float>(functor, multi, a, vec_ok, stream);
        case GNNOPS_F16: return dispatch<__half>(functor, multi, a, vec_ok, stream);
        default: return dispatch<__hip_bfloat16>(functor, multi, a, vec_ok, stream);
    }
}
