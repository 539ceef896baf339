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
// One lane group per (destination row, column chunk) as in segment.hip / spmm.hip — 16-B lanes for the bandwidth-bound
// messages, ONE ROW PER WAVE (4-byte lanes, scalar edge ids, rows software-pipelined) for the arithmetic-heavy ones (see
// edge_reduce_kernel): each gathered row is read once, each output row stored once, messages live in registers, fp32
// arithmetic, ONE rounding on store. Destinations with more than 8192 edges are reduced piecewise (hub passes below).
// Algorithmic bytes per launch: E * (q row + 8 B column id (+ w row)) + N * (p row + out row) + 4 (N + 1); measured traffic
// and the optimisation ladder: DESIGN.md §4 "Message-passing layers", profiles/round2_f_*.
#include "common.h"
#include <stdlib.h>

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
    int nt;   // gathered table does not fit the Infinity Cache: stream it past (nontemporal 16-B loads)
    // heavy destinations ("hubs": more than T_HUB edges) — set aside by the main kernel, reduced piecewise afterwards
    unsigned int* hub_count;   // [0] hubs found, [1] pieces in all
    int32_t* hub_rows;         // [max_hubs] destination ids
    int32_t* piece_base;       // [max_hubs + 1] first piece of each hub
    float* partial;            // [pieces][accs][K] fp32 partial accumulators (accs = 4 for MULTI, else 1)
    int max_hubs;
};

constexpr int T_HUB = 8192;    // as csrc/hub.h: a destination with more edges than this is not left to one lane group
constexpr int PIECE = 2048;    // edges per piece of a hub

template <int F> struct Parts;   // K-wide parts per row of q / p / w
template <> struct Parts<F_COPY> { static constexpr int Q = 1, P = 0, W = 0; };
template <> struct Parts<F_ADD> { static constexpr int Q = 1, P = 1, W = 1; };
template <> struct Parts<F_CGCONV> { static constexpr int Q = 2, P = 2, W = 2; };
template <> struct Parts<F_FILM> { static constexpr int Q = 1, P = 2, W = 0; };

// VEC consecutive elements of a row as one access of VEC * sizeof(T) bytes (16, 8, 4 or the element itself). Loading and
// unpacking are separate so that a step's loads can all be issued before the first conversion waits for one of them.
template <typename T, int VEC>
__device__ inline u32x4 load_raw(const T* p, bool nt = false) {
    constexpr int BYTES = VEC * (int)sizeof(T);
    u32x4 r = {0u, 0u, 0u, 0u};
    if constexpr (BYTES == 16) {
        r = nt ? load16<true>(p) : load16<false>(p);
    } else if constexpr (BYTES == 8) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        r.x = t.x; r.y = t.y;
    } else if constexpr (BYTES == 4) {
        r.x = *reinterpret_cast<const uint32_t*>(p);
    } else {
        r.x = *reinterpret_cast<const uint16_t*>(p);
    }
    return r;
}
template <typename T, int VEC>
__device__ inline void unpack_vec(const u32x4& r, float* f) {
    float g[Elem<T>::VEC];
    Elem<T>::unpack(r, g);
#pragma unroll
    for (int v = 0; v < VEC; ++v) f[v] = g[v];
}
template <typename T, int VEC>
__device__ inline void load_vec(const T* p, float* f, bool nt = false) { unpack_vec<T, VEC>(load_raw<T, VEC>(p, nt), f); }
template <typename T, int VEC>
__device__ inline void store_vec(T* p, const float* f) {
    constexpr int BYTES = VEC * (int)sizeof(T);
    if constexpr (VEC == 1) {
        Elem<T>::store(p, f[0]);
    } else {
        float g[Elem<T>::VEC];
#pragma unroll
        for (int v = 0; v < Elem<T>::VEC; ++v) g[v] = v < VEC ? f[v] : 0.f;
        const u32x4 r = Elem<T>::pack(g);
        if constexpr (BYTES == 16) store16<true>(p, r);
        else if constexpr (BYTES == 8) *reinterpret_cast<uint2*>(p) = uint2{r.x, r.y};
        else *reinterpret_cast<uint32_t*>(p) = r.x;
    }
}

// Straight-line fp32 forms (no per-element branch, no IEEE division sequence: at 5 edges x 128 columns per row the edge loop
// is otherwise VALU-bound — 2353 vector instructions and 191 branches per 32 messages before, see profiles/round2_f_*):
// one v_exp + one v_rcp for the sigmoid; one v_exp + one v_log for the softplus.
__device__ inline float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269504f)); }
// softplus(x) = max(x, 0) + log1p(exp(-|x|)). torch's form (beta 1, threshold 20: x itself above 20) is the same number in
// fp32: above 20 the log1p term is < 2.1e-9, below half an ulp of x. fp32 storage: log1p by its series where 1 + t would
// round t away; elsewhere 1 + t is in [1, 2], so the bare v_log_f32 needs none of logf's range handling.
template <bool PRECISE>
__device__ inline float softplus_f(float x) {
    const float t = __builtin_amdgcn_exp2f(fabsf(x) * -1.44269504f);
    const float lg = __builtin_amdgcn_logf(1.f + t) * 0.693147181f;
    if constexpr (PRECISE) {
        const float series = t * (1.f - t * (0.5f - t * (1.f / 3.f)));
        return fmaxf(x, 0.f) + (t < 1e-3f ? series : lg);
    } else {
        // 16-bit storage: 1 + t rounds t to 2^-24, an absolute error of 6e-8 in a message that is rounded to 2^-11 (fp16) or
        // 2^-8 (bf16) of the row's sum afterwards — the series is six instructions per element the result cannot show
        return fmaxf(x, 0.f) + lg;
    }
}

template <int F, bool HAS_W, int VEC, bool PRECISE>
__device__ inline void message(const float (*pv)[VEC], const float (*qv)[VEC], const float (*wv)[VEC], float* m) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        if constexpr (F == F_COPY) {
            m[v] = qv[0][v];
        } else if constexpr (F == F_ADD) {
            float t = pv[0][v] + qv[0][v];
            if constexpr (HAS_W) t = t + wv[0][v];
            m[v] = t;
        } else if constexpr (F == F_CGCONV) {
            float f = pv[0][v] + qv[0][v], s = pv[1][v] + qv[1][v];
            if constexpr (HAS_W) { f = f + wv[0][v]; s = s + wv[1][v]; }
            m[v] = sigmoid_f(f) * softplus_f<PRECISE>(s);
        } else {   // F_FILM: beta = part 0, gamma = part 1
            m[v] = fmaxf(pv[1][v] * qv[0][v] + pv[0][v], 0.f);
        }
    }
}

// MULTI = false: one running sum (sum or mean only); true: sum, sum of squares, min and max together.
// Lanes per row: a message that costs real arithmetic (CGCONV: two exp, a log and a reciprocal per element; the four
// accumulators of MULTI) is spread over as many lanes as the row has 4-byte pieces — up to the whole wave on ONE row — so
// that rows of different degree do not share a wave: with 16-B lanes four rows of K = 128 fp16 share one, the wave runs
// for its longest row (~7.7 edges at a mean of 5) and half the issued messages are masked off. Plain copies / adds keep
// the 16-B lanes (fewest memory instructions; they are bandwidth-bound).
// the edges [jb, je) of destination n folded into the running accumulators (columns c0 .. c0 + VEC of every part)
template <typename T, int F, bool MULTI, bool HAS_W, int VEC, bool WAVE_ROW>
__device__ inline void accumulate(const Args& a, int n, int c0, int32_t jb, int32_t je, float* sum, float* sq, float* mn, float* mx,
                                  bool have_first = false, int cl_first = 0, int el_first = 0) {
    constexpr int U = Parts<F>::Q == 2 ? 4 : 8;
    constexpr int NQ = Parts<F>::Q, NP = Parts<F>::P > 0 ? Parts<F>::P : 1, NW = Parts<F>::W > 0 ? Parts<F>::W : 1;
    const T* __restrict__ q = (const T*)a.q;
    const T* __restrict__ p = (const T*)a.p;
    const T* __restrict__ w = (const T*)a.w;
    const int32_t* __restrict__ perm = a.perm;
    const int64_t* __restrict__ col = a.col;
    const int K = (int)a.K, ldq = (int)a.ldq, ldw = (int)a.ldw;   // < 2^31 (host-checked): row offsets are one 32 x 32 -> 64 multiply
    if constexpr (WAVE_ROW) {   // the edge loop, its bounds checks and the row base addresses are scalar work here
        jb = __builtin_amdgcn_readfirstlane(jb);
        je = __builtin_amdgcn_readfirstlane(je);
    }
    float pv[NP][VEC];
    if constexpr (Parts<F>::P > 0) {
#pragma unroll
        for (int r = 0; r < NP; ++r) load_vec<T, VEC>(p + (int64_t)n * a.ldp + r * K + c0, pv[r]);
    }
    // WAVE_ROW: the ids of up to 64 edges are fetched by ONE coalesced load (lane l holds id l of the run) and handed out
    // lane by lane — they are wave-uniform there, and U dependent single-id loads per step cost a memory latency each
    const int lane = threadIdx.x & 63;
    for (int32_t jrun = jb; jrun < je; jrun += WAVE_ROW ? 64 : (je - jb)) {
    int cl = 0, el = 0;
    const int32_t jrun_end = WAVE_ROW ? min(jrun + 64, je) : je;
    if constexpr (WAVE_ROW) {
        if (have_first && jrun == jb) {   // the caller fetched this run's ids an iteration ago (edge_reduce_kernel's pipeline)
            cl = cl_first;
            el = el_first;
        } else if (jrun + lane < je) {
            cl = (int)col[jrun + lane];
            if constexpr (HAS_W) el = perm ? perm[jrun + lane] : jrun + lane;
        }
    }
    for (int32_t j = jrun; j < jrun_end; j += U) {
        int c[U], e[U];
        int64_t qrow[U], wrow[U];
        u32x4 qr[U][NQ], wr[U][NW];
#pragma unroll
        for (int u = 0; u < U; ++u) { c[u] = 0; e[u] = 0; }
        if constexpr (WAVE_ROW) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = __builtin_amdgcn_readlane(cl, (j - jrun + u) & 63);
                if constexpr (HAS_W) e[u] = __builtin_amdgcn_readlane(el, (j - jrun + u) & 63);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {       // all the step's edge ids first ...
                if (j + u < jrun_end) {
                    c[u] = (int)col[j + u];
                    if constexpr (HAS_W) e[u] = perm ? perm[j + u] : j + u;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {       // ... their row addresses: every wait for an id happens HERE, because ...
            qrow[u] = (int64_t)c[u] * ldq;      // element offsets, not pointers: a pointer through asm loses its address space
            if constexpr (HAS_W) wrow[u] = (int64_t)e[u] * ldw;
            // ... the empty volatile statements pin the address arithmetic to this spot. Left alone, the compiler sinks each
            // address into the guarded block of its row load, and the wait for id u lands between two row loads — where it
            // also waits for every row already in flight (vmcnt counts in order): a vmcnt(0) before every row load in one
            // build of this loop (gather-sum 3.5 -> 4.0 ms, cgconv 8.9 -> 10.7 ms at N = 10M, E = 50M).
            if constexpr (WAVE_ROW) {           // ids are scalars: the row base stays in SGPRs, the lane's column is the load's VGPR offset
                asm volatile("" : "+s"(qrow[u]));
                if constexpr (HAS_W) asm volatile("" : "+s"(wrow[u]));
            } else {
                qrow[u] += c0;
                if constexpr (HAS_W) wrow[u] += c0;
                asm volatile("" : "+v"(qrow[u]));
                if constexpr (HAS_W) asm volatile("" : "+v"(wrow[u]));
            }
        }
        const uint32_t lane_off = WAVE_ROW ? (uint32_t)c0 * (uint32_t)sizeof(T) : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {       // ... then all its row loads in flight together (raw words: no conversion here)
            if (j + u < jrun_end) {
#pragma unroll
                for (int r = 0; r < NQ; ++r) {
                    const T* row = q + qrow[u] + r * K;
                    qr[u][r] = load_raw<T, VEC>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(row) + lane_off), a.nt != 0);
                }
                if constexpr (HAS_W) {
#pragma unroll
                    for (int r = 0; r < NW; ++r) {
                        const T* row = w + wrow[u] + r * K;
                        wr[u][r] = load_raw<T, VEC>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(row) + lane_off), true);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j + u < jrun_end) {   // skipped outright once every row of the wave is past its end
                float qv[NQ][VEC], wv[NW][VEC], m[VEC];
#pragma unroll
                for (int r = 0; r < NQ; ++r) unpack_vec<T, VEC>(qr[u][r], qv[r]);
                if constexpr (HAS_W) {
#pragma unroll
                    for (int r = 0; r < NW; ++r) unpack_vec<T, VEC>(wr[u][r], wv[r]);
                }
                message<F, HAS_W, VEC, sizeof(T) == 4>(pv, qv, wv, m);
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
    }
}

// accumulators of a whole destination -> its output blocks: aggregators x degree scalers (+ the residual row)
template <typename T, int VEC>
__device__ inline void finish_row(const Args& a, int n, int c0, int32_t cnt, const float* sum, const float* sq, const float* mn,
                                  const float* mx) {
    const T* __restrict__ add = (const T*)a.add;
    const int K = (int)a.K;
    const float degc = (float)(cnt < 1 ? 1 : cnt);
    T* orow = (T*)a.out + (int64_t)n * a.ldo + c0;
    // The common layers (GIN / SAGE / CGConv / FiLM) want ONE block, sum or mean, no scaler: a handful of instructions. The
    // general form below costs ~500 per row (a logarithm, eight IEEE divisions and a square root that the compiler hoists out
    // of the aggregator switch) — twice the arithmetic of the five cgconv messages of an average row when it ran for every row.
    if (a.n_scal == 1 && a.scal[0] == S_IDENTITY && a.n_aggr == 1 && a.aggr[0] <= A_MEAN) {
        float o[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = a.aggr[0] == A_MEAN ? sum[v] / degc : sum[v];
        if (add) {
            float t[VEC];
            load_vec<T, VEC>(add + (int64_t)n * a.ldadd + c0, t);
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] += t[v];
        }
        store_vec<T, VEC>(orow, o);
        return;
    }
    // General form (PNAConv: several aggregators x degree scalers). Everything that depends on the row only is computed once:
    // ONE division (1 / deg) and the bare v_log / v_rcp / v_sqrt instructions — as first written (sum / deg per element and
    // aggregator, logf, sqrtf) this epilogue was ~500 vector instructions per row and bounded the whole PNA pass.
    const float inv_deg = 1.f / degc;
    const float logd = __builtin_amdgcn_logf(degc + 1.f) * 0.693147181f;
    float mean[VEC], stdv[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        mean[v] = sum[v] * inv_deg;
        // sqrt(relu(mean(m^2) - mean(m)^2) + 1e-5), PNAConv.aggregate
        stdv[v] = __builtin_amdgcn_sqrtf(fmaxf(sq[v] * inv_deg - mean[v] * mean[v], 0.f) + 1e-5f);
    }
    for (int s = 0; s < a.n_scal; ++s) {
        float scale = 1.f;
        switch (a.scal[s]) {
            case S_AMPLIFICATION: scale = logd * __builtin_amdgcn_rcpf(a.avg_log); break;
            case S_ATTENUATION: scale = a.avg_log * __builtin_amdgcn_rcpf(logd); break;
            case S_LINEAR: scale = degc * __builtin_amdgcn_rcpf(a.avg_lin); break;
            case S_INVERSE_LINEAR: scale = a.avg_lin * inv_deg; break;
            default: break;
        }
        for (int g = 0; g < a.n_aggr; ++g) {
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float r;
                switch (a.aggr[g]) {
                    case A_SUM: r = sum[v]; break;
                    case A_MEAN: r = mean[v]; break;
                    case A_MIN: r = cnt > 0 ? mn[v] : 0.f; break;
                    case A_MAX: r = cnt > 0 ? mx[v] : 0.f; break;
                    default: r = stdv[v]; break;
                }
                o[v] = r * scale;
            }
            if (add && s == 0 && g == 0) {
                float t[VEC];
                load_vec<T, VEC>(add + (int64_t)n * a.ldadd + c0, t);
#pragma unroll
                for (int v = 0; v < VEC; ++v) o[v] += t[v];
            }
            store_vec<T, VEC>(orow + (int64_t)(s * a.n_aggr + g) * K, o);
        }
    }
}

template <int VEC>
__device__ inline void reset(float* sum, float* sq, float* mn, float* mx) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) { sum[v] = 0.f; sq[v] = 0.f; mn[v] = __builtin_huge_valf(); mx[v] = -__builtin_huge_valf(); }
}

template <typename T, int F, bool MULTI, bool HAS_W, int VEC, bool WAVE_ROW>
__global__ __launch_bounds__(256) void edge_reduce_kernel(const Args a) {
    const int32_t* __restrict__ rowptr = a.rowptr;
    const int G = 1 << a.gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> a.gshift;
    const int gl = (int)(gtid & (G - 1));
    const int64_t items = (int64_t)a.kchunks * a.N;
    const int K = (int)a.K;

    if constexpr (WAVE_ROW) {
        // The whole wave is on one row: bounds, edge ids and the edge loop are scalar. A row is a chain of three dependent
        // memory latencies — rowptr -> edge ids -> gathered rows — and a wave has ~10 elements per lane to compute per row,
        // so the chain, not the arithmetic, set the pace (~6 ms of the 9 ms of cgconv D = 128 fp16 at N = 10M). The loop is
        // software-pipelined over the wave's rows: while row k is reduced, the ids of row k+1 and the bounds of row k+2 are
        // already in flight.
        const int64_t* __restrict__ col = a.col;
        const int32_t* __restrict__ perm = a.perm;
        auto decode = [&](int64_t it, int& n, int& chunk) {
            if (a.kchunks == 1) { n = (int)it; chunk = 0; } else { n = (int)(it % a.N); chunk = (int)(it / a.N); }
            n = __builtin_amdgcn_readfirstlane(n);
            chunk = __builtin_amdgcn_readfirstlane(chunk);
        };
        int64_t it0 = gtid >> 6;
        if (it0 >= items) return;
        int n0, ch0, n1 = 0, ch1 = 0;
        decode(it0, n0, ch0);
        int32_t b0 = rowptr[n0], e0 = rowptr[n0 + 1], b1 = 0, e1 = 0;
        int64_t it1 = it0 + ngroups;
        bool has1 = it1 < items;
        if (has1) { decode(it1, n1, ch1); b1 = rowptr[n1]; e1 = rowptr[n1 + 1]; }
        b0 = __builtin_amdgcn_readfirstlane(b0);
        e0 = __builtin_amdgcn_readfirstlane(e0);
        int cl0 = 0, el0 = 0;
        if (b0 + gl < e0) {
            cl0 = (int)col[b0 + gl];
            if constexpr (HAS_W) el0 = perm ? perm[b0 + gl] : b0 + gl;
        }
        while (true) {
            const int64_t it2 = it1 + ngroups;
            const bool has2 = has1 && it2 < items;
            int n2 = 0, ch2 = 0, cl1 = 0, el1 = 0;
            int32_t b2 = 0, e2 = 0;
            if (has2) { decode(it2, n2, ch2); b2 = rowptr[n2]; e2 = rowptr[n2 + 1]; }   // bounds of row k+2: used an iteration from now
            if (has1) {                                                                    // ids of row k+1: its bounds were fetched an iteration ago
                b1 = __builtin_amdgcn_readfirstlane(b1);
                e1 = __builtin_amdgcn_readfirstlane(e1);
                if (b1 + gl < e1) {
                    cl1 = (int)col[b1 + gl];
                    if constexpr (HAS_W) el1 = perm ? perm[b1 + gl] : b1 + gl;
                }
            }
            // row k
            int c0 = (ch0 * G + gl) * VEC;
            const bool in_row = c0 < K;
            if (!in_row) c0 = 0;   // lanes past the row end stay in the wave (its lanes hand the edge ids round): they redo column 0, store nothing
            if (a.hub_count && e0 - b0 > T_HUB) {   // a hub: set aside for the piecewise pass, neither reduced nor stored here
                if (gl == 0 && ch0 == 0) {
                    const unsigned int slot = atomicAdd(a.hub_count, 1u);
                    if ((int)slot < a.max_hubs) a.hub_rows[slot] = n0;
                }
            } else {
                float sum[VEC], sq[VEC], mn[VEC], mx[VEC];
                reset<VEC>(sum, sq, mn, mx);
                accumulate<T, F, MULTI, HAS_W, VEC, true>(a, n0, c0, b0, e0, sum, sq, mn, mx, true, cl0, el0);
                if (in_row) finish_row<T, VEC>(a, n0, c0, e0 - b0, sum, sq, mn, mx);
            }
            if (!has1) break;
            n0 = n1; ch0 = ch1; b0 = b1; e0 = e1; cl0 = cl1; el0 = el1;
            it1 = it2; has1 = has2; n1 = n2; ch1 = ch2; b1 = b2; e1 = e2;
        }
    } else {
        for (int64_t item = gtid >> a.gshift; item < items; item += ngroups) {
            int n, chunk;
            if (a.kchunks == 1) { n = (int)item; chunk = 0; } else { n = (int)(item % a.N); chunk = (int)(item / a.N); }
            const int c0 = (chunk * G + gl) * VEC;
            if (c0 >= K) continue;
            const int32_t beg = rowptr[n], end = rowptr[n + 1];
            if (a.hub_count && end - beg > T_HUB) {   // a hub: set aside for the piecewise pass, neither reduced nor stored here
                if (gl == 0 && chunk == 0) {
                    const unsigned int slot = atomicAdd(a.hub_count, 1u);
                    if ((int)slot < a.max_hubs) a.hub_rows[slot] = n;
                }
                continue;
            }
            float sum[VEC], sq[VEC], mn[VEC], mx[VEC];
            reset<VEC>(sum, sq, mn, mx);
            accumulate<T, F, MULTI, HAS_W, VEC, false>(a, n, c0, beg, end, sum, sq, mn, mx);
            finish_row<T, VEC>(a, n, c0, end - beg, sum, sq, mn, mx);
        }
    }
}

// ---- hubs: pieces of PIECE edges reduced by separate lane groups into fp32 partials, combined in piece order ----
__global__ void hub_pieces_kernel(const Args a) {   // one thread: first piece of every hub (there are at most E / T_HUB of them)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned int hubs = a.hub_count[0];
    if ((int)hubs > a.max_hubs) hubs = (unsigned)a.max_hubs;
    int32_t base = 0;
    for (unsigned int h = 0; h < hubs; ++h) {
        a.piece_base[h] = base;
        const int n = a.hub_rows[h];
        base += (a.rowptr[n + 1] - a.rowptr[n] + PIECE - 1) / PIECE;
    }
    a.piece_base[hubs] = base;
    a.hub_count[1] = (unsigned int)base;
}

template <typename T, int F, bool MULTI, bool HAS_W, int VEC, bool WAVE_ROW>
__global__ __launch_bounds__(256) void hub_partial_kernel(const Args a) {
    constexpr int ACCS = MULTI ? 4 : 1;
    const int G = 1 << a.gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> a.gshift;
    const int gl = (int)(gtid & (G - 1));
    const int hubs = min((int)a.hub_count[0], a.max_hubs);
    const int64_t items = (int64_t)a.hub_count[1] * a.kchunks;
    const int K = (int)a.K;
    for (int64_t item = gtid >> a.gshift; item < items; item += ngroups) {
        int piece = (int)(item / a.kchunks), chunk = (int)(item % a.kchunks);
        if constexpr (WAVE_ROW) {
            piece = __builtin_amdgcn_readfirstlane(piece);
            chunk = __builtin_amdgcn_readfirstlane(chunk);
        }
        int lo = 0, hi = hubs;   // hub h with piece_base[h] <= piece < piece_base[h + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.piece_base[mid] <= piece) lo = mid; else hi = mid;
        }
        const int n = a.hub_rows[lo];
        int c0 = (chunk * G + gl) * VEC;
        const bool in_row = c0 < K;
        if constexpr (WAVE_ROW) {
            if (!in_row) c0 = 0;
        } else if (!in_row) {
            continue;
        }
        int32_t jb = a.rowptr[n] + (piece - a.piece_base[lo]) * PIECE;
        int32_t je = min(jb + PIECE, a.rowptr[n + 1]);
        if constexpr (WAVE_ROW) {
            jb = __builtin_amdgcn_readfirstlane(jb);
            je = __builtin_amdgcn_readfirstlane(je);
        }
        float sum[VEC], sq[VEC], mn[VEC], mx[VEC];
        reset<VEC>(sum, sq, mn, mx);
        accumulate<T, F, MULTI, HAS_W, VEC, WAVE_ROW>(a, n, c0, jb, je, sum, sq, mn, mx);
        if (!in_row) continue;
        float* pp = a.partial + (int64_t)piece * ACCS * K + c0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            pp[v] = sum[v];
            if constexpr (MULTI) { pp[K + v] = sq[v]; pp[2 * K + v] = mn[v]; pp[3 * K + v] = mx[v]; }
        }
    }
}

template <typename T, bool MULTI, int VEC>
__global__ __launch_bounds__(256) void hub_finish_kernel(const Args a) {
    constexpr int ACCS = MULTI ? 4 : 1;
    const int G = 1 << a.gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> a.gshift;
    const int gl = (int)(gtid & (G - 1));
    const int hubs = min((int)a.hub_count[0], a.max_hubs);
    const int K = (int)a.K;
    for (int64_t item = gtid >> a.gshift; item < (int64_t)hubs * a.kchunks; item += ngroups) {
        const int h = (int)(item / a.kchunks), chunk = (int)(item % a.kchunks);
        const int c0 = (chunk * G + gl) * VEC;
        if (c0 >= K) continue;
        const int n = a.hub_rows[h];
        float sum[VEC], sq[VEC], mn[VEC], mx[VEC];
        reset<VEC>(sum, sq, mn, mx);
        for (int piece = a.piece_base[h]; piece < a.piece_base[h + 1]; ++piece) {   // in piece order: the same result every run
            const float* pp = a.partial + (int64_t)piece * ACCS * K + c0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                sum[v] += pp[v];
                if constexpr (MULTI) { sq[v] += pp[K + v]; mn[v] = fminf(mn[v], pp[2 * K + v]); mx[v] = fmaxf(mx[v], pp[3 * K + v]); }
            }
        }
        finish_row<T, VEC>(a, n, c0, a.rowptr[n + 1] - a.rowptr[n], sum, sq, mn, mx);
    }
}

template <typename T, int F, bool MULTI, bool HAS_W, int VEC>
int launch_vec(Args& a, hipStream_t stream) {
    // gathered tables beyond ~2 GiB do not stay in the Infinity Cache between visits: stream them past it (as spmm.hip)
    a.nt = (double)a.N * (double)a.ldq * sizeof(T) > 2.0 * 1024 * 1024 * 1024;
    const int64_t lanes = a.K / VEC;
    int gshift = 0;
    while ((1 << gshift) < lanes && gshift < 6) ++gshift;
    a.gshift = gshift;
    a.kchunks = (int)gnnops_cdiv(lanes, (int64_t)1 << gshift);
    const int64_t items = (int64_t)a.kchunks * a.N;
    const int grid = gnnops_grid_cap(gnnops_cdiv(items, 256 >> gshift), 256 * 64);
    if (a.hub_count && gnnops_memset_async(a.hub_count, 0, 8, stream) != hipSuccess) return gnnops_check_launch("edge_reduce hub memset");
    if (gshift == 6) hipLaunchKernelGGL((edge_reduce_kernel<T, F, MULTI, HAS_W, VEC, true>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((edge_reduce_kernel<T, F, MULTI, HAS_W, VEC, false>), dim3(grid), dim3(256), 0, stream, a);
    if (a.hub_count) {   // the hub passes find their own work on the device (counts are never read back); empty when there is no hub
        hipLaunchKernelGGL(hub_pieces_kernel, dim3(1), dim3(64), 0, stream, a);
        const int hgrid = 256 * 8;
        if (gshift == 6) hipLaunchKernelGGL((hub_partial_kernel<T, F, MULTI, HAS_W, VEC, true>), dim3(hgrid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((hub_partial_kernel<T, F, MULTI, HAS_W, VEC, false>), dim3(hgrid), dim3(256), 0, stream, a);
        hipLaunchKernelGGL((hub_finish_kernel<T, MULTI, VEC>), dim3(256), dim3(256), 0, stream, a);
    }
    return gnnops_check_launch("edge_reduce");
}

template <typename T, int F, bool MULTI, bool HAS_W>
int launch(const Args& a0, int max_vec, hipStream_t stream) {
    Args a = a0;
    constexpr int NATIVE = Elem<T>::VEC;
    constexpr bool HEAVY = F == F_CGCONV || MULTI;
    // widest piece the operands allow (max_vec: alignment of every pointer / pitch, in elements), narrowed for HEAVY
    // messages until the row fills the wave
    int vec = NATIVE;
    while (vec > 1 && (vec > max_vec || a.K % vec != 0)) vec >>= 1;
    if (HEAVY)
        while (vec > 1 && vec * sizeof(T) > 4 && a.K / vec < 64) vec >>= 1;
    const char* sw = getenv("GNNOPS_EDGE_VEC");   // A/B (tools/time_edge_reduce.py): force the piece width in elements
    if (sw) { int f = atoi(sw); if (f >= 1 && f <= vec && (f & (f - 1)) == 0) vec = f; }
    if constexpr (NATIVE == 8) {
        if (vec == 8) return launch_vec<T, F, MULTI, HAS_W, 8>(a, stream);
    }
    if (vec == 4) return launch_vec<T, F, MULTI, HAS_W, 4>(a, stream);
    if (vec == 2) return launch_vec<T, F, MULTI, HAS_W, 2>(a, stream);
    return launch_vec<T, F, MULTI, HAS_W, 1>(a, stream);
}

template <typename T>
int dispatch(int functor, bool multi, const Args& a, int max_vec, hipStream_t stream) {
#define GNNOPS_ER(F, W) return multi ? launch<T, F, true, W>(a, max_vec, stream) : launch<T, F, false, W>(a, max_vec, stream)
    const bool has_w = a.w != nullptr;
    switch (functor) {
        case F_COPY: GNNOPS_ER(F_COPY, false);
        case F_ADD: if (has_w) GNNOPS_ER(F_ADD, true); else GNNOPS_ER(F_ADD, false);
        case F_CGCONV: if (has_w) GNNOPS_ER(F_CGCONV, true); else GNNOPS_ER(F_CGCONV, false);
        case F_FILM: GNNOPS_ER(F_FILM, false);
    }
#undef GNNOPS_ER
    gnnops_set_error("edge_reduce: unknown functor %d", functor);
    return GNNOPS_EINVAL;
}

}  // namespace

namespace {
struct HubLayout { size_t counters, rows, base, partial, total; int max_hubs; };
inline HubLayout hub_layout(int64_t E, int64_t K) {
    HubLayout l{};
    l.max_hubs = (int)(E / T_HUB) + 1;
    const size_t pieces = (size_t)(E / PIECE) + (size_t)l.max_hubs;   // sum over hubs of ceil(deg / PIECE)
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    l.counters = 0;
    l.rows = 256;
    l.base = l.rows + up((size_t)l.max_hubs * 4);
    l.partial = l.base + up(((size_t)l.max_hubs + 1) * 4);
    l.total = l.partial + pieces * 4 * (size_t)K * 4;
    return l;
}
}  // namespace

extern "C" size_t gnnops_edge_reduce_hub_workspace_bytes(int64_t E, int64_t K) {
    // below 32768 edges a hub costs one wave at most ~1.5 ms, while the three (empty) hub launches would cost every
    // launch-bound layer call on a batch of small graphs ~12 us
    if (E <= 4 * T_HUB || K <= 0) return 0;
    const char* sw = getenv("GNNOPS_EDGE_HUBS");   // A/B (tools/time_edge_hubs.py): 0 = leave every destination to one lane group
    if (sw && sw[0] == '0') return 0;
    return hub_layout(E, K).total;
}

extern "C" int gnnops_edge_reduce(int functor, const void* q, int64_t ldq, const void* p, int64_t ldp, const void* w, int64_t ldw,
                                  const void* add, int64_t ldadd, const int32_t* rowptr, const int32_t* perm, const int64_t* col,
                                  void* out, int64_t ldo, int64_t N, int64_t E, int64_t K, const int* aggr, int n_aggr,
                                  const int* scalers, int n_scalers, float avg_deg_log, float avg_deg_lin, int dtype,
                                  gnnops_stream_t s) {
    return gnnops_edge_reduce_hubs(functor, q, ldq, p, ldp, w, ldw, add, ldadd, rowptr, perm, col, out, ldo, N, E, K, aggr, n_aggr,
                                   scalers, n_scalers, avg_deg_log, avg_deg_lin, dtype, nullptr, 0, s);
}

extern "C" int gnnops_edge_reduce_hubs(int functor, const void* q, int64_t ldq, const void* p, int64_t ldp, const void* w,
                                       int64_t ldw, const void* add, int64_t ldadd, const int32_t* rowptr, const int32_t* perm,
                                       const int64_t* col, void* out, int64_t ldo, int64_t N, int64_t E, int64_t K, const int* aggr,
                                       int n_aggr, const int* scalers, int n_scalers, float avg_deg_log, float avg_deg_lin, int dtype,
                                       void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t s) {
    GNNOPS_REQUIRE(N >= 0 && E >= 0 && K >= 0, GNNOPS_EINVAL, "edge_reduce: negative size");
    GNNOPS_REQUIRE(N < ((int64_t)1 << 31) && E < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "edge_reduce: N and E must be < 2^31");
    GNNOPS_REQUIRE(K < ((int64_t)1 << 24) && ldq < ((int64_t)1 << 31) && ldw < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED,
                   "edge_reduce: rows of 2^24 or more elements");
    GNNOPS_REQUIRE(functor >= F_COPY && functor <= F_FILM, GNNOPS_EINVAL, "edge_reduce: unknown functor %d", functor);
    GNNOPS_REQUIRE(n_aggr >= 1 && n_aggr <= 5 && aggr, GNNOPS_EINVAL, "edge_reduce: 1..5 aggregators");
    GNNOPS_REQUIRE(n_scalers >= 0 && n_scalers <= 5 && (n_scalers == 0 || scalers), GNNOPS_EINVAL, "edge_reduce: 0..5 scalers");
    if (N * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && out && (E == 0 || (q && col)), GNNOPS_EINVAL, "edge_reduce: null pointer");
    GNNOPS_REQUIRE(functor == F_COPY || p, GNNOPS_EINVAL, "edge_reduce: this message needs the per-destination rows p");
    if (functor == F_COPY || functor == F_FILM) w = nullptr;   // these messages have no per-edge term
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
    // widest aligned piece (elements) every operand row allows: pointers and pitches in bytes share a power-of-two factor
    int max_vec = vec;
    auto narrow = [&](const void* ptr, int64_t ld) {
        if (!ptr) return;
        while (max_vec > 1 && ((uintptr_t)ptr % (max_vec * es) != 0 || (ld * es) % (max_vec * es) != 0)) max_vec >>= 1;
    };
    narrow(q, ldq); narrow(p, ldp); narrow(w, ldw); narrow(add, ldadd); narrow(out, ldo);
    if ((K * es) % (max_vec * es) != 0)
        while (max_vec > 1 && K % max_vec != 0) max_vec >>= 1;
    if (hub_workspace && E > 4 * T_HUB) {   // destinations with more than T_HUB edges are reduced piecewise (sums re-associated)
        const HubLayout l = hub_layout(E, K);
        GNNOPS_REQUIRE(hub_workspace_bytes >= l.total && (uintptr_t)hub_workspace % 16 == 0, GNNOPS_EWORKSPACE,
                       "edge_reduce: hub workspace %zu < %zu", hub_workspace_bytes, l.total);
        char* wsp = (char*)hub_workspace;
        a.hub_count = (unsigned int*)(wsp + l.counters);
        a.hub_rows = (int32_t*)(wsp + l.rows);
        a.piece_base = (int32_t*)(wsp + l.base);
        a.partial = (float*)(wsp + l.partial);
        a.max_hubs = l.max_hubs;
    }
    hipStream_t stream = (hipStream_t)s;
    switch (dtype) {
        case GNNOPS_F32: return dispatch<float>(functor, multi, a, max_vec, stream);
        case GNNOPS_F16: return dispatch<__half>(functor, multi, a, max_vec, stream);
        default: return dispatch<__hip_bfloat16>(functor, multi, a, max_vec, stream);
    }
}

// ---- backward of the edge pass (training through gnnops.conv; the reference's OpProfiler.py:259-292 profiles a TRAIN loop) ----
// For sum / mean aggregation the chain rule needs, per edge e = (j -> i), the gradient of the message with respect to the
// rows it was made from; the sums of those per-edge rows by destination (d p) and by source (d q) are then plain segment
// reductions over the plans the forward pass already holds. This kernel writes the per-edge rows, in EDGE order (so d w is
// the rows themselves). Streaming: each edge's operand rows are gathered once, each gradient row stored once.
namespace {
template <typename T, int F, bool HAS_W, int VEC>
__global__ __launch_bounds__(256) void edge_grad_kernel(const T* __restrict__ p, int64_t ldp, const T* __restrict__ q, int64_t ldq,
                                                        const T* __restrict__ w, int64_t ldw, const T* __restrict__ g, int64_t ldg,
                                                        const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                        T* __restrict__ gp, T* __restrict__ gq, int64_t E, int64_t K) {
    constexpr bool PRECISE = sizeof(T) == 4;
    const int64_t kv = K / VEC;
    const int64_t items = E * kv;
    for (int64_t chunk = blockIdx.x; chunk * 256 < items; chunk += gridDim.x) {   // a workgroup step = 256 consecutive pieces
        const int64_t item = chunk * 256 + threadIdx.x;
        if (item >= items) continue;
        const int64_t e = item / kv;
        const int64_t c = (item - e * kv) * VEC;
        const int64_t i = dst[e], j = src[e];
        float gv[VEC];
        load_vec<T, VEC>(g + i * ldg + c, gv);
        if constexpr (F == F_CGCONV) {
            float pf[VEC], ps[VEC], qf[VEC], qs[VEC], wf[VEC], ws[VEC], of[VEC], os[VEC];
            load_vec<T, VEC>(p + i * ldp + c, pf);
            load_vec<T, VEC>(p + i * ldp + K + c, ps);
            load_vec<T, VEC>(q + j * ldq + c, qf);
            load_vec<T, VEC>(q + j * ldq + K + c, qs);
            if constexpr (HAS_W) {
                load_vec<T, VEC>(w + e * ldw + c, wf);
                load_vec<T, VEC>(w + e * ldw + K + c, ws);
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float zf = pf[v] + qf[v], zs = ps[v] + qs[v];
                if constexpr (HAS_W) { zf = zf + wf[v]; zs = zs + ws[v]; }
                const float sg = sigmoid_f(zf), sp = softplus_f<PRECISE>(zs);
                of[v] = gv[v] * sp * (sg * (1.f - sg));       // d/dz_f  sigmoid(z_f) softplus(z_s)
                os[v] = gv[v] * sg * sigmoid_f(zs);            // d/dz_s: softplus' = sigmoid
            }
            store_vec<T, VEC>(gp + e * 2 * K + c, of);
            store_vec<T, VEC>(gp + e * 2 * K + K + c, os);
        } else {   // F_FILM: p = [beta | gamma], message relu(gamma * q + beta)
            float be[VEC], ga[VEC], qv[VEC], ob[VEC], og[VEC], oq[VEC];
            load_vec<T, VEC>(p + i * ldp + c, be);
            load_vec<T, VEC>(p + i * ldp + K + c, ga);
            load_vec<T, VEC>(q + j * ldq + c, qv);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float a = ga[v] * qv[v] + be[v];
                const float gm = a > 0.f ? gv[v] : 0.f;
                ob[v] = gm;
                og[v] = gm * qv[v];
                oq[v] = gm * ga[v];
            }
            store_vec<T, VEC>(gp + e * 2 * K + c, ob);
            store_vec<T, VEC>(gp + e * 2 * K + K + c, og);
            store_vec<T, VEC>(gq + e * K + c, oq);
        }
    }
}

template <typename T, int F, bool HAS_W>
int launch_edge_grad(const void* p, int64_t ldp, const void* q, int64_t ldq, const void* w, int64_t ldw, const void* g, int64_t ldg,
                     const int64_t* src, const int64_t* dst, void* gp, void* gq, int64_t E, int64_t K, int vec, hipStream_t stream) {
    const int64_t items = E * (K / vec);
    const int grid = gnnops_grid_cap(gnnops_cdiv(items, 256), 256 * 32);
#define GNNOPS_EG(V)                                                                                                              \
    hipLaunchKernelGGL((edge_grad_kernel<T, F, HAS_W, V>), dim3(grid), dim3(256), 0, stream, (const T*)p, ldp, (const T*)q, ldq, \
                       (const T*)w, ldw, (const T*)g, ldg, src, dst, (T*)gp, (T*)gq, E, K)
    if (vec == Elem<T>::VEC) GNNOPS_EG(Elem<T>::VEC);
    else GNNOPS_EG(1);
#undef GNNOPS_EG
    return gnnops_check_launch("edge_grad");
}

template <typename T>
int dispatch_edge_grad(int functor, const void* p, int64_t ldp, const void* q, int64_t ldq, const void* w, int64_t ldw, const void* g,
                       int64_t ldg, const int64_t* src, const int64_t* dst, void* gp, void* gq, int64_t E, int64_t K, int vec,
                       hipStream_t stream) {
    if (functor == F_CGCONV)
        return w ? launch_edge_grad<T, F_CGCONV, true>(p, ldp, q, ldq, w, ldw, g, ldg, src, dst, gp, gq, E, K, vec, stream)
                 : launch_edge_grad<T, F_CGCONV, false>(p, ldp, q, ldq, w, ldw, g, ldg, src, dst, gp, gq, E, K, vec, stream);
    return launch_edge_grad<T, F_FILM, false>(p, ldp, q, ldq, w, ldw, g, ldg, src, dst, gp, gq, E, K, vec, stream);
}
}  // namespace

extern "C" int gnnops_edge_grad(int functor, const void* p, int64_t ldp, const void* q, int64_t ldq, const void* w, int64_t ldw,
                                const void* g, int64_t ldg, const int64_t* src, const int64_t* dst, void* gp, void* gq, int64_t E,
                                int64_t K, int dtype, gnnops_stream_t s) {
    GNNOPS_REQUIRE(functor == F_CGCONV || functor == F_FILM, GNNOPS_EUNSUPPORTED,
                   "edge_grad: functor %d has no per-edge gradient kernel (copy / add: the output gradient itself)", functor);
    GNNOPS_REQUIRE(E >= 0 && K >= 0, GNNOPS_EINVAL, "edge_grad: negative size");
    if (E == 0 || K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(p && q && g && src && dst && gp && (functor == F_CGCONV || gq), GNNOPS_EINVAL, "edge_grad: null pointer");
    GNNOPS_REQUIRE(ldp >= 2 * K && ldq >= (functor == F_CGCONV ? 2 : 1) * K && ldg >= K && (!w || ldw >= 2 * K), GNNOPS_EINVAL,
                   "edge_grad: a row pitch is shorter than the row");
    int es, vec;
    switch (dtype) {
        case GNNOPS_F32: es = 4; vec = 4; break;
        case GNNOPS_F16: case GNNOPS_BF16: es = 2; vec = 8; break;
        default: gnnops_set_error("edge_grad: unknown dtype %d", dtype); return GNNOPS_EINVAL;
    }
    bool wide = K % vec == 0;   // 16-B pieces need every row start 16-B aligned: pointers, pitches and the K offset of a part
    auto ok = [&](const void* ptr, int64_t ld) { return !ptr || ((uintptr_t)ptr % 16 == 0 && (ld * es) % 16 == 0); };
    wide = wide && ok(p, ldp) && ok(q, ldq) && ok(w, ldw) && ok(g, ldg) && ok(gp, 2 * K) && ok(gq, K) && (K * es) % 16 == 0;
    const int v = wide ? vec : 1;
    hipStream_t stream = (hipStream_t)s;
    switch (dtype) {
        case GNNOPS_F32: return dispatch_edge_grad<float>(functor, p, ldp, q, ldq, w, ldw, g, ldg, src, dst, gp, gq, E, K, v, stream);
        case GNNOPS_F16: return dispatch_edge_grad<__half>(functor, p, ldp, q, ldq, w, ldw, g, ldg, src, dst, gp, gq, E, K, v, stream);
        default: return dispatch_edge_grad<__hip_bfloat16>(functor, p, ldp, q, ldq, w, ldw, g, ldg, src, dst, gp, gq, E, K, v, stream);
    }
}
