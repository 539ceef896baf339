/*
 * gnnops_oracle.c — CPU restatement of the reference's op semantics. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (gnn-ops-benchmark_amd/) never does and fails loudly without its HIP library.
 *
 * What is restated, and from where (paths under the reference root):
 *   - native ATen ops called by the op_* bodies: Tensor.scatter_add_ (benchmark_scatter_add.py:22-25),
 *     torch.index_select (benchmark_native_index_select.py:12-15), Tensor.index_add_
 *     (benchmark_native_index_add_.py:13-16), torch.gather (benchmark_native_gather.py:14-17),
 *     scatter_(reduce="multiply") (benchmark_scatter_multiply.py:42-45). The algorithm lives in PyTorch
 *     (requirements.txt:209 pins torch==1.11.0), not in the reference tree; torch 2.10 CPU is importable
 *     in the build container and PINS these functions through tests/golden/ (make_golden.py).
 *   - torch_scatter 2.0.9 ops (requirements.txt:212): scatter_add/sum, scatter_mean, scatter_min,
 *     scatter_max, scatter_mul (call sites benchmark_scatter_add.py:18, benchmark_scatter_mean.py:17,
 *     benchmark_scatter_min.py:17, benchmark_scatter_max.py:17). The package is absent everywhere we run
 *     and the reference holds no test or fixture for it: its published semantics are
 *     restated here and cross-checked against torch's scatter_reduce_; arg_out tie-breaking (first
 *     position) is PARITY UNPINNED.
 *
 * Every reduction is a sequential loop over the source positions e = 0..E-1 in fp32, so the result is
 * the order-of-operations a single-threaded CPU scatter produces. 16-bit inputs are widened to fp32,
 * accumulated in fp32 and rounded ONCE on store (the comparison target stated in SURVEY.md §8c).
 *
 * Shapes follow include/gnnops.h: src [B,E,K], out [B,N,K]; index layout R ([E]) or F ([B,E,K]).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORA_F32 = 0, ORA_F16 = 1, ORA_BF16 = 2 };
enum { ORA_SUM = 0, ORA_MEAN = 1, ORA_MIN = 2, ORA_MAX = 3, ORA_MUL = 4 };

/* ---- 16-bit float conversions (round to nearest even), no compiler extensions ---- */
static float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) {
            u = sign;
        } else { /* subnormal */
            int sh = 0;
            while (!(man & 0x400u)) { man <<= 1; ++sh; }
            man &= 0x3ffu;
            u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7f800000u | (man << 13);
    } else {
        u = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t f32_to_f16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t abs = u & 0x7fffffffu;
    if (abs > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);      /* NaN */
    if (abs >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);     /* rounds to inf (>= 65520) */
    if (abs < 0x33000001u) return (uint16_t)sign;                  /* rounds to zero (<= 2^-25) */
    int32_t e = (int32_t)(abs >> 23) - 127;
    uint32_t m = (abs & 0x7fffffu) | 0x800000u;
    if (e < -14) { /* subnormal half */
        int shift = -14 - e + 13; /* 14..24 */
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3ffu);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r; /* carries into exponent correctly */
    return (uint16_t)(sign | r);
}

static float ld(const void* p, int64_t i, int dt) {
    if (dt == ORA_F32) return ((const float*)p)[i];
    if (dt == ORA_F16) return f16_to_f32(((const uint16_t*)p)[i]);
    return bf16_to_f32(((const uint16_t*)p)[i]);
}
static void st(void* p, int64_t i, int dt, float v) {
    if (dt == ORA_F32) ((float*)p)[i] = v;
    else if (dt == ORA_F16) ((uint16_t*)p)[i] = f32_to_f16(v);
    else ((uint16_t*)p)[i] = f32_to_bf16(v);
}

/* exported for the conversion unit tests */
float ora_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
uint16_t ora_f32_to_f16(float f) { return f32_to_f16(f); }
float ora_bf16_to_f32(uint16_t h) { return bf16_to_f32(h); }
uint16_t ora_f32_to_bf16(float f) { return f32_to_bf16(f); }

/*
 * scatter family. index_full = 0: index[e] (layout R); 1: index[(b*E+e)*K+k] (layout F).
 * init_from_out = 0: torch_scatter without `out=` (zeros / identity, empty min/max groups -> 0,
 * arg = E); 1: combine into the given out (index_add_, scatter_add_, `out=`).
 * Returns 0, or 1 on an out-of-range index.
 */
int ora_scatter(const void* src, const int64_t* index, void* out, int64_t* arg_out, int64_t B, int64_t E, int64_t K,
                int64_t N, int dtype, int reduce, int index_full, int init_from_out) {
    const int64_t nout = B * N * K;
    float* acc = (float*)malloc(sizeof(float) * (size_t)(nout > 0 ? nout : 1));
    int32_t* cnt = NULL;
    if (!acc) return 2;
    if (reduce == ORA_MEAN) {
        cnt = (int32_t*)calloc((size_t)(nout > 0 ? nout : 1), sizeof(int32_t));
        if (!cnt) { free(acc); return 2; }
    }
    for (int64_t i = 0; i < nout; ++i) {
        if (init_from_out) acc[i] = ld(out, i, dtype);
        else if (reduce == ORA_MUL) acc[i] = 1.0f;
        else if (reduce == ORA_MIN) acc[i] = INFINITY;
        else if (reduce == ORA_MAX) acc[i] = -INFINITY;
        else acc[i] = 0.0f;
        if (arg_out) arg_out[i] = E;
    }
    for (int64_t b = 0; b < B; ++b)
        for (int64_t e = 0; e < E; ++e)
            for (int64_t k = 0; k < K; ++k) {
                const int64_t s = (b * E + e) * K + k;
                const int64_t n = index_full ? index[s] : index[e];
                if (n < 0 || n >= N) { free(acc); free(cnt); return 1; }
                const int64_t d = (b * N + n) * K + k;
                const float v = ld(src, s, dtype);
                switch (reduce) {
                    case ORA_SUM: acc[d] = acc[d] + v; break;
                    case ORA_MEAN: acc[d] = acc[d] + v; cnt[d] += 1; break;
                    case ORA_MUL: acc[d] = acc[d] * v; break;
                    case ORA_MIN: if (v < acc[d]) { acc[d] = v; if (arg_out) arg_out[d] = e; } break;
                    case ORA_MAX: if (v > acc[d]) { acc[d] = v; if (arg_out) arg_out[d] = e; } break;
                }
            }
    for (int64_t i = 0; i < nout; ++i) {
        float v = acc[i];
        if (reduce == ORA_MEAN) v = v / (float)(cnt[i] < 1 ? 1 : cnt[i]);
        if ((reduce == ORA_MIN || reduce == ORA_MAX) && !init_from_out) {
            /* torch_scatter: out.masked_fill_(arg_out == src.size(dim), 0) */
            const int reached = arg_out ? (arg_out[i] != E) : (v != (reduce == ORA_MIN ? INFINITY : -INFINITY));
            if (!reached) v = 0.0f;
        }
        st(out, i, dtype, v);
    }
    free(acc);
    free(cnt);
    return 0;
}

/* torch.index_select along the middle axis: out[b,e,k] = in[b,index[e],k]; byte copy. */
int ora_index_select(const void* in, const int64_t* index, void* out, int64_t B, int64_t N, int64_t K, int64_t E,
                     int elem_bytes) {
    const size_t row = (size_t)K * (size_t)elem_bytes;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t e = 0; e < E; ++e) {
            const int64_t n = index[e];
            if (n < 0 || n >= N) return 1;
            memcpy((char*)out + (size_t)(b * E + e) * row, (const char*)in + (size_t)(b * N + n) * row, row);
        }
    return 0;
}

/* torch.gather: out[b,e,k] = in[b,index[b,e,k],k]; byte copy. */
int ora_gather(const void* in, const int64_t* index, void* out, int64_t B, int64_t N, int64_t K, int64_t E,
               int elem_bytes) {
    for (int64_t b = 0; b < B; ++b)
        for (int64_t e = 0; e < E; ++e)
            for (int64_t k = 0; k < K; ++k) {
                const int64_t o = (b * E + e) * K + k;
                const int64_t n = index[o];
                if (n < 0 || n >= N) return 1;
                memcpy((char*)out + (size_t)o * elem_bytes, (const char*)in + (size_t)((b * N + n) * K + k) * elem_bytes,
                       (size_t)elem_bytes);
            }
    return 0;
}

/* Stable counting sort of index by value: rowptr[N+1], perm[E] (ascending e inside a segment). */
int ora_plan(const int64_t* index, int64_t E, int64_t N, int32_t* rowptr, int32_t* perm) {
    memset(rowptr, 0, sizeof(int32_t) * (size_t)(N + 1));
    for (int64_t e = 0; e < E; ++e) {
        if (index[e] < 0 || index[e] >= N) return 1;
        rowptr[index[e] + 1] += 1;
    }
    for (int64_t n = 0; n < N; ++n) rowptr[n + 1] += rowptr[n];
    int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    if (!cur) return 2;
    memcpy(cur, rowptr, sizeof(int32_t) * (size_t)N);
    for (int64_t e = 0; e < E; ++e) perm[cur[index[e]]++] = (int32_t)e;
    free(cur);
    return 0;
}

/* index_select(...).sum() with a double accumulator (the fp32 device sum is compared with a tolerance). */
int ora_index_select_sum(const void* in, const int64_t* index, double* out_sum, int64_t B, int64_t N, int64_t K,
                         int64_t E, int dtype) {
    double s = 0.0;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t e = 0; e < E; ++e) {
            const int64_t n = index[e];
            if (n < 0 || n >= N) return 1;
            for (int64_t k = 0; k < K; ++k) s += (double)ld(in, (b * N + n) * K + k, dtype);
        }
    *out_sum = s;
    return 0;
}

/* ---- row-major R-layout scatter_add, fp32, tuned only as far as a plain C port goes: the CPU baseline
 * timed by bench.py (cores = 1). Same arithmetic as ora_scatter(SUM, R). ---- */
int ora_scatter_add_rows_f32(const float* src, const int64_t* index, float* out, int64_t E, int64_t K, int64_t N) {
    memset(out, 0, sizeof(float) * (size_t)(N * K));
    for (int64_t e = 0; e < E; ++e) {
        const int64_t n = index[e];
        if (n < 0 || n >= N) return 1;
        float* o = out + n * K;
        const float* s = src + e * K;
        for (int64_t k = 0; k < K; ++k) o[k] += s[k];
    }
    return 0;
}

/*
 * torch_sparse.spmm restated: out = scatter_add(matrix[col] * value[:,None], row, dim_size=m),
 * sequential over the entries e = 0..nnz-1. fp32: p = round(v*x) then acc = round(acc+p) (two roundings, as
 * torch's elementwise multiply followed by scatter_add performs); 16-bit: products are exact in fp32,
 * accumulated in fp32, rounded once. value == NULL means ones. Reference call site:
 * op_bm_scripts/benchmark_sparse_spmm.py:12-14 (torch.sparse.mm form).
 */
int ora_spmm(const int64_t* row, const int64_t* col, const void* value, const void* mat, void* out, int64_t nnz,
             int64_t M, int64_t Ncols, int64_t D, int dtype) {
    float* acc = (float*)calloc((size_t)(M * D > 0 ? M * D : 1), sizeof(float));
    if (!acc) return 2;
    for (int64_t e = 0; e < nnz; ++e) {
        const int64_t r = row[e], c = col[e];
        if (r < 0 || r >= M || c < 0 || c >= Ncols) { free(acc); return 1; }
        const float v = value ? ld(value, e, dtype) : 1.0f;
        for (int64_t k = 0; k < D; ++k) {
            volatile float p = v * ld(mat, c * D + k, dtype); /* volatile: keep the product rounded, no FMA */
            acc[r * D + k] = acc[r * D + k] + p;
        }
    }
    for (int64_t i = 0; i < M * D; ++i) st(out, i, dtype, acc[i]);
    free(acc);
    return 0;
}

/*
 * Run reduction for coalesce: entries already ordered by `perm` (stable sort by row*n+col, done in numpy);
 * seg_start[u] is the first sorted position of distinct key u. out[u,c] = sum in sorted order (fp32).
 * Restates torch_sparse.coalesce's scatter_add over the sorted entries (call site
 * op_bm_scripts/benchmark_sparse_coalesce.py:35-37).
 */
int ora_reduce_runs(const void* value, const int64_t* perm, const int64_t* seg_start, int64_t count, int64_t nnz,
                    int64_t C, int dtype, void* out) {
    for (int64_t u = 0; u < count; ++u) {
        const int64_t beg = seg_start[u], end = (u + 1 < count) ? seg_start[u + 1] : nnz;
        for (int64_t c = 0; c < C; ++c) {
            float acc = 0.0f;
            for (int64_t p = beg; p < end; ++p) acc = acc + ld(value, perm[p] * C + c, dtype);
            st(out, u * C + c, dtype, acc);
        }
    }
    return 0;
}

/* addmm / matmul (benchmark_native_addmm.py:13-16): out = input + A[M,K] @ B[K,N], double accumulation
 * (the device sums inside the MFMA in fp32; compared with a tolerance that scales with sqrt(K)). */
int ora_addmm(const void* input, const void* A, const void* B, double* out, int64_t M, int64_t N, int64_t K, int dtype) {
    for (int64_t i = 0; i < M; ++i)
        for (int64_t j = 0; j < N; ++j) out[i * N + j] = input ? (double)ld(input, i * N + j, dtype) : 0.0;
    for (int64_t i = 0; i < M; ++i)
        for (int64_t k = 0; k < K; ++k) {
            const double a = (double)ld(A, i * K + k, dtype);
            for (int64_t j = 0; j < N; ++j) out[i * N + j] += a * (double)ld(B, k * N + j, dtype);
        }
    return 0;
}

/* index_select(index_add(input, dim, index, other), dim, index).sum(dim)
 * (benchmark_fused_index_add_reduce.py:12-20) restated literally on [B,N,K] / [B,E,K]: tmp = input with
 * other accumulated in fp32 per destination in order and rounded once to the storage type (the
 * index_add convention of this oracle), then a double sum over the selected rows. */
int ora_index_add_select_sum(const void* input, const void* other, const int64_t* index, double* out, int64_t B,
                             int64_t N, int64_t E, int64_t K, int dtype) {
    float* tmp = (float*)malloc(sizeof(float) * (size_t)(B * N * K > 0 ? B * N * K : 1));
    if (!tmp) return 2;
    for (int64_t i = 0; i < B * N * K; ++i) tmp[i] = ld(input, i, dtype);
    for (int64_t b = 0; b < B; ++b)
        for (int64_t e = 0; e < E; ++e) {
            if (index[e] < 0 || index[e] >= N) { free(tmp); return 1; }
            for (int64_t k = 0; k < K; ++k) tmp[(b * N + index[e]) * K + k] += ld(other, (b * E + e) * K + k, dtype);
        }
    if (dtype != ORA_F32) { /* one rounding to the storage type */
        uint16_t h;
        for (int64_t i = 0; i < B * N * K; ++i) { st(&h, 0, dtype, tmp[i]); tmp[i] = ld(&h, 0, dtype); }
    }
    for (int64_t b = 0; b < B; ++b)
        for (int64_t k = 0; k < K; ++k) {
            double s = 0.0;
            for (int64_t e = 0; e < E; ++e) s += (double)tmp[(b * N + index[e]) * K + k];
            out[b * K + k] = s;
        }
    free(tmp);
    return 0;
}
