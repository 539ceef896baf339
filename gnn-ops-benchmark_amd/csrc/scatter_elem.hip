// scatter_elem.hip — element-wise scatter for a full-shape index (layout F): the literal shapes the
// reference scripts time — torch_scatter.scatter_*(src, idx, dim) with idx.shape == src.shape
// (op_bm_scripts/benchmark_scatter_add.py:67,78-84), `zeros_like(src).scatter_add_(dim, idx, src)`
// (benchmark_scatter_add.py:22-25) and `scatter_(-1, idx, src, reduce="multiply")`
// (benchmark_scatter_multiply.py:42-45).
//
// Every element has its own destination, so there is no row structure to sort on; this is the atomic
// path. fp32 adds are single `global_atomic_add_f32` instructions; 16-bit sums/products accumulate in
// an fp32 scratch and are rounded once; min/max (and products) use a CAS loop on the 32-bit word. The arg pass
// picks the smallest position among ties (atomicMin on int64), which is what a sequential CPU loop gives.
#include "common.h"

namespace {

__device__ inline void decode(int64_t o, int64_t E, int64_t K, int64_t& b, int64_t& e, int64_t& k) {
    k = o % K;
    const int64_t be = o / K;
    e = be % E;
    b = be / E;
}

#define GRID_STRIDE(o, total) \
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < (total); o += (int64_t)gridDim.x * blockDim.x)

template <typename T>
__global__ void fill_kernel(T* p, int64_t n, float v) {
    GRID_STRIDE(i, n) Elem<T>::store(p + i, v);
}
__global__ void fill_i64_kernel(int64_t* p, int64_t n, int64_t v) { GRID_STRIDE(i, n) p[i] = v; }

// ---- sum ----
template <typename T>
__global__ void scatter_add_f32acc_kernel(const T* __restrict__ src, const int64_t* __restrict__ index,
                                          float* __restrict__ acc, int64_t B, int64_t E, int64_t K, int64_t N) {
    GRID_STRIDE(o, B * E * K) {
        int64_t b, e, k;
        decode(o, E, K, b, e, k);
        atomicAdd(acc + (b * N + index[o]) * K + k, Elem<T>::load(src + o));
    }
}
__global__ void count_kernel(const int64_t* __restrict__ index, float* __restrict__ cnt, int64_t B, int64_t E,
                             int64_t K, int64_t N) {
    GRID_STRIDE(o, B * E * K) {
        int64_t b, e, k;
        decode(o, E, K, b, e, k);
        atomicAdd(cnt + (b * N + index[o]) * K + k, 1.0f);
    }
}
// out = round(acc [/ max(cnt,1)])
template <typename T>
__global__ void finish_kernel(T* __restrict__ out, const float* __restrict__ acc, const float* __restrict__ cnt,
                              int64_t n) {
    GRID_STRIDE(i, n) {
        float v = acc[i];
        if (cnt) {
            const float c = cnt[i];
            v = v / (c < 1.f ? 1.f : c);
        }
        Elem<T>::store(out + i, v);
    }
}
template <typename T>
__global__ void widen_kernel(const T* __restrict__ in, float* __restrict__ acc, int64_t n) {
    GRID_STRIDE(i, n) acc[i] = Elem<T>::load(in + i);
}

// ---- CAS-based update of one element (fp32 word or a 16-bit half of a word) ----
template <typename T, typename F>
__device__ inline void atomic_update(T* addr, float v, F f) {
    if constexpr (sizeof(T) == 4) {
        unsigned int* w = reinterpret_cast<unsigned int*>(addr);
        unsigned int old = *w;
        while (true) {
            const float cur = __uint_as_float(old);
            const float nv = f(cur, v);
            if (__float_as_uint(nv) == old) break;
            const unsigned int got = atomicCAS(w, old, __float_as_uint(nv));
            if (got == old) break;
            old = got;
        }
    } else {
        const uintptr_t a = reinterpret_cast<uintptr_t>(addr);
        unsigned int* w = reinterpret_cast<unsigned int*>(a & ~(uintptr_t)3);
        const int sh = (a & 2) ? 16 : 0;
        unsigned int old = *w;
        while (true) {
            unsigned short bits = (unsigned short)(old >> sh);
            T curT = *reinterpret_cast<T*>(&bits);
            const float nv = f(Elem<T>::load(&curT), v);
            T nT;
            Elem<T>::store(&nT, nv);
            const unsigned short nbits = *reinterpret_cast<unsigned short*>(&nT);
            if (nbits == bits) break;
            const unsigned int neww = (old & ~(0xffffu << sh)) | ((unsigned int)nbits << sh);
            const unsigned int got = atomicCAS(w, old, neww);
            if (got == old) break;
            old = got;
        }
    }
}

template <typename T, int R>
__global__ void scatter_cas_kernel(const T* __restrict__ src, const int64_t* __restrict__ index, T* __restrict__ out,
                                   int64_t B, int64_t E, int64_t K, int64_t N) {
    GRID_STRIDE(o, B * E * K) {
        int64_t b, e, k;
        decode(o, E, K, b, e, k);
        T* dst = out + (b * N + index[o]) * K + k;
        const float v = Elem<T>::load(src + o);
        if constexpr (R == GNNOPS_MIN) {
            atomic_update(dst, v, [](float a, float x) { return x < a ? x : a; });
        } else {
            atomic_update(dst, v, [](float a, float x) { return x > a ? x : a; });
        }
    }
}

template <typename T>
__global__ void scatter_mul_f32acc_kernel(const T* __restrict__ src, const int64_t* __restrict__ index,
                                          float* __restrict__ acc, int64_t B, int64_t E, int64_t K, int64_t N) {
    GRID_STRIDE(o, B * E * K) {
        int64_t b, e, k;
        decode(o, E, K, b, e, k);
        atomic_update(acc + (b * N + index[o]) * K + k, Elem<T>::load(src + o), [](float a, float x) { return a * x; });
    }
}

// arg pass: smallest e whose value equals the reduced value. A contribution equal to the reduce's identity (+inf for min,
// -inf for max) never takes a slot — the sequential loop replaces on a strict improvement only — so a group that
// received nothing else stays "empty" (arg = E, value 0), exactly as in the row kernels (segment.hip, bucket.hip).
template <typename T>
__global__ void scatter_arg_kernel(const T* __restrict__ src, const int64_t* __restrict__ index,
                                   const T* __restrict__ out, int64_t* __restrict__ arg_out, int64_t B, int64_t E,
                                   int64_t K, int64_t N, float ident) {
    GRID_STRIDE(o, B * E * K) {
        int64_t b, e, k;
        decode(o, E, K, b, e, k);
        const int64_t d = (b * N + index[o]) * K + k;
        const float v = Elem<T>::load(src + o);
        if (v != ident && v == Elem<T>::load(out + d))
            atomicMin(reinterpret_cast<unsigned long long*>(arg_out + d), (unsigned long long)e);
    }
}
// torch_scatter: groups nothing reached (arg == E) become 0.
template <typename T>
__global__ void zero_empty_kernel(T* __restrict__ out, const int64_t* __restrict__ arg_out, int64_t n, int64_t E) {
    GRID_STRIDE(i, n) if (arg_out[i] == E) Elem<T>::store(out + i, 0.f);
}

// ---- LDS-privatised form ---------------------------------------------------------------------------------
// When all destinations of one (b, column strip) fit in LDS — N * TC accumulators — a workgroup owns the
// strip: it streams its part of src / index once (coalesced along the strip), combines into LDS with LDS
// atomics (two orders of magnitude cheaper than memory-side atomics, MI355X_MICROARCH.md "Global float
// atomics"), and writes each output element exactly once. The reference's own shapes ((L, L) fp16, index of
// the full shape, dim 0 or 1, L <= 6708) all take this path: HBM-bound on the 8-B index.
constexpr int LDS_THREADS = 1024;
constexpr size_t LDS_BUDGET = 160 * 1024 - 512;

template <typename T, int R, typename I>
__global__ __launch_bounds__(LDS_THREADS) void scatter_lds_kernel(const T* __restrict__ src,
                                                                  const I* __restrict__ index,
                                                                  T* __restrict__ out, int64_t* __restrict__ arg_out,
                                                                  int64_t B, int64_t E, int64_t K, int64_t N, int TC,
                                                                  int strips, int64_t rows, int nchunks,
                                                                  int init_from_out, int tshift) {
    // item = (b * strips + strip) * nchunks + chunk; the chunk owns destinations [n_lo, n_lo + nloc)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* acc = reinterpret_cast<float*>(lds_raw);
    int* aux = reinterpret_cast<int*>(lds_raw) + (size_t)rows * TC;  // counts (MEAN) or arg (MIN/MAX)
    constexpr bool IS_ARG = (R == GNNOPS_MIN || R == GNNOPS_MAX);
    constexpr int UNR = 8;  // loads in flight per thread: the strip is streamed, not chased
    const int64_t item = xcd_contiguous(blockIdx.x, gridDim.x);
    const int chunk = (int)(item % nchunks);
    const int64_t bs = item / nchunks;
    const int64_t b = bs / strips;
    const int64_t k0 = (int64_t)(bs % strips) * TC;
    const int tc = (int)((K - k0 < TC) ? (K - k0) : TC);
    const int64_t n_lo = (int64_t)chunk * rows;
    const int nloc = (int)((N - n_lo < rows) ? (N - n_lo) : rows);
    const float ident = (R == GNNOPS_MUL) ? 1.f : (R == GNNOPS_MIN) ? __builtin_huge_valf()
                        : (R == GNNOPS_MAX) ? -__builtin_huge_valf() : 0.f;
    // thread -> (row slot er, column kk of the strip): kk = tid mod 2^tshift (2^tshift >= TC), no divisions anywhere
    const int kk = threadIdx.x & ((1 << tshift) - 1);
    const int er = threadIdx.x >> tshift;
    const int rpi = (int)blockDim.x >> tshift;  // rows per sweep of the workgroup
    const bool col_ok = kk < tc;

    if (col_ok) {
        for (int r0 = er; r0 < nloc; r0 += rpi * UNR) {
            float iv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {  // unconditional, clamped loads: all in flight together (see below)
                const int r = r0 + u * rpi;
                const int rc = r < nloc ? r : nloc - 1;
                iv[u] = init_from_out ? Elem<T>::load(out + (b * N + n_lo + rc) * K + k0 + kk) : ident;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = r0 + u * rpi;
                if (r < nloc) {
                    const int i = r * tc + kk;
                    acc[i] = iv[u];
                    if (R == GNNOPS_MEAN) aux[i] = 0;
                    if (IS_ARG) aux[i] = (int)E;
                }
            }
        }
    }
    __syncthreads();

    const T* sp = src + (b * E) * K + k0 + kk;
    const I* ip = index + (b * E) * K + k0 + kk;
    // one element into its LDS cell (dl = destination relative to this chunk; past the end / another chunk's: dropped)
    auto feed = [&](int64_t dl, float val) {
        if (dl < 0 || dl >= nloc) return;
        const int a = (int)dl * tc + kk;
        // the float add goes through a read + compare-and-swap on LDS, not ds_add_f32: measured on gfx950, the
        // float LDS atomic costs ~170 LDS-array cycles per wave-instruction (SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS)
        // and bounded this kernel; the integer CAS path does not
        if constexpr (R == GNNOPS_SUM) {
            atomic_update(&acc[a], val, [](float x, float y) { return x + y; });
        } else if constexpr (R == GNNOPS_MEAN) {
            atomic_update(&acc[a], val, [](float x, float y) { return x + y; });
            atomicAdd(&aux[a], 1);
        } else if constexpr (R == GNNOPS_MUL) {
            atomic_update(&acc[a], val, [](float x, float y) { return x * y; });
        } else if constexpr (R == GNNOPS_MIN) {
            atomic_update(&acc[a], val, [](float x, float y) { return y < x ? y : x; });
        } else {
            atomic_update(&acc[a], val, [](float x, float y) { return y > x ? y : x; });
        }
    };
    // K == 1, 4-byte elements: four consecutive elements per lane and load (see scatter_lds_minmax_kernel)
    struct alignas(sizeof(T) * 4) TV { T v[4]; };
    struct alignas(sizeof(I) * 4 > 16 ? 16 : sizeof(I) * 4) IV { I v[4]; };
    const bool vec4 = !IS_ARG && K == 1 && (E & 3) == 0 && sizeof(T) == 4 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)index % 16) == 0;
    if (vec4) {
        constexpr int VU = 4;
        for (int64_t e0 = (int64_t)er * 4; e0 < E; e0 += (int64_t)rpi * 4 * VU) {
            TV vt4[VU];
            IV nl4[VU];
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const int64_t e = e0 + (int64_t)u * rpi * 4;
                const int64_t ec = e < E ? e : E - 4;
                vt4[u] = *reinterpret_cast<const TV*>(sp + ec);
                nl4[u] = *reinterpret_cast<const IV*>(ip + ec);
            }
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                if (e0 + (int64_t)u * rpi * 4 >= E) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) feed((int64_t)nl4[u].v[j] - n_lo, Elem<T>::load(&vt4[u].v[j]));
            }
        }
    } else if (col_ok) {
        for (int64_t e0 = er; e0 < E; e0 += (int64_t)rpi * UNR) {
            int64_t nl[UNR];
            float v[UNR];
            // unconditional loads (rows past the end re-read row E-1 and are masked afterwards): a load under a branch
            // is waited for inside the branch, which serialises the sixteen of them
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int64_t e = e0 + (int64_t)u * rpi;
                const int64_t ec = e < E ? e : E - 1;
                nl[u] = (int64_t)ip[ec * K];
                v[u] = Elem<T>::load(sp + ec * K);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) nl[u] = (e0 + (int64_t)u * rpi < E) ? nl[u] - n_lo : -1;
#pragma unroll
            for (int u = 0; u < UNR; ++u) feed(nl[u], v[u]);
        }
    }
    __syncthreads();

    if constexpr (IS_ARG) {  // smallest position attaining the extremum
        if (col_ok) {
            for (int64_t e0 = er; e0 < E; e0 += (int64_t)rpi * UNR) {
                int64_t nl[UNR];
                float v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int64_t e = e0 + (int64_t)u * rpi;
                    const int64_t ec = e < E ? e : E - 1;
                    nl[u] = (int64_t)ip[ec * K];
                    v[u] = Elem<T>::load(sp + ec * K);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) nl[u] = (e0 + (int64_t)u * rpi < E) ? nl[u] - n_lo : -1;
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    if (nl[u] < 0 || nl[u] >= nloc) continue;
                    const int a = (int)nl[u] * tc + kk;
                    if (v[u] == acc[a]) atomicMin(&aux[a], (int)(e0 + (int64_t)u * rpi));
                }
            }
        }
        __syncthreads();
    }

    if (col_ok) {
        for (int r = er; r < nloc; r += rpi) {
            const int i = r * tc + kk;
            const int64_t o = (b * N + n_lo + r) * K + k0 + kk;
            float v = acc[i];
            if (R == GNNOPS_MEAN) v = v / (float)(aux[i] < 1 ? 1 : aux[i]);
            if (IS_ARG) {
                if (!init_from_out && aux[i] == (int)E) v = 0.f;
                if (arg_out) arg_out[o] = aux[i];
            }
            Elem<T>::store(out + o, v);
        }
    }
}

// min / max in ONE pass over the strip: the LDS cell of a destination is a 64-bit word (order-preserving image of the value
// << 32 | position), combined with one 64-bit LDS integer atomic per element — smallest value then smallest position for
// min; for max the low half holds ~position, so the largest word is the largest value at the smallest position. That is
// the extremum AND its arg in one sweep (the two-pass form above streams src / index twice and spins on a float CAS).
// -0.0 and +0.0 get the same image (they compare equal: the earlier position wins, like the sequential loop), NaNs never
// win; the stored value is re-read from src at the winning position, so its bits are exact.
__device__ inline uint32_t f32_order(float v) {
    uint32_t u = __float_as_uint(v);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// 16-bit inputs whose positions fit 16 bits (E < 65535: every reference shape) pack into a 32-bit cell — half the LDS per
// destination, so strips twice as wide. fp16 and bf16 share sign-magnitude order; both widen to fp32 exactly, so the top
// 16 bits of the fp32 image order bf16 exactly and fp16 after the exact widening (f32_order >> 16 would lose fp16 bits:
// the image is taken from the 16 stored bits instead).
__device__ inline uint32_t b16_order(uint16_t u) {
    if (u == 0x8000u) u = 0u;
    return (u & 0x8000u) ? (uint16_t)~u : (uint16_t)(u | 0x8000u);
}
template <typename CellT, typename T>
__device__ inline CellT order_image(const T* p, float v) {
    if constexpr (sizeof(CellT) == 8) return (CellT)f32_order(v);
    else return (CellT)b16_order(*reinterpret_cast<const uint16_t*>(p));
}

template <typename T, int R, typename CellT, typename I>
__global__ __launch_bounds__(LDS_THREADS) void scatter_lds_minmax_kernel(const T* __restrict__ src,
                                                                         const I* __restrict__ index,
                                                                         T* __restrict__ out, int64_t* __restrict__ arg_out,
                                                                         int64_t B, int64_t E, int64_t K, int64_t N, int TC,
                                                                         int strips, int64_t rows, int nchunks,
                                                                         int init_from_out, int tshift, int arg32) {
    static_assert(R == GNNOPS_MIN || R == GNNOPS_MAX, "min / max only");
    constexpr bool IS_MIN = R == GNNOPS_MIN;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    CellT* cell = reinterpret_cast<CellT*>(lds_raw);
    constexpr int HB = sizeof(CellT) * 4;                     // bits of each half: value image above, position below
    constexpr CellT LO_MASK = (CellT)(((CellT)1 << HB) - 1);
    constexpr int UNR = 8;
    const int64_t item = xcd_contiguous(blockIdx.x, gridDim.x);
    const int chunk = (int)(item % nchunks);
    const int64_t bs = item / nchunks;
    const int64_t b = bs / strips;
    const int64_t k0 = (int64_t)(bs % strips) * TC;
    const int tc = (int)((K - k0 < TC) ? (K - k0) : TC);
    const int64_t n_lo = (int64_t)chunk * rows;
    const int nloc = (int)((N - n_lo < rows) ? (N - n_lo) : rows);
    const int kk = threadIdx.x & ((1 << tshift) - 1);
    const int er = threadIdx.x >> tshift;
    const int rpi = (int)blockDim.x >> tshift;
    const bool col_ok = kk < tc;
    const CellT EMPTY = IS_MIN ? (CellT)~(CellT)0 : (CellT)0;
    // low half: position + 1 (min) or its complement (max); 0 / ~0 stand for "the value already in out", which therefore
    // wins a tie against any source element — the sequential loop only replaces on a strict improvement
    const CellT lo_out = IS_MIN ? (CellT)0 : LO_MASK;
    auto winner = [&](CellT c) -> int64_t {  // source position that won the cell, -1 if none did
        if (c == EMPTY && !init_from_out) return -1;
        const CellT l = (IS_MIN ? c : (CellT)~c) & LO_MASK;
        return l == 0 ? -1 : (int64_t)l - 1;
    };

    if (col_ok) {
        for (int r0 = er; r0 < nloc; r0 += rpi * UNR) {
            T it[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = r0 + u * rpi;
                const int rc = r < nloc ? r : nloc - 1;
                it[u] = init_from_out ? out[(b * N + n_lo + rc) * K + k0 + kk] : T{};
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = r0 + u * rpi;
                if (r < nloc) {
                    CellT c = EMPTY;
                    if (init_from_out) {  // a NaN in out is never replaced (nothing compares below / above it)
                        const float iv = Elem<T>::load(&it[u]);
                        c = (iv != iv) ? (IS_MIN ? (CellT)0 : (CellT)~(CellT)0)
                                       : (CellT)((order_image<CellT, T>(&it[u], iv) << HB) | lo_out);
                    }
                    cell[r * tc + kk] = c;
                }
            }
        }
    }
    __syncthreads();

    const T* sp = src + (b * E) * K + k0 + kk;
    const I* ip = index + (b * E) * K + k0 + kk;
    // one element: NaNs never win, and neither does the reduce's identity (+inf for min, -inf for max): the sequential loop
    // replaces on a strict improvement only, so a group fed nothing else stays empty — as in segment.hip
    auto feed = [&](int64_t e, int64_t dst, const T* vp) {
        const int64_t d = dst - n_lo;
        const float v = Elem<T>::load(vp);
        if (d < 0 || d >= nloc || v != v || v == (IS_MIN ? __builtin_huge_valf() : -__builtin_huge_valf())) return;
        const CellT pos = (CellT)(e + 1);
        const CellT lo = IS_MIN ? pos : (CellT)(~pos & LO_MASK);
        const CellT w = (CellT)((order_image<CellT, T>(vp, v) << HB) | lo);
        if (IS_MIN) atomicMin(&cell[(int)d * tc + kk], w); else atomicMax(&cell[(int)d * tc + kk], w);
    };
    // K == 1 (the last-dim scatter, and the transposed dim-0 route of the big shapes): a lane takes FOUR consecutive elements
    // per load — 16 B of a 4-byte src, 16 / 32 B of index — so a workgroup has 4 x the bytes in flight (one 1024-thread
    // workgroup per CU streams a 300 KB row at the latency-bound rate otherwise: (38000)^2 scatter_max)
    constexpr int VU = 4;
    struct alignas(sizeof(T) * 4) TV { T v[4]; };
    struct alignas(sizeof(I) * 4 > 16 ? 16 : sizeof(I) * 4) IV { I v[4]; };
    const bool vec4 = K == 1 && (E & 3) == 0 && sizeof(T) == 4 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)index % 16) == 0;
    if (vec4) {
        for (int64_t e0 = (int64_t)er * 4; e0 < E; e0 += (int64_t)rpi * 4 * VU) {
            TV vt4[VU];
            IV nl4[VU];
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const int64_t e = e0 + (int64_t)u * rpi * 4;
                const int64_t ec = e < E ? e : E - 4;
                vt4[u] = *reinterpret_cast<const TV*>(sp + ec);
                nl4[u] = *reinterpret_cast<const IV*>(ip + ec);
            }
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const int64_t e = e0 + (int64_t)u * rpi * 4;
                if (e >= E) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) feed(e + j, (int64_t)nl4[u].v[j], &vt4[u].v[j]);
            }
        }
    } else if (col_ok) {
        for (int64_t e0 = er; e0 < E; e0 += (int64_t)rpi * UNR) {
            int64_t nl[UNR];
            T vt[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int64_t e = e0 + (int64_t)u * rpi;
                const int64_t ec = e < E ? e : E - 1;
                nl[u] = (int64_t)ip[ec * K];
                vt[u] = sp[ec * K];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int64_t e = e0 + (int64_t)u * rpi;
                if (e < E) feed(e, nl[u], &vt[u]);
            }
        }
    }
    __syncthreads();

    if (col_ok) {
        for (int r0 = er; r0 < nloc; r0 += rpi * UNR) {
            CellT c[UNR];
            float val[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = r0 + u * rpi;
                c[u] = cell[(r < nloc ? r : nloc - 1) * tc + kk];
                val[u] = 0.f;
            }
            if constexpr (sizeof(CellT) == 8 && sizeof(T) == 4) {
                // the fp32 image is the value: undo it instead of a random 4-B read per destination. Only a zero has to be
                // looked up — +0.0 and -0.0 share an image and the stored bits are the winner's.
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const uint32_t img = (uint32_t)(c[u] >> HB);
                    const uint32_t bits = (img & 0x80000000u) ? (img & 0x7fffffffu) : ~img;
                    val[u] = __uint_as_float(bits);
                    const int64_t e = winner(c[u]);
                    if (img == 0x80000000u && e >= 0) val[u] = Elem<T>::load(sp + e * K);
                }
            } else if (E > 0) {  // uniform: the eight loads below stay together
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int64_t e = winner(c[u]);
                    val[u] = Elem<T>::load(sp + (e >= 0 ? e : 0) * K);  // unconditional; ignored when no position won
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = r0 + u * rpi;
                if (r >= nloc) continue;
                const int64_t o = (b * N + n_lo + r) * K + k0 + kk;
                const int64_t e = winner(c[u]);
                if (arg_out) {   // arg32: positions as int32 rows (E < 2^31), for a caller that widens them itself (ops.py dim-0 route)
                    if (arg32) reinterpret_cast<int32_t*>(arg_out)[o] = (int32_t)(e >= 0 ? e : E);
                    else arg_out[o] = e >= 0 ? e : E;
                }
                if (e >= 0) Elem<T>::store(out + o, val[u]);
                else if (!init_from_out) Elem<T>::store(out + o, 0.f);  // torch_scatter: groups nothing reached become 0
            }
        }
    }
}

// Geometry of the LDS form: strip width TC, destination rows per chunk, chunk count. tc == 0: does not apply.
// When all N destinations of a strip fit, there is one chunk. Otherwise the destinations are cut into chunks of
// `rows` and every chunk re-scans the strip's elements, keeping only its own (the (38000, 38000) shapes of
// data/scatter_max.csv): the re-reads are mostly cache hits and still far cheaper than memory-side atomics.
struct LdsGeom { int tc; int64_t rows; int nchunks; };
constexpr int LDS_MAX_CHUNKS = 16;

// 16-bit min / max with positions below 2^16 - 1 use 32-bit cells (scatter_lds_minmax_kernel)
inline bool small_cells(int reduce, int elem_bytes, int64_t E) {
    return (reduce == GNNOPS_MIN || reduce == GNNOPS_MAX) && elem_bytes == 2 && E < 65535;
}

inline LdsGeom lds_geometry(int64_t N, int64_t K, int reduce, bool small_cell = false, int64_t B = 1) {
    const size_t per = (reduce == GNNOPS_SUM || reduce == GNNOPS_MUL || small_cell) ? 4 : 8;
    LdsGeom g{0, 0, 0};
    if (N <= 0) return g;
    if ((size_t)N * per <= LDS_BUDGET) {
        int64_t tc = (int64_t)(LDS_BUDGET / ((size_t)N * per));
        if (tc > K) tc = K;
        if (tc > 64) tc = 64;
        if (tc >= 4) tc &= ~(int64_t)3;      // whole 8-B / 32-B pieces of a row
        // few destinations: the strips would be wide and few ((1000)^2: 28 workgroups on 256 CUs) — narrow them until there
        // is about a workgroup per CU (GNNOPS_LDS_NARROW=0 keeps the wide strips, for A/B runs)
        static const bool narrow = !(getenv("GNNOPS_LDS_NARROW") && getenv("GNNOPS_LDS_NARROW")[0] == '0');
        while (narrow && tc >= 8 && B * gnnops_cdiv(K, tc) < 192) tc = (tc >> 1) & ~(int64_t)3;
        if (tc >= 2 || K < 2) return LdsGeom{(int)tc, N, 1};
    }
    const int64_t tc = K < 4 ? K : 4;
    const int64_t rows = (int64_t)(LDS_BUDGET / (per * (size_t)tc));
    const int64_t nchunks = gnnops_cdiv(N, rows);
    if (nchunks > LDS_MAX_CHUNKS) return g;  // too many re-scans: memory-side atomics instead
    return LdsGeom{(int)tc, rows, (int)nchunks};
}

template <typename T, int R, typename I>
int launch_lds(const T* src, const I* index, T* out, int64_t* arg_out, int64_t B, int64_t E, int64_t K, int64_t N,
               LdsGeom g, int init_from_out, hipStream_t stream, int arg32 = 0) {
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_lds_kernel<T, R, I>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET) != hipSuccess)
            return gnnops_check_launch("scatter_lds attribute");
        configured = true;
    }
    const size_t per = (R == GNNOPS_SUM || R == GNNOPS_MUL) ? 4 : 8;
    const int strips = (int)gnnops_cdiv(K, g.tc);
    const size_t lds = (size_t)g.rows * g.tc * per;
    int tshift = 0;
    while ((1 << tshift) < g.tc) ++tshift;
    // a strip that leaves room for several workgroups per CU gets smaller ones: more of them resident, their
    // init / stream / write-back phases overlap (the kernel needs ~70 VGPRs: one 1024-thread workgroup per CU otherwise)
    const int threads = lds > 80 * 1024 ? LDS_THREADS : lds > 40 * 1024 ? 512 : 256;
    if constexpr (R == GNNOPS_MIN || R == GNNOPS_MAX) {
        auto go = [&](auto cell_tag) -> int {
            using CellT = decltype(cell_tag);
            static bool configured_mm = false;
            if (!configured_mm) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_lds_minmax_kernel<T, R, CellT, I>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET) != hipSuccess)
                    return gnnops_check_launch("scatter_lds attribute");
                configured_mm = true;
            }
            const size_t lds_mm = (size_t)g.rows * g.tc * sizeof(CellT);
            const int th = lds_mm > 80 * 1024 ? LDS_THREADS : lds_mm > 40 * 1024 ? 512 : 256;
            hipLaunchKernelGGL((scatter_lds_minmax_kernel<T, R, CellT, I>), dim3((unsigned)(B * strips * g.nchunks)), dim3(th),
                               lds_mm, stream, src, index, out, arg_out, B, E, K, N, g.tc, strips, g.rows, g.nchunks,
                               init_from_out, tshift, arg32);
            return gnnops_check_launch("scatter_lds");
        };
        if (small_cells(R, (int)sizeof(T), E)) return go(uint32_t{});
        return go((unsigned long long)0);
    }
    hipLaunchKernelGGL((scatter_lds_kernel<T, R, I>), dim3((unsigned)(B * strips * g.nchunks)), dim3(threads), lds, stream,
                       src, index, out, arg_out, B, E, K, N, g.tc, strips, g.rows, g.nchunks, init_from_out, tshift);
    return gnnops_check_launch("scatter_lds");
}

template <typename T, typename I>
int dispatch_lds(int reduce, const T* src, const I* index, T* out, int64_t* arg_out, int64_t B, int64_t E,
                 int64_t K, int64_t N, LdsGeom g, int init_from_out, hipStream_t stream, int arg32 = 0) {
    switch (reduce) {
        case GNNOPS_SUM: return launch_lds<T, GNNOPS_SUM, I>(src, index, out, arg_out, B, E, K, N, g, init_from_out, stream);
        case GNNOPS_MEAN: return launch_lds<T, GNNOPS_MEAN, I>(src, index, out, arg_out, B, E, K, N, g, init_from_out, stream);
        case GNNOPS_MUL: return launch_lds<T, GNNOPS_MUL, I>(src, index, out, arg_out, B, E, K, N, g, init_from_out, stream);
        case GNNOPS_MIN: return launch_lds<T, GNNOPS_MIN, I>(src, index, out, arg_out, B, E, K, N, g, init_from_out, stream, arg32);
        case GNNOPS_MAX: return launch_lds<T, GNNOPS_MAX, I>(src, index, out, arg_out, B, E, K, N, g, init_from_out, stream, arg32);
    }
    return GNNOPS_EINVAL;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int grid_for(int64_t n) { return gnnops_grid_cap(gnnops_cdiv(n, 256), 256 * 16); }

template <typename T>
int run(const void* src_, const void* index_, int index_bytes, void* out_, int64_t* arg_out, int64_t B, int64_t E, int64_t K,
        int64_t N, int reduce, int init_from_out, void* workspace, hipStream_t stream, int arg32 = 0) {
    const int64_t* index = (const int64_t*)index_;   // the memory-side-atomic fallback below takes the int64 index only
    const T* src = (const T*)src_;
    T* out = (T*)out_;
    const int64_t nout = B * N * K, nsrc = B * E * K;
    const int gs = grid_for(nsrc), go = grid_for(nout);
    constexpr bool IS_F32 = sizeof(T) == 4;
    char* w = (char*)workspace;

    if (const LdsGeom g = lds_geometry(N, K, reduce, small_cells(reduce, (int)sizeof(T), E), B);
        g.tc > 0 && B * gnnops_cdiv(K, g.tc) * g.nchunks < ((int64_t)1 << 31) && E < ((int64_t)1 << 31)) {
        if (index_bytes == 4)
            return dispatch_lds<T, int32_t>(reduce, src, (const int32_t*)index_, out, arg_out, B, E, K, N, g, init_from_out, stream, arg32);
        if (index_bytes == 2)
            return dispatch_lds<T, uint16_t>(reduce, src, (const uint16_t*)index_, out, arg_out, B, E, K, N, g, init_from_out, stream, arg32);
        return dispatch_lds<T, int64_t>(reduce, src, index, out, arg_out, B, E, K, N, g, init_from_out, stream, arg32);
    }
    if (index_bytes != 8 || arg32) {
        gnnops_set_error("scatter_elementwise: a narrowed index / int32 arg is taken by the LDS-strip form only (B=%lld N=%lld K=%lld)",
                         (long long)B, (long long)N, (long long)K);
        return GNNOPS_EUNSUPPORTED;
    }

    if (reduce == GNNOPS_SUM || reduce == GNNOPS_MEAN || reduce == GNNOPS_MUL) {
        // fp32 accumulator: `out` itself for fp32, a scratch for 16-bit types (rounded once at the end)
        float* acc;
        if (IS_F32) {
            acc = (float*)out;
        } else {
            acc = (float*)w;
            w += align_up((size_t)nout * 4, 256);
        }
        if (init_from_out) {
            if (!IS_F32) hipLaunchKernelGGL((widen_kernel<T>), dim3(go), dim3(256), 0, stream, out, acc, nout);
        } else {
            hipLaunchKernelGGL((fill_kernel<float>), dim3(go), dim3(256), 0, stream, acc, nout,
                               reduce == GNNOPS_MUL ? 1.f : 0.f);
        }
        float* cnt = nullptr;
        if (reduce == GNNOPS_MEAN) {
            cnt = (float*)w;
            hipLaunchKernelGGL((fill_kernel<float>), dim3(go), dim3(256), 0, stream, cnt, nout, 0.f);
            if (nsrc > 0) hipLaunchKernelGGL(count_kernel, dim3(gs), dim3(256), 0, stream, index, cnt, B, E, K, N);
        }
        if (nsrc > 0) {
            if (reduce == GNNOPS_MUL)
                hipLaunchKernelGGL((scatter_mul_f32acc_kernel<T>), dim3(gs), dim3(256), 0, stream, src, index, acc, B, E, K, N);
            else
                hipLaunchKernelGGL((scatter_add_f32acc_kernel<T>), dim3(gs), dim3(256), 0, stream, src, index, acc, B, E, K, N);
        }
        if (!IS_F32 || cnt)
            hipLaunchKernelGGL((finish_kernel<T>), dim3(go), dim3(256), 0, stream, out, acc, cnt, nout);
        return gnnops_check_launch("scatter_elementwise sum/mean/mul");
    }

    // min / max: exact on the stored type, CAS on the containing word
    if (!init_from_out) {
        const float ident = reduce == GNNOPS_MIN ? __builtin_huge_valf() : -__builtin_huge_valf();
        hipLaunchKernelGGL((fill_kernel<T>), dim3(go), dim3(256), 0, stream, out, nout, ident);
    }
    if (nsrc > 0) {
        if (reduce == GNNOPS_MIN)
            hipLaunchKernelGGL((scatter_cas_kernel<T, GNNOPS_MIN>), dim3(gs), dim3(256), 0, stream, src, index, out, B, E, K, N);
        else
            hipLaunchKernelGGL((scatter_cas_kernel<T, GNNOPS_MAX>), dim3(gs), dim3(256), 0, stream, src, index, out, B, E, K, N);
    }
    if (arg_out) {
        hipLaunchKernelGGL(fill_i64_kernel, dim3(go), dim3(256), 0, stream, arg_out, nout, E);
        if (nsrc > 0)
            hipLaunchKernelGGL((scatter_arg_kernel<T>), dim3(gs), dim3(256), 0, stream, src, index, out, arg_out, B, E, K, N,
                               reduce == GNNOPS_MIN ? __builtin_huge_valf() : -__builtin_huge_valf());
        if (!init_from_out)
            hipLaunchKernelGGL((zero_empty_kernel<T>), dim3(go), dim3(256), 0, stream, out, arg_out, nout, E);
    }
    return gnnops_check_launch("scatter_elementwise min/max");
}

}  // namespace

extern "C" size_t gnnops_scatter_elementwise_workspace_bytes(int64_t B, int64_t N, int64_t K, int dtype, int reduce) {
    if (B < 0 || N < 0 || K < 0) return 0;
    const size_t nout = (size_t)(B * N * K);
    size_t b = 0;
    if ((reduce == GNNOPS_SUM || reduce == GNNOPS_MEAN || reduce == GNNOPS_MUL) && dtype != GNNOPS_F32)
        b += align_up(nout * 4, 256);
    if (reduce == GNNOPS_MEAN) b += align_up(nout * 4, 256);
    return b;
}

extern "C" int gnnops_scatter_elementwise(const void* src, const int64_t* index, void* out, int64_t* arg_out,
                                          int64_t B, int64_t E, int64_t K, int64_t N, int dtype, int reduce,
                                          int init_from_out, void* workspace, size_t workspace_bytes,
                                          gnnops_stream_t s) {
    return gnnops_scatter_elementwise_ix(src, index, 8, out, arg_out, B, E, K, N, dtype, reduce, init_from_out, workspace,
                                         workspace_bytes, s);
}

// The same with the index stored in `index_bytes` bytes per element: 8 (int64, what the reference builds), 4 (int32) or 2
// (uint16, N <= 65536) — a narrowed copy made once by gnnops_narrow_index and reused while the index tensor lives (SURVEY.md
// 8(f) rank 2: at the reference's layout-F shapes the int64 index is 8 of every 10 bytes the op reads).
extern "C" int gnnops_scatter_elementwise_ix(const void* src, const void* index, int index_bytes, void* out, int64_t* arg_out,
                                             int64_t B, int64_t E, int64_t K, int64_t N, int dtype, int reduce,
                                             int init_from_out, void* workspace, size_t workspace_bytes,
                                             gnnops_stream_t s) {
    return gnnops_scatter_elementwise_ixa(src, index, index_bytes, out, arg_out, 8, B, E, K, N, dtype, reduce, init_from_out, workspace,
                                          workspace_bytes, s);
}

// The same with the arg rows stored in `arg_bytes` bytes per element: 8 (int64, what torch_scatter returns) or 4 (int32; min / max,
// E < 2^31, the LDS-strip form only — GNNOPS_EUNSUPPORTED otherwise) for a caller that widens them itself on a later pass.
extern "C" int gnnops_scatter_elementwise_ixa(const void* src, const void* index, int index_bytes, void* out, void* arg_out_,
                                              int arg_bytes, int64_t B, int64_t E, int64_t K, int64_t N, int dtype, int reduce,
                                              int init_from_out, void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    int64_t* arg_out = (int64_t*)arg_out_;
    const int arg32 = arg_bytes == 4 ? 1 : 0;
    GNNOPS_REQUIRE(arg_bytes == 8 || (arg_bytes == 4 && (reduce == GNNOPS_MIN || reduce == GNNOPS_MAX) && E < ((int64_t)1 << 31)),
                   GNNOPS_EINVAL, "scatter_elementwise: arg_bytes %d", arg_bytes);
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && E >= 0 && K >= 0 && N >= 0, GNNOPS_EINVAL, "scatter_elementwise: negative size");
    GNNOPS_REQUIRE(reduce >= GNNOPS_SUM && reduce <= GNNOPS_MUL, GNNOPS_EINVAL, "scatter_elementwise: reduce %d", reduce);
    GNNOPS_REQUIRE(index_bytes == 8 || index_bytes == 4 || (index_bytes == 2 && N <= 65536), GNNOPS_EINVAL,
                   "scatter_elementwise: index_bytes %d (N=%lld)", index_bytes, (long long)N);
    GNNOPS_REQUIRE(!(reduce == GNNOPS_MEAN && init_from_out), GNNOPS_EINVAL,
                   "scatter_elementwise: mean cannot start from out");
    GNNOPS_REQUIRE((reduce != GNNOPS_MIN && reduce != GNNOPS_MAX) || arg_out != nullptr || init_from_out,
                   GNNOPS_EINVAL, "scatter_elementwise: min/max without out needs arg_out (empty groups -> 0)");
    if (B * N * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(out && (B * E * K == 0 || (src && index)), GNNOPS_EINVAL, "scatter_elementwise: null pointer");
    const size_t need = gnnops_scatter_elementwise_workspace_bytes(B, N, K, dtype, reduce);
    GNNOPS_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), GNNOPS_EWORKSPACE,
                   "scatter_elementwise: workspace %zu < %zu", workspace_bytes, need);
    switch (dtype) {
        case GNNOPS_F32: return run<float>(src, index, index_bytes, out, arg_out, B, E, K, N, reduce, init_from_out, workspace, stream, arg32);
        case GNNOPS_F16: return run<__half>(src, index, index_bytes, out, arg_out, B, E, K, N, reduce, init_from_out, workspace, stream, arg32);
        case GNNOPS_BF16: return run<__hip_bfloat16>(src, index, index_bytes, out, arg_out, B, E, K, N, reduce, init_from_out, workspace, stream, arg32);
    }
    gnnops_set_error("scatter_elementwise: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}

namespace {
// int64 -> int32 / uint16, streaming: 16-B loads (two entries), 8-B / 4-B stores; eight loads in flight per lane
template <typename O>
__global__ __launch_bounds__(256) void narrow_index_kernel(const int64_t* __restrict__ in, O* __restrict__ out, int64_t n,
                                                           int64_t bound) {
    typedef long long ll2_t __attribute__((ext_vector_type(2)));
    // an id outside [0, bound) becomes the all-ones pattern (-1 / 0xFFFF, never a valid id: bound <= 65535 for two bytes), so
    // the element kernels drop it exactly as they drop it when they read the int64 index — not a wrapped-around valid id
    auto nar = [bound](long long v) -> O { return (v >= 0 && v < bound) ? (O)v : (O)~(O)0; };
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % (2 * sizeof(O)) == 0);
    const int64_t n2 = vec ? n / 2 : 0;
    const ll2_t* p = reinterpret_cast<const ll2_t*>(in);
    int64_t i = gtid;
    for (; i + 7 * stride < n2; i += 8 * stride) {
        ll2_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            O* q = out + 2 * (i + u * stride);
            q[0] = nar(v[u].x);
            q[1] = nar(v[u].y);
        }
    }
    for (; i < n2; i += stride) {
        const ll2_t a = __builtin_nontemporal_load(p + i);
        out[2 * i] = nar(a.x);
        out[2 * i + 1] = nar(a.y);
    }
    for (int64_t j = 2 * n2 + gtid; j < n; j += stride) out[j] = nar(in[j]);
}
}  // namespace

// out[i] = (int32 / uint16) index[i] for ids in [0, bound), all ones otherwise: the narrowed copy
// gnnops_scatter_elementwise_ix reads. bound <= 65535 for two bytes, < 2^31 for four.
extern "C" int gnnops_narrow_index(const int64_t* index, void* out, int64_t n, int out_bytes, int64_t bound, gnnops_stream_t s) {
    GNNOPS_REQUIRE(n >= 0 && bound >= 0 && ((out_bytes == 4 && bound < ((int64_t)1 << 31)) || (out_bytes == 2 && bound <= 65535)),
                   GNNOPS_EINVAL, "narrow_index: bad argument (out_bytes=%d bound=%lld)", out_bytes, (long long)bound);
    if (n == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(index && out, GNNOPS_EINVAL, "narrow_index: null pointer");
    const int grid = gnnops_grid_cap(gnnops_cdiv(n, 256 * 16), 256 * 8);
    if (out_bytes == 4)
        hipLaunchKernelGGL(narrow_index_kernel<int32_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, index, (int32_t*)out, n, bound);
    else
        hipLaunchKernelGGL(narrow_index_kernel<uint16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, index, (uint16_t*)out, n, bound);
    return gnnops_check_launch("narrow_index");
}
