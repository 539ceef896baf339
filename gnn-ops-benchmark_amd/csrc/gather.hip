// gather.hip — row/element gathers: torch.index_select, torch.gather, and the fused
// index_select(...).sum(). Reference call sites: op_bm_scripts/benchmark_native_index_select.py:12-15,
// benchmark_native_gather.py:14-17, benchmark_fused_index_select_reduce.py:12-20.
//
// All HBM-bound byte movers: 16-B lane accesses along the feature dimension, several rows in flight
// per lane group to cover the dependent index -> row latency. No MFMA here by design.
#include "common.h"
#include <stdlib.h>
#include "hub.h"
#include <type_traits>

namespace {

constexpr int ROWS_IN_FLIGHT = 4;

// Pull form. Row = rowbytes bytes (multiple of 16), G = 2^gshift lanes per row, chunks of G*16 bytes.
// item = ((b*chunks + c) * E + e). RIF rows are in flight per lane group.
template <bool NT_LD, bool NT_ST, int RIF>
__global__ __launch_bounds__(256) void select_rows_kernel(const char* __restrict__ in, const int64_t* __restrict__ index,
                                                          char* __restrict__ out, int64_t B, int64_t N, int64_t E,
                                                          int64_t rowbytes, int gshift, int chunks, int blk_map) {
    const int G = 1 << gshift;
    const int gl = threadIdx.x & (G - 1);
    const int gi = threadIdx.x >> gshift, groups = 256 >> gshift;
    const int64_t items = B * (int64_t)chunks * E;
    // one matrix, one chunk, every lane of a group on a piece of the row: item = position in `index`
    const bool whole = B == 1 && chunks == 1 && (int64_t)G * 16 == rowbytes;
    // blk_map: a workgroup step covers groups * RIF CONSECUTIVE items — its RIF stores per lane are one contiguous run of
    // output rows, issued back to back; otherwise the items are grid-strided (a lane's RIF rows a whole grid apart).
    // Which one a launch takes: gnnops_index_select below (measured, same box: tools/ab_store_map.py).
    const int64_t step_items = (int64_t)groups * RIF;
    const int64_t ngroups = ((int64_t)gridDim.x * 256) >> gshift;
    const int64_t ustride = blk_map ? groups : ngroups;
    const int64_t step = blk_map ? (int64_t)gridDim.x * step_items : ngroups * RIF;
    const int64_t lead = blk_map ? gi : 0;                      // item0 - lead = first item of a workgroup step
    for (int64_t item0 = blk_map ? (int64_t)blockIdx.x * step_items + gi : ((int64_t)blockIdx.x * 256 + threadIdx.x) >> gshift;
         item0 - lead < items; item0 += step) {
        if (whole && (blk_map ? item0 - lead + step_items <= items : item0 + (RIF - 1) * ustride < items)) {
            // full step, straight-line: RIF index entries, RIF row pieces, RIF stores — every load of a phase in flight
            // (guarded loads each get a vmcnt(0) from the compiler's wait insertion, and nullable pointers become flat loads)
            int64_t n[RIF];
#pragma unroll
            for (int u = 0; u < RIF; ++u) n[u] = index[item0 + u * ustride];
            u32x4 v[RIF];
#pragma unroll
            for (int u = 0; u < RIF; ++u) v[u] = load16<NT_LD>(in + n[u] * rowbytes + (int64_t)gl * 16);
#pragma unroll
            for (int u = 0; u < RIF; ++u) store16<NT_ST>(out + (item0 + u * ustride) * rowbytes + (int64_t)gl * 16, v[u]);
            continue;
        }
        const char* sp[RIF];
        char* dp[RIF];
#pragma unroll
        for (int u = 0; u < RIF; ++u) {
            const int64_t item = item0 + u * ustride;
            sp[u] = nullptr;
            dp[u] = nullptr;
            if (item < items) {
                const int64_t e = item % E;
                const int64_t bc = item / E;
                const int c = (int)(bc % chunks);
                const int64_t b = bc / chunks;
                const int64_t colb = ((int64_t)c * G + gl) * 16;
                if (colb < rowbytes) {
                    const int64_t n = index[e];
                    sp[u] = in + (b * N + n) * rowbytes + colb;
                    dp[u] = out + (b * E + e) * rowbytes + colb;
                }
            }
        }
        u32x4 v[RIF];
#pragma unroll
        for (int u = 0; u < RIF; ++u)
            if (sp[u]) v[u] = load16<NT_LD>(sp[u]);
#pragma unroll
        for (int u = 0; u < RIF; ++u)
            if (sp[u]) store16<NT_ST>(dp[u], v[u]);
    }
}

// Push form over a plan: item = ((b*chunks + c) * N + n); the input row is loaded once and stored to
// every output row of its segment (PU positions fetched per step so the stores issue back to back).
template <bool NT_LD, bool NT_ST, int PU>
__global__ __launch_bounds__(256) void select_rows_push_kernel(const char* __restrict__ in,
                                                               const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ perm, char* __restrict__ out,
                                                               int64_t B, int64_t N, int64_t E, int64_t rowbytes,
                                                               int gshift, int chunks, hub::Ws hw, int hub_on) {
    const int G = 1 << gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> gshift;
    const int gl = (int)(gtid & (G - 1));
    const int64_t items = B * (int64_t)chunks * N;
    for (int64_t item = gtid >> gshift; item < items; item += ngroups) {
        const int64_t n = item % N;
        const int64_t bc = item / N;
        const int c = (int)(bc % chunks);
        const int64_t b = bc / chunks;
        const int64_t colb = ((int64_t)c * G + gl) * 16;
        if (colb >= rowbytes) continue;
        const int32_t beg = rowptr[n], end = rowptr[n + 1];
        if (beg == end) continue;
        if (hub_on && end - beg > hub::T_HUB) {  // a hot row (hub.h): its outputs are written by the hub pass
            if (gl == 0 && c == 0) hub::append(hw, (int)n, beg, end, end - beg);
            continue;
        }
        const u32x4 v = load16<NT_LD>(in + (b * N + n) * rowbytes + colb);
        char* outb = out + (b * E) * rowbytes + colb;
        for (int32_t j = beg; j < end; j += PU) {
            int32_t e[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) e[u] = (j + u < end) ? perm[j + u] : -1;
#pragma unroll
            for (int u = 0; u < PU; ++u)
                if (e[u] >= 0) store16<NT_ST>(outb + (int64_t)e[u] * rowbytes, v);
        }
    }
}

// Element forms (any K / alignment). index_select: index[e]; gather: index[b,e,k].
template <typename U, bool FULL_INDEX>
__global__ __launch_bounds__(256) void select_elems_kernel(const U* __restrict__ in, const int64_t* __restrict__ index,
                                                           U* __restrict__ out, int64_t B, int64_t N, int64_t K,
                                                           int64_t E) {
    const int64_t total = B * E * K;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = o % K;
        const int64_t be = o / K;
        const int64_t e = be % E;
        const int64_t b = be / E;
        const int64_t n = FULL_INDEX ? index[o] : index[e];
        out[o] = in[(b * N + n) * K + k];
    }
}

// LDS-staged element form: a workgroup owns one (b, column strip of TC columns), parks input[b, :, strip] in
// LDS (N * TC elements, coalesced along the strip) and serves every selected element from there, so the
// reads of `input` are sequential and each output element is one LDS read + one coalesced store. Covers what
// the row form cannot: K == 1 (index_select / gather along the last dim of a matrix, the reference's dim-1
// sweeps) and gather along dim 0 with its per-element index (benchmark_native_gather.py:68-73).
constexpr int GL_THREADS = 1024;
constexpr size_t GL_BUDGET = 160 * 1024 - 512;

// Consecutive block ids go round-robin to the 8 XCDs; give each XCD a contiguous run of work items so that neighbouring
// column strips (which share the 64/128-B lines of every input / index row) stream through the same L2.
__device__ inline int64_t gl_xcd_contiguous(int64_t bid, int64_t total) {
    const int64_t q = total / 8, r = total % 8, x = bid % 8;
    return x * q + (x < r ? x : r) + bid / 8;
}

template <typename U, bool FULL_INDEX>
__global__ __launch_bounds__(GL_THREADS) void gather_lds_kernel(const U* __restrict__ in, const int64_t* __restrict__ index,
                                                                U* __restrict__ out, int64_t B, int64_t N, int64_t K,
                                                                int64_t E, int TC, int strips, int tshift) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gl_raw[];
    U* tile = reinterpret_cast<U*>(gl_raw);
    constexpr int UNR = 8;  // loads in flight per thread (unconditional, on clamped rows: a load under a branch is
                            // waited for inside the branch)
    const int64_t item = gl_xcd_contiguous(blockIdx.x, gridDim.x);
    const int64_t b = item / strips;
    const int64_t k0 = (int64_t)(item % strips) * TC;
    const int tc = (int)((K - k0 < TC) ? (K - k0) : TC);
    const int threads = (int)blockDim.x;
    // thread -> (row slot er, column kk of the strip), kk = tid mod 2^tshift >= TC: no divisions
    const int kk = threadIdx.x & ((1 << tshift) - 1);
    const int er = threadIdx.x >> tshift;
    const int rpi = threads >> tshift;
    const bool col_ok = kk < tc;

    const U* ib = in + (b * N) * K + k0;
    if (K == 1 && (((uintptr_t)ib) & 15) == 0) {  // the strip is the whole contiguous row: 16-B loads
        constexpr int PER = 16 / (int)sizeof(U);
        const int64_t nvec = N / PER;
        const u32x4* sv = reinterpret_cast<const u32x4*>(ib);
        u32x4* dv = reinterpret_cast<u32x4*>(tile);
        int64_t i = threadIdx.x;
        for (; i + 3 * threads < nvec; i += 4 * threads) {
            const u32x4 a = sv[i], c = sv[i + threads], d = sv[i + 2 * threads], f = sv[i + 3 * threads];
            dv[i] = a; dv[i + threads] = c; dv[i + 2 * threads] = d; dv[i + 3 * threads] = f;
        }
        for (; i < nvec; i += threads) dv[i] = sv[i];
        for (int64_t j = nvec * PER + threadIdx.x; j < N; j += threads) tile[j] = ib[j];
    } else if (col_ok) {
        for (int64_t n0 = er; n0 < N; n0 += (int64_t)rpi * UNR) {
            U v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int64_t n = n0 + (int64_t)u * rpi;
                v[u] = ib[(n < N ? n : N - 1) * K + kk];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int64_t n = n0 + (int64_t)u * rpi;
                if (n < N) tile[n * tc + kk] = v[u];
            }
        }
    }
    __syncthreads();
    if (!col_ok) return;
    U* ob = out + (b * E) * K + k0 + kk;
    const int64_t* xb = FULL_INDEX ? index + (b * E) * K + k0 + kk : index;
    const int64_t xstride = FULL_INDEX ? K : 1;
    for (int64_t e0 = er; e0 < E; e0 += (int64_t)rpi * UNR) {
        int64_t n[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t e = e0 + (int64_t)u * rpi;
            n[u] = xb[(e < E ? e : E - 1) * xstride];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t e = e0 + (int64_t)u * rpi;
            if (e < E) ob[e * K] = tile[n[u] * tc + kk];
        }
    }
}

// index_select along the last dim of a matrix (K == 1) with ONE index for every row — the reference's dim-1 sweeps
// (benchmark_native_index_select.py:61-67): a workgroup parks TB whole input rows in LDS and every index value it loads
// serves all TB rows, so the index is read B / TB times instead of B times (it is as large as a row of the output).
constexpr int SK1_THREADS = 512, SK1_MAX_TB = 4;
constexpr size_t SK1_LDS_TARGET = 64 * 1024;

template <typename U>
__global__ __launch_bounds__(SK1_THREADS) void select_k1_kernel(const U* __restrict__ in, const int64_t* __restrict__ index,
                                                                U* __restrict__ out, int64_t B, int64_t N, int64_t E,
                                                                int TB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sk1_raw[];
    U* rows = reinterpret_cast<U*>(sk1_raw);  // [TB][N]
    constexpr int UNR = 8;
    const int64_t b0 = (int64_t)blockIdx.x * TB;
    const int tb = (int)((B - b0 < TB) ? (B - b0) : TB);
    const U* sb = in + b0 * N;
    const int64_t nelem = (int64_t)tb * N;  // the tb rows are contiguous
    if ((((uintptr_t)sb) & 15) == 0) {
        constexpr int PER = 16 / (int)sizeof(U);
        const int64_t nvec = nelem / PER;
        const u32x4* sv = reinterpret_cast<const u32x4*>(sb);
        u32x4* dv = reinterpret_cast<u32x4*>(rows);
        int64_t i = threadIdx.x;
        for (; i + 3 * SK1_THREADS < nvec; i += 4 * SK1_THREADS) {
            const u32x4 a = sv[i], c = sv[i + SK1_THREADS], d = sv[i + 2 * SK1_THREADS], f = sv[i + 3 * SK1_THREADS];
            dv[i] = a; dv[i + SK1_THREADS] = c; dv[i + 2 * SK1_THREADS] = d; dv[i + 3 * SK1_THREADS] = f;
        }
        for (; i < nvec; i += SK1_THREADS) dv[i] = sv[i];
        for (int64_t j = nvec * PER + threadIdx.x; j < nelem; j += SK1_THREADS) rows[j] = sb[j];
    } else {
        for (int64_t i = threadIdx.x; i < nelem; i += SK1_THREADS) rows[i] = sb[i];
    }
    __syncthreads();
    U* ob = out + b0 * E;
    if constexpr (sizeof(U) == 2) {
        // 16-bit elements, even E: a lane produces TWO neighbouring outputs per step — one 16-B load of two indices, one
        // 4-byte store per row — instead of 2-byte stores (sub-dword stores cost as much address work as dword ones)
        if ((E & 1) == 0 && (((uintptr_t)ob) & 3) == 0 && (((uintptr_t)index) & 15) == 0) {
            const int64_t E2 = E >> 1;
            const longlong2* ix2 = reinterpret_cast<const longlong2*>(index);
            for (int64_t p0 = threadIdx.x; p0 < E2; p0 += (int64_t)SK1_THREADS * UNR) {
                longlong2 n2[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int64_t p = p0 + (int64_t)u * SK1_THREADS;
                    n2[u] = ix2[p < E2 ? p : E2 - 1];
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int64_t p = p0 + (int64_t)u * SK1_THREADS;
                    if (p >= E2) continue;
#pragma unroll
                    for (int t = 0; t < SK1_MAX_TB; ++t) {
                        if (t < tb) {
                            const uint32_t lo = rows[(int64_t)t * N + n2[u].x], hi = rows[(int64_t)t * N + n2[u].y];
                            reinterpret_cast<uint32_t*>(ob + (int64_t)t * E)[p] = lo | (hi << 16);
                        }
                    }
                }
            }
            return;
        }
    }
    for (int64_t e0 = threadIdx.x; e0 < E; e0 += (int64_t)SK1_THREADS * UNR) {
        int64_t n[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {  // unconditional, clamped: the eight index loads are in flight together
            const int64_t e = e0 + (int64_t)u * SK1_THREADS;
            n[u] = index[e < E ? e : E - 1];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t e = e0 + (int64_t)u * SK1_THREADS;
            if (e >= E) continue;
#pragma unroll
            for (int t = 0; t < SK1_MAX_TB; ++t)
                if (t < tb) ob[(int64_t)t * E + e] = rows[(int64_t)t * N + n[u]];
        }
    }
}

template <typename U>
int launch_select_k1(const void* in, const int64_t* index, void* out, int64_t B, int64_t N, int64_t E, hipStream_t stream) {
    int tb = (int)(SK1_LDS_TARGET / ((size_t)N * sizeof(U)));
    if (tb < 1) tb = 1;
    if (tb > SK1_MAX_TB) tb = SK1_MAX_TB;
    if (tb > B) tb = (int)B;
    hipLaunchKernelGGL((select_k1_kernel<U>), dim3((unsigned)gnnops_cdiv(B, tb)), dim3(SK1_THREADS), (size_t)tb * N * sizeof(U),
                       stream, (const U*)in, index, (U*)out, B, N, E, tb);
    return gnnops_check_launch("index_select");
}

inline int gather_lds_width(int64_t N, int64_t K, int64_t E, int elem_bytes, int64_t B = 1) {
    if (N <= 0 || (size_t)N * elem_bytes > GL_BUDGET) return 0;
    int64_t tc = (int64_t)(GL_BUDGET / ((size_t)N * elem_bytes));
    if (tc > K) tc = K;
    if (tc > 64) tc = 64;
    if (tc >= 8) tc &= ~(int64_t)7;
    // a short table would fit in few wide strips — (1224)^2 fp16: 20 workgroups on 256 CUs, 45 us — so narrow them (not
    // below 16 B per row) until there is about a workgroup per CU
    while (tc >= 16 && B * gnnops_cdiv(K, tc) < 256) tc = (tc >> 1) & ~(int64_t)7;  // stays a multiple of 8, at least 8
    if (tc * elem_bytes < 8 && K * elem_bytes >= 8) return 0;  // strips under 8 bytes waste most of every line
    // staging reads the whole strip (N * elem bytes per column); gathering from HBM costs a 32-B sector per selected
    // element: stage unless fewer than one element per sector-equivalent of the strip is selected
    if (E * 32 < N * elem_bytes) return 0;
    return (int)tc;
}

template <typename U, bool FULL_INDEX>
int launch_gather_lds(const void* in, const int64_t* index, void* out, int64_t B, int64_t N, int64_t K, int64_t E, int tc,
                      hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_lds_kernel<U, FULL_INDEX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GL_BUDGET) != hipSuccess)
            return gnnops_check_launch("gather_lds attribute");
        configured = true;
    }
    const int strips = (int)gnnops_cdiv(K, tc);
    int tshift = 0;
    while ((1 << tshift) < tc) ++tshift;
    const size_t lds = (size_t)N * tc * sizeof(U);
    // smaller workgroups when several fit a CU: their stage / gather phases overlap
    const int threads = lds > 80 * 1024 ? GL_THREADS : lds > 40 * 1024 ? 512 : 256;
    hipLaunchKernelGGL((gather_lds_kernel<U, FULL_INDEX>), dim3((unsigned)(B * strips)), dim3(threads), lds, stream,
                       (const U*)in, index, (U*)out, B, N, K, E, tc, strips, tshift);
    return gnnops_check_launch("gather_lds");
}

// ---- fused index_select + sum ----
constexpr int FUSED_BLOCKS = 256 * 8;

template <typename T, int RIF>
__global__ __launch_bounds__(256) void select_sum_rows_kernel(const T* __restrict__ in, const int64_t* __restrict__ index,
                                                              float* __restrict__ partial, int64_t B, int64_t N,
                                                              int64_t K, int64_t E, int gshift, int chunks) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float s_part[4];
    const int G = 1 << gshift;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> gshift;
    const int gl = (int)(gtid & (G - 1));
    const int64_t items = B * (int64_t)chunks * E;
    float acc = 0.f;
    // One matrix, one column chunk (BASELINE config 4 and every row of at most 1 KiB): item = position in `index`, no
    // divisions. The RIF index entries of a step are loaded first, then the RIF row addresses are formed (and pinned: the
    // compiler otherwise sinks each into its guarded load and waits for index u between two row loads, which also waits for
    // the rows in flight), then the RIF rows are loaded together. Offsets from the typed base pointer, not nullable
    // pointers — those lose their address space and become flat loads that wait on both counters.
    const bool simple = B == 1 && chunks == 1;
    const bool whole = simple && (int64_t)G * VEC == K;   // every lane of a group has a column: no per-lane guard at all
    for (int64_t item0 = gtid >> gshift; item0 < items; item0 += ngroups * RIF) {
        if (whole && item0 + (int64_t)(RIF - 1) * ngroups < items) {
            // full step, straight-line: RIF index entries, then RIF rows, all in flight, nothing guarded (guarded loads get a
            // vmcnt(0) each from the compiler's wait insertion — the RIF rows of a step then arrive one after the other)
            int64_t idx[RIF];
#pragma unroll
            for (int u = 0; u < RIF; ++u) idx[u] = index[item0 + (int64_t)u * ngroups];
            u32x4 v[RIF];
#pragma unroll
            for (int u = 0; u < RIF; ++u) v[u] = load16<true>(in + idx[u] * K + (int64_t)gl * VEC);
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                float f[VEC];
                Elem<T>::unpack(v[u], f);
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc += f[q];
            }
            continue;
        }
        int64_t off[RIF];
        bool ok[RIF];
#pragma unroll
        for (int u = 0; u < RIF; ++u) {
            const int64_t item = item0 + (int64_t)u * ngroups;
            ok[u] = false;
            off[u] = 0;
            if (item < items) {
                int64_t e = item, b = 0, col = (int64_t)gl * VEC;
                if (!simple) {
                    e = item % E;
                    const int64_t bc = item / E;
                    const int c = (int)(bc % chunks);
                    b = bc / chunks;
                    col = ((int64_t)c * G + gl) * VEC;
                }
                if (col < K) {
                    ok[u] = true;
                    off[u] = (b * N + index[e]) * K + col;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RIF; ++u) asm volatile("" : "+v"(off[u]));
        u32x4 v[RIF];
#pragma unroll
        for (int u = 0; u < RIF; ++u)
            if (ok[u]) v[u] = load16<true>(in + off[u]);
#pragma unroll
        for (int u = 0; u < RIF; ++u) {
            if (ok[u]) {
                float f[VEC];
                Elem<T>::unpack(v[u], f);
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc += f[q];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void select_sum_elems_kernel(const T* __restrict__ in, const int64_t* __restrict__ index,
                                                               float* __restrict__ partial, int64_t B, int64_t N,
                                                               int64_t K, int64_t E) {
    __shared__ float s_part[4];
    const int64_t total = B * E * K;
    float acc = 0.f;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = o % K;
        const int64_t be = o / K;
        const int64_t e = be % E;
        const int64_t b = be / E;
        acc += Elem<T>::load(in + (b * N + index[e]) * K + k);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// K == 1 (index_select along the last dim of a matrix, then sum): a workgroup walks rows b, parks each row in LDS and
// sums the selected elements from there (coalesced row read, coalesced index read, LDS gathers).
template <typename T>
__global__ __launch_bounds__(1024) void select_sum_lds_kernel(const T* __restrict__ in, const int64_t* __restrict__ index,
                                                              float* __restrict__ partial, int64_t B, int64_t N, int64_t E) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_raw[];
    T* row = reinterpret_cast<T*>(ss_raw);
    __shared__ float s_part[16];
    float acc = 0.f;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();  // previous row fully consumed
        for (int64_t n = threadIdx.x; n < N; n += 1024) row[n] = in[b * N + n];
        __syncthreads();
        for (int64_t e = threadIdx.x; e < E; e += 1024) acc += Elem<T>::load(row + index[e]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_part[w];
        partial[blockIdx.x] = t;
    }
}

// Fixed-order combine of the block partials (one block), so the result is run-to-run reproducible.
__global__ __launch_bounds__(256) void combine_partials_kernel(const float* __restrict__ partial, int n,
                                                               float* __restrict__ out) {
    __shared__ float s_part[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

struct RowGeom { int gshift; int chunks; };
inline RowGeom row_geom(int64_t vecs) {
    RowGeom g{0, 1};
    while ((1 << g.gshift) < vecs && g.gshift < 6) ++g.gshift;
    g.chunks = (int)gnnops_cdiv(vecs, (int64_t)1 << g.gshift);
    return g;
}

// Long rows whose byte length is not a multiple of 16 (the reference's (L, L) fp16 sweeps: 2 L bytes, e.g. 28284): source
// and destination rows are then misaligned relative to each other, so the copy unit is the widest type that divides the
// row (4 or 8 bytes, 2 for odd L). One wave per output row: the index is read once per row, the lanes sweep the row with
// eight loads in flight each, no per-element division (the element kernel pays one per element).
template <typename U>
__global__ __launch_bounds__(256) void select_longrows_kernel(const U* __restrict__ in, const int64_t* __restrict__ index,
                                                              U* __restrict__ out, int64_t B, int64_t N, int64_t KU,
                                                              int64_t E) {
    constexpr int UNR = 8, SEG = 64 * UNR;  // a wave copies one 512-unit piece of a row per step (few long rows still fill the chip)
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t segs = (KU + SEG - 1) / SEG;
    for (int64_t item = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; item < B * E * segs; item += nwaves) {
        const int64_t row = item / segs, seg = item - row * segs;
        const int64_t b = row / E, e = row - b * E;
        const U* src = in + (b * N + index[e]) * KU;
        U* dst = out + row * KU;
        const int64_t k0 = seg * SEG + lane;
        U v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t k = k0 + 64 * u;
            v[u] = __builtin_nontemporal_load(src + (k < KU ? k : KU - 1));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t k = k0 + 64 * u;
            if (k < KU) __builtin_nontemporal_store(v[u], dst + k);
        }
    }
}

// fused index_select + sum over such rows: the same sweep, PAIR = 2 sums two 16-bit values per 4-byte load.
template <typename T, int PAIR>
__global__ __launch_bounds__(256) void select_sum_longrows_kernel(const T* __restrict__ in, const int64_t* __restrict__ index,
                                                                  float* __restrict__ partial, int64_t B, int64_t N,
                                                                  int64_t K, int64_t E) {
    constexpr int UNR = 8;
    // raw bits: a 4-byte pair of 16-bit values, or one value as an unsigned integer of its own width
    using Raw = typename std::conditional<sizeof(T) == 4, uint32_t, uint16_t>::type;
    using Unit = typename std::conditional<PAIR == 2, uint32_t, Raw>::type;
    __shared__ float s_part[4];
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t KU = K / PAIR;
    float acc = 0.f;
    for (int64_t item = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; item < B * E; item += nwaves) {
        const int64_t b = item / E, e = item - b * E;
        const Unit* src = reinterpret_cast<const Unit*>(in + (b * N + index[e]) * K);
        for (int64_t k0 = lane; k0 < KU; k0 += 64 * UNR) {
            Unit v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int64_t k = k0 + 64 * u;
                v[u] = __builtin_nontemporal_load(src + (k < KU ? k : KU - 1));
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (k0 + 64 * u < KU) {
                    if constexpr (PAIR == 2) {
                        const uint16_t lo = (uint16_t)(v[u] & 0xffffu), hi = (uint16_t)(v[u] >> 16);
                        acc += Elem<T>::load(reinterpret_cast<const T*>(&lo));
                        acc += Elem<T>::load(reinterpret_cast<const T*>(&hi));
                    } else {
                        acc += Elem<T>::load(reinterpret_cast<const T*>(&v[u]));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

constexpr int64_t LONGROW_MIN_UNITS = 512;  // from here on a wave per row beats a thread per element

template <typename T>
int launch_select_sum(const void* input, const int64_t* index, float* d_sum, int64_t B, int64_t N, int64_t K,
                      int64_t E, float* partial, hipStream_t stream) {
    constexpr int VEC = Elem<T>::VEC;
    int grid;
    if (K % VEC == 0 && (uintptr_t)input % 16 == 0) {
        RowGeom g = row_geom(K / VEC);
        const int64_t items = B * g.chunks * E;
        // 4 rows in flight per lane group: 2 and 4 tie, 8 loses a third (register pressure) — tools/time_selsum.py
        grid = gnnops_grid_cap(gnnops_cdiv(items, (256 >> g.gshift) * ROWS_IN_FLIGHT), FUSED_BLOCKS);
        hipLaunchKernelGGL((select_sum_rows_kernel<T, ROWS_IN_FLIGHT>), dim3(grid), dim3(256), 0, stream, (const T*)input,
                           index, partial, B, N, K, E, g.gshift, g.chunks);
    } else if (K == 1 && (size_t)N * sizeof(T) <= GL_BUDGET && E * 32 >= N * (int64_t)sizeof(T)) {
        static bool configured = false;
        if (!configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&select_sum_lds_kernel<T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)GL_BUDGET) != hipSuccess)
                return gnnops_check_launch("select_sum_lds attribute");
            configured = true;
        }
        grid = gnnops_grid_cap(B, FUSED_BLOCKS);
        hipLaunchKernelGGL((select_sum_lds_kernel<T>), dim3(grid), dim3(1024), (size_t)N * sizeof(T), stream, (const T*)input,
                           index, partial, B, N, E);
    } else if (K >= LONGROW_MIN_UNITS) {
        grid = gnnops_grid_cap(gnnops_cdiv(B * E, 4), FUSED_BLOCKS);
        if (sizeof(T) == 2 && K % 2 == 0 && (uintptr_t)input % 4 == 0)
            hipLaunchKernelGGL((select_sum_longrows_kernel<T, 2>), dim3(grid), dim3(256), 0, stream, (const T*)input, index,
                               partial, B, N, K, E);
        else
            hipLaunchKernelGGL((select_sum_longrows_kernel<T, 1>), dim3(grid), dim3(256), 0, stream, (const T*)input, index,
                               partial, B, N, K, E);
    } else {
        grid = gnnops_grid_cap(gnnops_cdiv(B * E * K, 256 * 4), FUSED_BLOCKS);
        hipLaunchKernelGGL((select_sum_elems_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)input, index,
                           partial, B, N, K, E);
    }
    hipLaunchKernelGGL(combine_partials_kernel, dim3(1), dim3(256), 0, stream, partial, grid, d_sum);
    return gnnops_check_launch("fused_index_select_sum");
}

}  // namespace

extern "C" int gnnops_index_select(const void* input, const int64_t* index, void* out, int64_t B, int64_t N,
                                   int64_t K, int64_t E, int elem_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && N >= 0 && K >= 0 && E >= 0, GNNOPS_EINVAL, "index_select: negative size");
    GNNOPS_REQUIRE(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8, GNNOPS_EUNSUPPORTED,
                   "index_select: elem_bytes %d", elem_bytes);
    if (B * E * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(input && index && out, GNNOPS_EINVAL, "index_select: null pointer");
    const int64_t rowbytes = K * elem_bytes;
    if (rowbytes % 16 == 0 && (uintptr_t)input % 16 == 0 && (uintptr_t)out % 16 == 0) {
        RowGeom g = row_geom(rowbytes / 16);
        const int64_t items = B * g.chunks * E;
        // nontemporal on both sides, 4 rows in flight per lane group, 64 workgroups per CU of grid:
        // the fastest of the variants swept with tools/time_sel.py at config 2
        const int sgrid = gnnops_grid_cap(gnnops_cdiv(items, (256 >> g.gshift) * ROWS_IN_FLIGHT), 256 * 64);
        const char* mapenv = getenv("GNNOPS_PULL_MAP");   // g: the grid-strided map of rounds 1-2 (A/B: tools/ab_store_map.py; contiguous is 1.1-1.3 % faster)
        const int blk_map = !(mapenv && mapenv[0] == 'g');
        hipLaunchKernelGGL((select_rows_kernel<true, true, ROWS_IN_FLIGHT>), dim3(sgrid), dim3(256), 0, stream,
                           (const char*)input, index, (char*)out, B, N, E, rowbytes, g.gshift, g.chunks, blk_map);
    } else {
        // K == 1 with a batch of rows that fit 64 KiB of LDS: rows parked on chip, the index shared by TB rows
        if (K == 1 && B > 1 && N >= 512 && E >= 256 && (size_t)N * elem_bytes <= 64 * 1024 && E * 32 >= N * elem_bytes &&
            gnnops_cdiv(B, 1) < ((int64_t)1 << 31)) {
            switch (elem_bytes) {
                case 1: return launch_select_k1<uint8_t>(input, index, out, B, N, E, stream);
                case 2: return launch_select_k1<uint16_t>(input, index, out, B, N, E, stream);
                case 4: return launch_select_k1<uint32_t>(input, index, out, B, N, E, stream);
                default: return launch_select_k1<uint64_t>(input, index, out, B, N, E, stream);
            }
        }
        // a row index with K > 1 is already coalesced along k in the element kernel; LDS staging pays for K == 1 rows
        if (const int tc = gather_lds_width(N, K, E, elem_bytes, B);
            tc > 0 && K * elem_bytes <= 8 && B * gnnops_cdiv(K, tc) < ((int64_t)1 << 31)) {
            switch (elem_bytes) {
                case 1: return launch_gather_lds<uint8_t, false>(input, index, out, B, N, K, E, tc, stream);
                case 2: return launch_gather_lds<uint16_t, false>(input, index, out, B, N, K, E, tc, stream);
                case 4: return launch_gather_lds<uint32_t, false>(input, index, out, B, N, K, E, tc, stream);
                default: return launch_gather_lds<uint64_t, false>(input, index, out, B, N, K, E, tc, stream);
            }
        }
        // copy in the widest unit that divides the row and the base alignment (a row is one opaque byte string)
        const uintptr_t al = (uintptr_t)input | (uintptr_t)out | (uintptr_t)rowbytes;
        const int unit = al % 8 == 0 ? 8 : al % 4 == 0 ? 4 : al % 2 == 0 ? 2 : 1;
        if (rowbytes / unit >= LONGROW_MIN_UNITS) {
            const int64_t KU = rowbytes / unit;
            const int lgrid = gnnops_grid_cap(gnnops_cdiv(B * E * gnnops_cdiv(KU, 512), 4), 256 * 32);
            if (unit == 8)
                hipLaunchKernelGGL((select_longrows_kernel<uint64_t>), dim3(lgrid), dim3(256), 0, stream, (const uint64_t*)input,
                                   index, (uint64_t*)out, B, N, KU, E);
            else if (unit == 4)
                hipLaunchKernelGGL((select_longrows_kernel<uint32_t>), dim3(lgrid), dim3(256), 0, stream, (const uint32_t*)input,
                                   index, (uint32_t*)out, B, N, KU, E);
            else if (unit == 2)
                hipLaunchKernelGGL((select_longrows_kernel<uint16_t>), dim3(lgrid), dim3(256), 0, stream, (const uint16_t*)input,
                                   index, (uint16_t*)out, B, N, KU, E);
            else
                hipLaunchKernelGGL((select_longrows_kernel<uint8_t>), dim3(lgrid), dim3(256), 0, stream, (const uint8_t*)input,
                                   index, (uint8_t*)out, B, N, KU, E);
            return gnnops_check_launch("index_select");
        }
        int grid = gnnops_grid_cap(gnnops_cdiv(B * E * K, 256), 256 * 32);
        if (al % 8 == 0)
            hipLaunchKernelGGL((select_elems_kernel<uint64_t, false>), dim3(grid), dim3(256), 0, stream,
                               (const uint64_t*)input, index, (uint64_t*)out, B, N, rowbytes / 8, E);
        else if (al % 4 == 0)
            hipLaunchKernelGGL((select_elems_kernel<uint32_t, false>), dim3(grid), dim3(256), 0, stream,
                               (const uint32_t*)input, index, (uint32_t*)out, B, N, rowbytes / 4, E);
        else if (al % 2 == 0)
            hipLaunchKernelGGL((select_elems_kernel<uint16_t, false>), dim3(grid), dim3(256), 0, stream,
                               (const uint16_t*)input, index, (uint16_t*)out, B, N, rowbytes / 2, E);
        else
            hipLaunchKernelGGL((select_elems_kernel<uint8_t, false>), dim3(grid), dim3(256), 0, stream,
                               (const uint8_t*)input, index, (uint8_t*)out, B, N, rowbytes, E);
    }
    return gnnops_check_launch("index_select");
}

extern "C" int gnnops_index_select_planned(const void* input, const int32_t* rowptr, const int32_t* perm, void* out,
                                           int64_t B, int64_t N, int64_t K, int64_t E, int elem_bytes,
                                           gnnops_stream_t s) {
    return gnnops_index_select_planned_hubs(input, rowptr, perm, out, B, N, K, E, elem_bytes, nullptr, 0, s);
}

// The same with hot rows (selected by more than 8192 outputs) set aside and written by whole workgroups (hub.h);
// hub_workspace: gnnops_hub_workspace_bytes(E, 0, 0) bytes, or NULL.
extern "C" int gnnops_index_select_planned_hubs(const void* input, const int32_t* rowptr, const int32_t* perm, void* out,
                                                int64_t B, int64_t N, int64_t K, int64_t E, int elem_bytes,
                                                void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && N >= 0 && K >= 0 && E >= 0, GNNOPS_EINVAL, "index_select_planned: negative size");
    GNNOPS_REQUIRE(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8, GNNOPS_EUNSUPPORTED,
                   "index_select_planned: elem_bytes %d", elem_bytes);
    if (B * E * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(input && rowptr && perm && out, GNNOPS_EINVAL, "index_select_planned: null pointer");
    const int64_t rowbytes = K * elem_bytes;
    GNNOPS_REQUIRE(rowbytes % 16 == 0 && (uintptr_t)input % 16 == 0 && (uintptr_t)out % 16 == 0, GNNOPS_EUNSUPPORTED,
                   "index_select_planned: rows must be 16-byte multiples and 16-byte aligned");
    RowGeom g = row_geom(rowbytes / 16);
    const int64_t items = B * g.chunks * N;
    const int grid = gnnops_grid_cap(gnnops_cdiv(items, 256 >> g.gshift), 256 * 64);
    hub::Ws hw{};
    int hub_on = 0;
    if (hub_workspace && B == 1 && E > hub::T_HUB) {
        const hub::Layout hl = hub::layout(E, 0, false);
        if (hub_workspace_bytes >= hl.total) {
            hw = hub::make_ws(hub_workspace, hl, E, false);
            if (gnnops_memset_async(hw.counters, 0, 8, stream) != hipSuccess) return gnnops_check_launch("hub memset");
            hub_on = 1;
        }
    }
    hipLaunchKernelGGL((select_rows_push_kernel<true, true, 8>), dim3(grid), dim3(256), 0, stream, (const char*)input,
                       rowptr, perm, (char*)out, B, N, E, rowbytes, g.gshift, g.chunks, hw, hub_on);
    if (hub_on)
        hub::launch_push_pass(false, (const char*)input, perm, nullptr, nullptr, (char*)out, hw, rowbytes, g.gshift, g.chunks,
                              stream);
    return gnnops_check_launch("index_select_planned");
}

extern "C" int gnnops_gather(const void* input, const int64_t* index, void* out, int64_t B, int64_t N, int64_t K,
                             int64_t E, int elem_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && N >= 0 && K >= 0 && E >= 0, GNNOPS_EINVAL, "gather: negative size");
    GNNOPS_REQUIRE(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8, GNNOPS_EUNSUPPORTED,
                   "gather: elem_bytes %d", elem_bytes);
    if (B * E * K == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(input && index && out, GNNOPS_EINVAL, "gather: null pointer");
    const int tc = gather_lds_width(N, K, E, elem_bytes, B);
    const bool lds = tc > 0 && B * gnnops_cdiv(K, tc) < ((int64_t)1 << 31);
    const int grid = gnnops_grid_cap(gnnops_cdiv(B * E * K, 256), 256 * 32);
#define GATHER_CASE(U)                                                                                              \
    if (lds) return launch_gather_lds<U, true>(input, index, out, B, N, K, E, tc, stream);                          \
    hipLaunchKernelGGL((select_elems_kernel<U, true>), dim3(grid), dim3(256), 0, stream, (const U*)input, index,     \
                       (U*)out, B, N, K, E);                                                                         \
    break
    switch (elem_bytes) {
        case 1: GATHER_CASE(uint8_t);
        case 2: GATHER_CASE(uint16_t);
        case 4: GATHER_CASE(uint32_t);
        default: GATHER_CASE(uint64_t);
    }
#undef GATHER_CASE
    return gnnops_check_launch("gather");
}

extern "C" size_t gnnops_fused_select_sum_workspace_bytes(void) { return (size_t)FUSED_BLOCKS * sizeof(float); }

extern "C" int gnnops_fused_index_select_sum(const void* input, const int64_t* index, float* d_sum_f32, int64_t B,
                                             int64_t N, int64_t K, int64_t E, int dtype, void* workspace,
                                             size_t workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && N >= 0 && K >= 0 && E >= 0, GNNOPS_EINVAL, "fused_index_select_sum: negative size");
    GNNOPS_REQUIRE(d_sum_f32 != nullptr, GNNOPS_EINVAL, "fused_index_select_sum: null output");
    if (B * E * K == 0) {
        if (gnnops_memset_async(d_sum_f32, 0, sizeof(float), stream) != hipSuccess)
            return gnnops_check_launch("fused_index_select_sum memset");
        return GNNOPS_OK;
    }
    GNNOPS_REQUIRE(input && index, GNNOPS_EINVAL, "fused_index_select_sum: null pointer");
    GNNOPS_REQUIRE(workspace && workspace_bytes >= gnnops_fused_select_sum_workspace_bytes(), GNNOPS_EWORKSPACE,
                   "fused_index_select_sum: workspace too small");
    float* partial = (float*)workspace;
    switch (dtype) {
        case GNNOPS_F32: return launch_select_sum<float>(input, index, d_sum_f32, B, N, K, E, partial, stream);
        case GNNOPS_F16: return launch_select_sum<__half>(input, index, d_sum_f32, B, N, K, E, partial, stream);
        case GNNOPS_BF16: return launch_select_sum<__hip_bfloat16>(input, index, d_sum_f32, B, N, K, E, partial, stream);
    }
    gnnops_set_error("fused_index_select_sum: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}
