// Shared device/host helpers for the gfx950 kernels. CDNA4 only: 64-lane waves are hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gnnops.h"

#define GNNOPS_WAVE 64

// ---- error plumbing (thread-local message, C ABI never throws) ----
void gnnops_set_error(const char* fmt, ...);
int gnnops_check_launch(const char* what);

#define GNNOPS_REQUIRE(cond, code, ...)            \
    do {                                           \
        if (!(cond)) {                             \
            gnnops_set_error(__VA_ARGS__);         \
            return (code);                         \
        }                                          \
    } while (0)

static inline int64_t gnnops_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Memory-bound launches: enough workgroups to fill 256 CUs several times over, grid-stride the rest
// (cdna_hip_programming.md Guideline 11).
static inline int gnnops_grid_cap(int64_t want, int64_t cap = 256 * 16) {
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

// ---- device-side clears ----
// Counters, flags and row pointers are cleared by a KERNEL, never by hipMemsetAsync: inside a captured graph the runtime
// turns hipMemsetAsync into a memset NODE, and the node that follows did not reliably see it (round 3: the split-K flags of
// gemm.hip read as still set on 3 of 8 replays, tools/diag_sk_handoff.py; eager launches — where the runtime's memset is
// a fill kernel anyway — never failed). `bytes` and the pointer must be multiples of 4; every byte gets `byte_value`.
namespace {
__global__ void gnnops_fill_words_kernel(uint32_t* __restrict__ p, uint32_t word, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = word;
}
}  // namespace
static inline hipError_t gnnops_memset_async(void* p, int byte_value, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if (((uintptr_t)p | bytes) & 3) return hipErrorInvalidValue;
    const uint32_t b = (uint32_t)(byte_value & 0xff), word = b | (b << 8) | (b << 16) | (b << 24);
    const int64_t n = (int64_t)(bytes / 4);
    const int64_t want = (n + 1023) / 1024;
    hipLaunchKernelGGL(gnnops_fill_words_kernel, dim3((unsigned)(want < 1 ? 1 : want > 4096 ? 4096 : want)), dim3(256), 0, stream,
                       (uint32_t*)p, word, n);
    return hipGetLastError();
}

// ---- 16-byte vector type used for every wide global access ----
struct __attribute__((aligned(16))) u32x4 { uint32_t x, y, z, w; };

// ---- 16-byte global accesses, optionally nontemporal (streamed-once data) ----
template <bool NT>
__device__ inline u32x4 load16(const void* p) {
    if constexpr (NT) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
        u32x4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w; return r;
    } else {
        return *reinterpret_cast<const u32x4*>(p);
    }
}
template <bool NT>
__device__ inline void store16(void* p, const u32x4& v) {
    if constexpr (NT) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        v4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, reinterpret_cast<v4*>(p));
    } else {
        *reinterpret_cast<u32x4*>(p) = v;
    }
}

// ---- element traits: fp32 compute for every storage type ----
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;  // elements per 16 B
    __device__ static inline float load(const float* p) { return *p; }
    __device__ static inline void store(float* p, float v) { *p = v; }
    __device__ static inline void unpack(const u32x4& r, float* f) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y);
        f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    }
    __device__ static inline u32x4 pack(const float* f) {
        u32x4 r; r.x = __float_as_uint(f[0]); r.y = __float_as_uint(f[1]);
        r.z = __float_as_uint(f[2]); r.w = __float_as_uint(f[3]); return r;
    }
};
template <> struct Elem<__half> {
    static constexpr int VEC = 8;
    __device__ static inline float load(const __half* p) { return __half2float(*p); }
    __device__ static inline void store(__half* p, float v) { *p = __float2half(v); }
    __device__ static inline void unpack(const u32x4& r, float* f) {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __half2 h = *reinterpret_cast<const __half2*>(&w[i]);
            float2 t = __half22float2(h);
            f[2 * i] = t.x; f[2 * i + 1] = t.y;
        }
    }
    __device__ static inline u32x4 pack(const float* f) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __half2 h = __floats2half2_rn(f[2 * i], f[2 * i + 1]);
            w[i] = *reinterpret_cast<uint32_t*>(&h);
        }
        u32x4 r; r.x = w[0]; r.y = w[1]; r.z = w[2]; r.w = w[3]; return r;
    }
};
template <> struct Elem<__hip_bfloat16> {
    static constexpr int VEC = 8;
    __device__ static inline float load(const __hip_bfloat16* p) { return __bfloat162float(*p); }
    __device__ static inline void store(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
    __device__ static inline void unpack(const u32x4& r, float* f) {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // bf16 -> f32 is a 16-bit shift
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static inline u32x4 pack(const float* f) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __hip_bfloat16 lo = __float2bfloat16(f[2 * i]);
            __hip_bfloat16 hi = __float2bfloat16(f[2 * i + 1]);
            uint16_t l = *reinterpret_cast<uint16_t*>(&lo);
            uint16_t h = *reinterpret_cast<uint16_t*>(&hi);
            w[i] = (uint32_t)l | ((uint32_t)h << 16);
        }
        u32x4 r; r.x = w[0]; r.y = w[1]; r.z = w[2]; r.w = w[3]; return r;
    }
};

// ---- reduce functors (compute in fp32) ----
template <int R> struct Red;
template <> struct Red<GNNOPS_SUM> {
    __device__ static inline float identity() { return 0.f; }
    __device__ static inline float apply(float a, float v) { return a + v; }
};
template <> struct Red<GNNOPS_MEAN> : Red<GNNOPS_SUM> {};
template <> struct Red<GNNOPS_MUL> {
    __device__ static inline float identity() { return 1.f; }
    __device__ static inline float apply(float a, float v) { return a * v; }
};
template <> struct Red<GNNOPS_MIN> {
    __device__ static inline float identity() { return __builtin_huge_valf(); }
    __device__ static inline bool better(float v, float a) { return v < a; }
};
template <> struct Red<GNNOPS_MAX> {
    __device__ static inline float identity() { return -__builtin_huge_valf(); }
    __device__ static inline bool better(float v, float a) { return v > a; }
};

__device__ static inline int lane_id() { return threadIdx.x & 63; }

// ---- launch order -> work item, XCD-aware ----
// Consecutive block ids go round-robin to the 8 XCDs (each with its own L2; MI355X_MICROARCH.md "Workgroup dispatch"), so
// XCD x is given a CONTIGUOUS run of items: neighbouring items — which touch neighbouring bytes of the same lines (column
// strips of one row, the digit runs two adjacent radix tiles write, their slots of one tile-histogram line) — then meet in
// ONE L2, where partial lines merge before they reach memory, instead of in eight. Speed only: any placement is correct.
__device__ inline int64_t xcd_contiguous(int64_t bid, int64_t total) {
    const int64_t q = total / 8, r = total % 8, x = bid % 8;
    return x * q + (x < r ? x : r) + bid / 8;
}

// ---- peers of this lane: the lanes of `valid` whose 8-bit digit equals mine (the match step of every stable ranking) ----
// Eight ballots; per bit one sign-extension, one compare and ONE three-input boolean per 32-lane half
// (v_bitop3_b32, table 0x90 = a & ~(b ^ c)): 4 vector instructions per bit. The asm barrier keeps the compiler from
// re-deriving the ballot operand from `d` (it would add a shift per bit).
__device__ inline uint64_t match_digit8(uint32_t d, uint64_t valid) {
    uint32_t m_lo = (uint32_t)valid, m_hi = (uint32_t)(valid >> 32);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        uint32_t x = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);  // all ones if bit b of the digit is set
        asm("" : "+v"(x));
        const uint64_t bal = __ballot(x != 0u);
        m_lo = __builtin_amdgcn_bitop3_b32(m_lo, (uint32_t)bal, x, 0x90);
        m_hi = __builtin_amdgcn_bitop3_b32(m_hi, (uint32_t)(bal >> 32), x, 0x90);
    }
    return ((uint64_t)m_hi << 32) | m_lo;
}

// ---- block-level exclusive scan of one u32 per thread (NW waves); s_tmp: NW words of LDS ----
__device__ inline uint32_t wave_incl_scan_u32(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}
template <int NW>
__device__ inline uint32_t block_excl_scan_u32(uint32_t v, uint32_t* s_tmp, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan_u32(v);
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint32_t t = s_tmp[w];
        if (w < wave) off += t;
        tot += t;
    }
    __syncthreads();
    if (total) *total = tot;
    return off + incl - v;
}
