// sort.hip — torch.sort(input, dim, descending, stable) (reference: op_bm_scripts/benchmark_native_sort.py:28-30;
// fp32, shapes 1-D 8e8, (20000,20000), 800^3, dims 0..2, stable in {True, False}, tie-heavy dropout inputs).
//
// Always stable (a stable order is a valid answer for stable=False too). Built on the radix engine of the plan
// builder; every dtype is first mapped to an order-preserving unsigned key (descending: the complement):
//   32-bit keys (f32, f16, bf16, i32)   1-D: 4 passes over u32 keys, values = positions;
//                                       along a dim: one sort of 64-bit keys (segment << 32 | key),
//                                       4 + ceil(log2(segments)/8) passes, fed in memory order so ties keep position
//   64-bit keys (i64, f64)              1-D only: 8 passes over u64 keys
// 16-bit floats skip the passes over key bytes that are identically zero. Values are recovered from the keys, except
// where the key does not determine the bits: -0.0 is keyed like +0.0 (torch compares them equal) and every NaN gets the
// top key (NaNs sort last; first when descending) — those elements are re-read from the input at their source position,
// so `values` is bit for bit `input.gather(dim, indices)` as with torch.sort (signed zeros, NaN signs and payloads kept).
// HBM-bound: per pass hist reads the keys, scatter reads and writes keys + values.
#include "common.h"
#include "sort_engine.h"

namespace {

enum { ST_F32 = 0, ST_F16 = 1, ST_BF16 = 2, ST_I32 = 3, ST_I64 = 4, ST_F64 = 5 };

__device__ inline uint32_t f32_key(float x) {
    uint32_t u = __float_as_uint(x);
    if (u == 0x80000000u) u = 0u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key_f32(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(u);
}
__device__ inline uint64_t f64_key(double x) {
    uint64_t u = (uint64_t)__double_as_longlong(x);
    if (u == 0x8000000000000000ull) u = 0ull;
    if ((u & 0x7fffffffffffffffull) > 0x7ff0000000000000ull) return ~0ull;
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ inline double key_f64(uint64_t k) {
    const uint64_t u = (k & 0x8000000000000000ull) ? (k ^ 0x8000000000000000ull) : ~k;
    return __longlong_as_double((long long)u);
}

template <int ST> struct SortType;
template <> struct SortType<ST_F32> { using T = float; using Key = uint32_t;
    __device__ static Key key(T x) { return f32_key(x); } __device__ static T val(Key k) { return key_f32(k); }
    __device__ static bool special(Key k) { return k == 0x80000000u || k == 0xffffffffu; } };
template <> struct SortType<ST_F16> { using T = __half; using Key = uint32_t;
    __device__ static Key key(T x) { return f32_key(__half2float(x)); } __device__ static T val(Key k) { return __float2half(key_f32(k)); }
    __device__ static bool special(Key k) { return k == 0x80000000u || k == 0xffffffffu; } };
template <> struct SortType<ST_BF16> { using T = __hip_bfloat16; using Key = uint32_t;
    __device__ static Key key(T x) { return f32_key(__bfloat162float(x)); } __device__ static T val(Key k) { return __float2bfloat16(key_f32(k)); }
    __device__ static bool special(Key k) { return k == 0x80000000u || k == 0xffffffffu; } };
template <> struct SortType<ST_I32> { using T = int32_t; using Key = uint32_t;
    __device__ static Key key(T x) { return (uint32_t)x ^ 0x80000000u; } __device__ static T val(Key k) { return (int32_t)(k ^ 0x80000000u); }
    __device__ static bool special(Key) { return false; } };
template <> struct SortType<ST_I64> { using T = int64_t; using Key = uint64_t;
    __device__ static Key key(T x) { return (uint64_t)x ^ 0x8000000000000000ull; } __device__ static T val(Key k) { return (int64_t)(k ^ 0x8000000000000000ull); }
    __device__ static bool special(Key) { return false; } };
template <> struct SortType<ST_F64> { using T = double; using Key = uint64_t;
    __device__ static Key key(T x) { return f64_key(x); } __device__ static T val(Key k) { return key_f64(k); }
    __device__ static bool special(Key k) { return k == 0x8000000000000000ull || k == ~0ull; } };

#define GRID_STRIDE(i, total) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

// flat keys (1-D): key of the element, complemented for descending order
template <int ST>
__global__ void build_flat_keys_kernel(const typename SortType<ST>::T* __restrict__ in,
                                       typename SortType<ST>::Key* __restrict__ keys, int64_t n, int descending) {
    GRID_STRIDE(i, n) {
        typename SortType<ST>::Key k = SortType<ST>::key(in[i]);
        keys[i] = descending ? (typename SortType<ST>::Key)~k : k;
    }
}
// segmented keys: segment (b, k) << 32 | 32-bit key
template <int ST>
__global__ void build_seg_keys_kernel(const typename SortType<ST>::T* __restrict__ in, uint64_t* __restrict__ keys,
                                      int64_t B, int64_t E, int64_t K, int descending) {
    GRID_STRIDE(i, B * E * K) {
        const int64_t k = i % K;
        const int64_t b = i / (E * K);
        uint32_t key = SortType<ST>::key(in[i]);
        if (descending) key = ~key;
        keys[i] = ((uint64_t)(b * K + k) << 32) | (uint64_t)key;
    }
}
template <int ST>
__global__ void finish_flat_kernel(const typename SortType<ST>::Key* __restrict__ keys, const uint32_t* __restrict__ vals,
                                   const typename SortType<ST>::T* __restrict__ in,
                                   typename SortType<ST>::T* __restrict__ values, int64_t* __restrict__ indices, int64_t n,
                                   int descending) {
    GRID_STRIDE(p, n) {
        typename SortType<ST>::Key k = keys[p];
        if (descending) k = (typename SortType<ST>::Key)~k;
        const uint32_t e = vals[p];
        // zeros and NaNs: the key does not fix the bits. Equal keys keep position order, so these re-reads ascend.
        values[p] = SortType<ST>::special(k) ? in[e] : SortType<ST>::val(k);
        indices[p] = (int64_t)e;
    }
}
// sorted position p = seg*E + r  ->  output element [b, r, k]; source position e from the linear index.
template <int ST>
__global__ void finish_seg_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                  const typename SortType<ST>::T* __restrict__ in,
                                  typename SortType<ST>::T* __restrict__ values, int64_t* __restrict__ indices, int64_t B,
                                  int64_t E, int64_t K, int descending) {
    GRID_STRIDE(p, B * E * K) {
        const int64_t seg = p / E, r = p % E;
        const int64_t b = seg / K, k = seg % K;
        const int64_t o = (b * E + r) * K + k;
        uint32_t key = (uint32_t)keys[p];
        if (descending) key = ~key;
        const uint32_t lin = vals[p];   // linear position of the element in `in`
        values[o] = SortType<ST>::special(key) ? in[lin] : SortType<ST>::val(key);
        indices[o] = ((int64_t)lin / K) % E;
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int grid_for(int64_t n) { return gnnops_grid_cap(gnnops_cdiv(n, 256), 256 * 16); }

struct Work {
    void *keys_a, *keys_b;
    uint32_t *vals_a, *vals_b, *tile_hist, *digit_total;
    int tiles;
};

// LSD passes over u32 keys already built in w.keys_b; returns the buffers holding the sorted data.
int sort_u32(const Work& w, int64_t n, int first_shift, hipStream_t stream, uint32_t** keys, uint32_t** vals) {
    uint32_t* kin = (uint32_t*)w.keys_b;
    uint32_t* kout = (uint32_t*)w.keys_a;
    uint32_t* vin = nullptr;
    uint32_t* vout = w.vals_a;
    bool first = true;
    for (int shift = first_shift; shift < 32; shift += 8) {
        const int rc = first ? sortengine::pass_first_u32(kin, kout, vout, n, shift, w.tile_hist, w.digit_total, w.tiles, stream)
                             : sortengine::pass_u32(kin, vin, kout, vout, n, shift, w.tile_hist, w.digit_total, w.tiles, stream);
        if (rc) return rc;
        first = false;
        uint32_t* t = kin; kin = kout; kout = t;
        uint32_t* nv = (vout == w.vals_a) ? w.vals_b : w.vals_a;
        vin = vout; vout = nv;
    }
    *keys = kin; *vals = vin;
    return GNNOPS_OK;
}
// LSD passes over u64 keys built in w.keys_b, bytes [first_shift/8, nbytes).
int sort_u64(const Work& w, int64_t n, int first_shift, int last_shift, int skip_lo, int skip_hi, hipStream_t stream,
             uint64_t** keys, uint32_t** vals) {
    uint64_t* kin = (uint64_t*)w.keys_b;
    uint64_t* kout = (uint64_t*)w.keys_a;
    uint32_t* vin = nullptr;
    uint32_t* vout = w.vals_a;
    bool first = true;
    for (int shift = first_shift; shift < last_shift; shift += 8) {
        if (shift >= skip_lo && shift < skip_hi) continue;
        const int rc = first ? sortengine::pass_first_u64(kin, kout, vout, n, shift, w.tile_hist, w.digit_total, w.tiles, stream)
                             : sortengine::pass_u64(kin, vin, kout, vout, n, shift, w.tile_hist, w.digit_total, w.tiles, stream);
        if (rc) return rc;
        first = false;
        uint64_t* t = kin; kin = kout; kout = t;
        uint32_t* nv = (vout == w.vals_a) ? w.vals_b : w.vals_a;
        vin = vout; vout = nv;
    }
    *keys = kin; *vals = vin;
    return GNNOPS_OK;
}

template <int ST>
int sort_typed(const void* input, void* values, int64_t* indices, int64_t B, int64_t E, int64_t K, int descending,
               const Work& w, hipStream_t stream) {
    using TT = typename SortType<ST>::T;
    using Key = typename SortType<ST>::Key;
    const int64_t n = B * E * K;
    const bool flat = (B * K == 1);
    const int zero_lo = (ST == ST_BF16) ? 16 : (ST == ST_F16) ? 8 : 0;  // low key bits that are identically zero
    if constexpr (sizeof(Key) == 8) {
        if (!flat) {
            gnnops_set_error("sort: 64-bit dtypes are sorted along a dimension of a 1-D tensor only");
            return GNNOPS_EUNSUPPORTED;
        }
        hipLaunchKernelGGL((build_flat_keys_kernel<ST>), dim3(grid_for(n)), dim3(256), 0, stream, (const TT*)input,
                           (uint64_t*)w.keys_b, n, descending);
        uint64_t* sk; uint32_t* sv;
        if (int rc = sort_u64(w, n, 0, 64, 0, 0, stream, &sk, &sv)) return rc;
        hipLaunchKernelGGL((finish_flat_kernel<ST>), dim3(grid_for(n)), dim3(256), 0, stream, sk, sv, (const TT*)input, (TT*)values, indices, n, descending);
        return gnnops_check_launch("sort finish");
    } else {
        if (flat) {
            uint32_t* sk; uint32_t* sv;
            if (ST == ST_F32 && !descending) {
                // pass 0 reads the floats directly (KeyF32 adapter), values implicit
                int rc = sortengine::pass_first_f32((const float*)input, (uint32_t*)w.keys_a, w.vals_a, n, 0, w.tile_hist, w.digit_total, w.tiles, stream);
                if (rc) return rc;
                rc = sortengine::pass_u32((uint32_t*)w.keys_a, w.vals_a, (uint32_t*)w.keys_b, w.vals_b, n, 8, w.tile_hist, w.digit_total, w.tiles, stream);
                if (rc) return rc;
                rc = sortengine::pass_u32((uint32_t*)w.keys_b, w.vals_b, (uint32_t*)w.keys_a, w.vals_a, n, 16, w.tile_hist, w.digit_total, w.tiles, stream);
                if (rc) return rc;
                rc = sortengine::pass_u32((uint32_t*)w.keys_a, w.vals_a, (uint32_t*)w.keys_b, w.vals_b, n, 24, w.tile_hist, w.digit_total, w.tiles, stream);
                if (rc) return rc;
                sk = (uint32_t*)w.keys_b; sv = w.vals_b;
            } else {
                hipLaunchKernelGGL((build_flat_keys_kernel<ST>), dim3(grid_for(n)), dim3(256), 0, stream, (const TT*)input,
                                   (uint32_t*)w.keys_b, n, descending);
                // a complemented key has ones, not zeros, in the dead low bits: still constant, still skippable
                if (int rc = sort_u32(w, n, zero_lo, stream, &sk, &sv)) return rc;
            }
            hipLaunchKernelGGL((finish_flat_kernel<ST>), dim3(grid_for(n)), dim3(256), 0, stream, sk, sv, (const TT*)input, (TT*)values, indices, n, descending);
            return gnnops_check_launch("sort finish");
        }
        const int64_t segs = B * K;
        int segbits = 0;
        while (((int64_t)1 << segbits) < segs) ++segbits;
        hipLaunchKernelGGL((build_seg_keys_kernel<ST>), dim3(grid_for(n)), dim3(256), 0, stream, (const TT*)input,
                           (uint64_t*)w.keys_b, B, E, K, descending);
        uint64_t* sk; uint32_t* sv;
        if (int rc = sort_u64(w, n, zero_lo, 32 + 8 * ((segbits + 7) / 8), 0, 0, stream, &sk, &sv)) return rc;
        hipLaunchKernelGGL((finish_seg_kernel<ST>), dim3(grid_for(n)), dim3(256), 0, stream, sk, sv, (const TT*)input, (TT*)values, indices, B, E, K, descending);
        return gnnops_check_launch("sort finish");
    }
}

inline size_t key_bytes(int st, bool flat) { return (st == ST_I64 || st == ST_F64 || !flat) ? 8 : 4; }

}  // namespace

extern "C" size_t gnnops_sort_workspace_bytes(int64_t B, int64_t E, int64_t K, int sort_dtype) {
    if (B < 0 || E < 0 || K < 0) return 0;
    const size_t n = (size_t)(B * E * K);
    const size_t tiles = (size_t)gnnops_cdiv(n > 0 ? (int64_t)n : 1, sortengine::TILE);
    const size_t keyb = key_bytes(sort_dtype, B * K == 1);
    return 2 * align_up(n * keyb, 256) + 2 * align_up(n * 4, 256) + align_up(256 * tiles * 4, 256) + 1024;
}

extern "C" int gnnops_sort(const void* input, void* values, int64_t* indices, int64_t B, int64_t E, int64_t K,
                           int sort_dtype, int descending, void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && E >= 0 && K >= 0, GNNOPS_EINVAL, "sort: negative size");
    GNNOPS_REQUIRE(sort_dtype >= ST_F32 && sort_dtype <= ST_F64, GNNOPS_EUNSUPPORTED, "sort: dtype code %d", sort_dtype);
    const int64_t n = B * E * K;
    if (n == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(n < ((int64_t)1 << 32), GNNOPS_EUNSUPPORTED, "sort: numel must be < 2^32 (got %lld)", (long long)n);
    GNNOPS_REQUIRE(input && values && indices, GNNOPS_EINVAL, "sort: null pointer");
    GNNOPS_REQUIRE(workspace && workspace_bytes >= gnnops_sort_workspace_bytes(B, E, K, sort_dtype), GNNOPS_EWORKSPACE,
                   "sort: workspace %zu < %zu", workspace_bytes, gnnops_sort_workspace_bytes(B, E, K, sort_dtype));
    const size_t keyb = key_bytes(sort_dtype, B * K == 1);
    Work w;
    w.tiles = (int)gnnops_cdiv(n, sortengine::TILE);
    char* p = (char*)workspace;
    w.keys_a = p; p += align_up((size_t)n * keyb, 256);
    w.keys_b = p; p += align_up((size_t)n * keyb, 256);
    w.vals_a = (uint32_t*)p; p += align_up((size_t)n * 4, 256);
    w.vals_b = (uint32_t*)p; p += align_up((size_t)n * 4, 256);
    w.tile_hist = (uint32_t*)p; p += align_up((size_t)256 * w.tiles * 4, 256);
    w.digit_total = (uint32_t*)p;
    switch (sort_dtype) {
        case ST_F32: return sort_typed<ST_F32>(input, values, indices, B, E, K, descending, w, stream);
        case ST_F16: return sort_typed<ST_F16>(input, values, indices, B, E, K, descending, w, stream);
        case ST_BF16: return sort_typed<ST_BF16>(input, values, indices, B, E, K, descending, w, stream);
        case ST_I32: return sort_typed<ST_I32>(input, values, indices, B, E, K, descending, w, stream);
        case ST_I64: return sort_typed<ST_I64>(input, values, indices, B, E, K, descending, w, stream);
        default: return sort_typed<ST_F64>(input, values, indices, B, E, K, descending, w, stream);
    }
}
