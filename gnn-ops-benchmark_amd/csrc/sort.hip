// sort.hip — torch.sort(input, dim, stable) for fp32 (reference: op_bm_scripts/benchmark_native_sort.py:28-30;
// shapes 1-D 8e8, (20000,20000), 800^3, dims 0..2, stable in {True, False}, tie-heavy dropout inputs).
//
// Ascending, always stable (a stable order is a valid answer for stable=False too). Built on the radix
// engine of the plan builder:
//   1-D            4 passes over u32 keys (order-preserving image of the float), values = positions
//   along a dim    one sort of 64-bit keys (segment << 32 | float image): 4 + ceil(log2(segments)/8)
//                  passes; the element is fed in memory order, so ties keep ascending position
// -0.0 is keyed like +0.0 (torch compares them equal) and returned as +0.0; every NaN sorts last.
// HBM-bound: per pass hist reads the keys, scatter reads and writes keys + values.
#include "common.h"
#include "sort_engine.h"

namespace {

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

#define GRID_STRIDE(i, total) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

__global__ void build_keys64_kernel(const float* __restrict__ in, uint64_t* __restrict__ keys, int64_t B, int64_t E,
                                    int64_t K) {
    GRID_STRIDE(i, B * E * K) {
        const int64_t k = i % K;
        const int64_t b = i / (E * K);
        keys[i] = ((uint64_t)(b * K + k) << 32) | (uint64_t)f32_key(in[i]);
    }
}
__global__ void finish32_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                float* __restrict__ values, int64_t* __restrict__ indices, int64_t n) {
    GRID_STRIDE(p, n) {
        values[p] = key_f32(keys[p]);
        indices[p] = (int64_t)vals[p];
    }
}
// sorted position p = seg*E + r  ->  output element [b, r, k]; source position e from the linear index.
__global__ void finish64_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                float* __restrict__ values, int64_t* __restrict__ indices, int64_t B, int64_t E,
                                int64_t K) {
    GRID_STRIDE(p, B * E * K) {
        const int64_t seg = p / E, r = p % E;
        const int64_t b = seg / K, k = seg % K;
        const int64_t o = (b * E + r) * K + k;
        values[o] = key_f32((uint32_t)keys[p]);
        indices[o] = ((int64_t)vals[p] / K) % E;
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int grid_for(int64_t n) { return gnnops_grid_cap(gnnops_cdiv(n, 256), 256 * 16); }

}  // namespace

extern "C" size_t gnnops_sort_workspace_bytes(int64_t B, int64_t E, int64_t K) {
    if (B < 0 || E < 0 || K < 0) return 0;
    const size_t n = (size_t)(B * E * K);
    const size_t tiles = (size_t)gnnops_cdiv(n > 0 ? (int64_t)n : 1, sortengine::TILE);
    const size_t keyb = (B * K == 1) ? 4 : 8;
    return 2 * align_up(n * keyb, 256) + 2 * align_up(n * 4, 256) + align_up(256 * tiles * 4, 256) + 1024;
}

extern "C" int gnnops_sort_f32(const float* input, float* values, int64_t* indices, int64_t B, int64_t E, int64_t K,
                               void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(B >= 0 && E >= 0 && K >= 0, GNNOPS_EINVAL, "sort: negative size");
    const int64_t n = B * E * K;
    if (n == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(n < ((int64_t)1 << 32), GNNOPS_EUNSUPPORTED, "sort: numel must be < 2^32 (got %lld)", (long long)n);
    GNNOPS_REQUIRE(input && values && indices, GNNOPS_EINVAL, "sort: null pointer");
    GNNOPS_REQUIRE(workspace && workspace_bytes >= gnnops_sort_workspace_bytes(B, E, K), GNNOPS_EWORKSPACE,
                   "sort: workspace %zu < %zu", workspace_bytes, gnnops_sort_workspace_bytes(B, E, K));
    const int tiles = (int)gnnops_cdiv(n, sortengine::TILE);
    const bool flat = (B * K == 1);
    const size_t keyb = flat ? 4 : 8;
    char* w = (char*)workspace;
    void* keys_a = w; w += align_up((size_t)n * keyb, 256);
    void* keys_b = w; w += align_up((size_t)n * keyb, 256);
    uint32_t* vals_a = (uint32_t*)w; w += align_up((size_t)n * 4, 256);
    uint32_t* vals_b = (uint32_t*)w; w += align_up((size_t)n * 4, 256);
    uint32_t* tile_hist = (uint32_t*)w; w += align_up((size_t)256 * tiles * 4, 256);
    uint32_t* digit_total = (uint32_t*)w;

    int rc;
    if (flat) {
        // pass 0 reads the floats directly (KeyF32 adapter), values implicit
        rc = sortengine::pass_first_f32(input, (uint32_t*)keys_a, vals_a, n, 0, tile_hist, digit_total, tiles, stream);
        if (rc) return rc;
        rc = sortengine::pass_u32((uint32_t*)keys_a, vals_a, (uint32_t*)keys_b, vals_b, n, 8, tile_hist, digit_total, tiles, stream);
        if (rc) return rc;
        rc = sortengine::pass_u32((uint32_t*)keys_b, vals_b, (uint32_t*)keys_a, vals_a, n, 16, tile_hist, digit_total, tiles, stream);
        if (rc) return rc;
        rc = sortengine::pass_u32((uint32_t*)keys_a, vals_a, (uint32_t*)keys_b, vals_b, n, 24, tile_hist, digit_total, tiles, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(finish32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (const uint32_t*)keys_b, vals_b, values, indices, n);
        return gnnops_check_launch("sort finish");
    }
    const int64_t segs = B * K;
    int segbits = 0;
    while (((int64_t)1 << segbits) < segs) ++segbits;
    const int passes = 4 + (segbits + 7) / 8;
    hipLaunchKernelGGL(build_keys64_kernel, dim3(grid_for(n)), dim3(256), 0, stream, input, (uint64_t*)keys_b, B, E, K);
    uint64_t* kin = (uint64_t*)keys_b;
    uint64_t* kout = (uint64_t*)keys_a;
    uint32_t* vin = nullptr;
    uint32_t* vout = vals_a;
    for (int p = 0; p < passes; ++p) {
        if (p == 0)
            rc = sortengine::pass_first_u64(kin, kout, vout, n, 0, tile_hist, digit_total, tiles, stream);
        else
            rc = sortengine::pass_u64(kin, vin, kout, vout, n, 8 * p, tile_hist, digit_total, tiles, stream);
        if (rc) return rc;
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* nv = (vout == vals_a) ? vals_b : vals_a;
        vin = vout; vout = nv;
    }
    // after the swap `kin` / `vin` hold the sorted data
    hipLaunchKernelGGL(finish64_kernel, dim3(grid_for(n)), dim3(256), 0, stream, kin, vin, values, indices, B, E, K);
    return gnnops_check_launch("sort finish");
}
