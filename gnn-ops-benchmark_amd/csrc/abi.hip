// abi.hip — version / error plumbing of the C ABI (include/gnnops.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void gnnops_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int gnnops_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gnnops_set_error("%s: %s", what, hipGetErrorString(e));
        return GNNOPS_ELAUNCH;
    }
    return GNNOPS_OK;
}

extern "C" int gnnops_version(void) { return GNNOPS_ABI_VERSION; }
extern "C" const char* gnnops_last_error(void) { return g_err; }

// ---- measurement aid (bench.py's roofline leg) -------------------------------------------------------------------
// What this box's memory system gives a plain stream MIX: `reads` sequential 16-B nontemporal read streams per one
// nontemporal write stream, no index and no row structure. The segment reduction of config 2 moves 5 source rows and the
// index per output row: its achievable rate is that of the 5 : 1 mix (reads alone run at ~6.8 TB/s, the 5 : 1 mix at
// ~5.3 TB/s on MI355X — writes cost HBM more than reads), not the 8 TB/s of the data sheet. Measured live, next to the
// kernel, because boxes differ by a few per cent.
namespace {
template <int RD>
__global__ __launch_bounds__(256) void stream_mix_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t nw) {
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = gtid; i < nw; i += stride) {
        u32x4 v[RD];
#pragma unroll
        for (int u = 0; u < RD; ++u) v[u] = load16<true>(src + i + (int64_t)u * nw);
        u32x4 a = v[0];
#pragma unroll
        for (int u = 1; u < RD; ++u) { a.x ^= v[u].x; a.y ^= v[u].y; a.z ^= v[u].z; a.w ^= v[u].w; }
        if (dst) store16<true>(dst + i, a);
        else if (a.x == 0x12345678u && a.y == 0x9abcdef0u) const_cast<u32x4*>(src)[0] = a;   // keeps the reads alive
    }
}
}  // namespace

extern "C" int gnnops_diag_stream_mix(const void* src, void* dst, int64_t pieces, int reads, gnnops_stream_t s) {
    GNNOPS_REQUIRE(pieces >= 0 && src && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), GNNOPS_EINVAL,
                   "diag_stream_mix: bad argument");
    if (pieces == 0) return GNNOPS_OK;
    const dim3 grid(256 * 32), block(256);
    switch (reads) {
        case 1: hipLaunchKernelGGL(stream_mix_kernel<1>, grid, block, 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, pieces); break;
        case 2: hipLaunchKernelGGL(stream_mix_kernel<2>, grid, block, 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, pieces); break;
        case 3: hipLaunchKernelGGL(stream_mix_kernel<3>, grid, block, 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, pieces); break;
        case 4: hipLaunchKernelGGL(stream_mix_kernel<4>, grid, block, 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, pieces); break;
        case 5: hipLaunchKernelGGL(stream_mix_kernel<5>, grid, block, 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, pieces); break;
        case 8: hipLaunchKernelGGL(stream_mix_kernel<8>, grid, block, 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, pieces); break;
        default: gnnops_set_error("diag_stream_mix: reads must be 1..5 or 8 (got %d)", reads); return GNNOPS_EINVAL;
    }
    return gnnops_check_launch("diag_stream_mix");
}
