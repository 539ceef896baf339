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
// index per output row: its achievable rate is that of the 5 : 1 mix, not the 8 TB/s of the data sheet. Measured live,
// next to the kernel, because boxes differ by a few per cent.
// Round 3: the STORE SHAPE decides what this measures (tools/micro/store_sweep.hip, profiles/round3_store_sweep.txt). A
// grid-strided loop (every lane's consecutive 16-B stores one whole grid apart — this kernel until round 2) gives a 1 : 1
// copy 4.5-5.5 TB/s, a fill 4.2-5.3 and the 5 : 1 mix 5.0-5.4; when a workgroup owns U x 4 KiB of CONTIGUOUS bytes per
// step and issues its U stores back to back, the same streams run at 5.7-5.9 (copy), 5.6-5.9 (fill) and 5.7-5.8 (mix) for
// every grid from 4 to 32 workgroups per CU, nontemporal stores 4 % ahead of plain ones. That plateau is the ceiling.
namespace {
constexpr int MIX_U = 4;
template <int RD>
__global__ __launch_bounds__(256) void stream_mix_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t nw) {
    const int64_t chunk_words = (int64_t)MIX_U * 256;
    for (int64_t c = blockIdx.x; c * chunk_words < nw; c += gridDim.x) {
        const int64_t w0 = c * chunk_words + threadIdx.x;
        u32x4 v[MIX_U];
        if ((c + 1) * chunk_words <= nw) {   // full step, straight-line: every load of a phase in flight
#pragma unroll
            for (int u = 0; u < MIX_U; ++u) v[u] = load16<true>(src + w0 + u * 256);
#pragma unroll
            for (int r = 1; r < RD; ++r)
#pragma unroll
                for (int u = 0; u < MIX_U; ++u) {
                    const u32x4 x = load16<true>(src + w0 + u * 256 + (int64_t)r * nw);
                    v[u].x ^= x.x; v[u].y ^= x.y; v[u].z ^= x.z; v[u].w ^= x.w;
                }
#pragma unroll
            for (int u = 0; u < MIX_U; ++u) {
                if (dst) store16<true>(dst + w0 + u * 256, v[u]);
                else if (v[u].x == 0x12345678u && v[u].y == 0x9abcdef0u) const_cast<u32x4*>(src)[0] = v[u];   // keeps the reads alive
            }
        } else {
            for (int u = 0; u < MIX_U; ++u) {
                const int64_t w = w0 + u * 256;
                if (w >= nw) break;
                u32x4 a = load16<true>(src + w);
                for (int r = 1; r < RD; ++r) {
                    const u32x4 x = load16<true>(src + w + (int64_t)r * nw);
                    a.x ^= x.x; a.y ^= x.y; a.z ^= x.z; a.w ^= x.w;
                }
                if (dst) store16<true>(dst + w, a);
                else if (a.x == 0x12345678u && a.y == 0x9abcdef0u) const_cast<u32x4*>(src)[0] = a;
            }
        }
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
