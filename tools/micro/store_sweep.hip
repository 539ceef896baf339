// Why is a 1:1 copy 5.40 TB/s in row_stream.hip when MI355X_MICROARCH.md:36 quotes 6.29 TB/s for a float4 copy?
// Sweep of the STORE side of a streaming kernel (round 3):
//   pattern   GS   grid-stride, 16 B per lane, a lane's consecutive accesses one whole grid apart (what row_stream.hip's
//                  mix / fill kernels and abi.hip's diag kernel do)
//             WG   a workgroup owns U * 4 KiB of CONTIGUOUS bytes per iteration (U wave-instructions deep per wave)
//             WV   a wave owns U KiB of contiguous bytes per iteration
//   store     plain | nt | sc0 sc1 (write-through) | sc1 nt
//   U         loads issued before the first store (burst depth): 1, 4, 8, 16
//   grid      256 CUs x {2, 4, 8, 16, 32} workgroups of 256 threads (persistent grid-stride) — 8192 is what round 2 ran
//   shapes    copy (1 read : 1 write), fill (stores only), mix (5 reads : 1 write — the byte ratio of config 2's reduction)
// hipMemcpyDtoDAsync / hipMemsetAsync are timed beside them as the runtime's own answer.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_sweep.hip -o tools/micro/store_sweep.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

enum { ST_PLAIN = 0, ST_NT = 1, ST_WT = 2, ST_SC1NT = 3 };
enum { P_GS = 0, P_WG = 1, P_WV = 2 };

template <int ST>
__device__ __forceinline__ void store16(v4u* p, v4u v) {
    if (ST == ST_PLAIN) *p = v;
    else if (ST == ST_NT) __builtin_nontemporal_store(v, p);
    else if (ST == ST_WT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// index of the u-th 16-B word of this lane in iteration `it`; n16 words in all. Returns -1 past the end.
template <int PAT, int U>
__device__ __forceinline__ int64_t word_of(int64_t it, int u, int64_t n16) {
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    int64_t w;
    if (PAT == P_GS) {
        w = (it * U + u) * T + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    } else if (PAT == P_WG) {
        const int64_t chunk = it * gridDim.x + blockIdx.x;                  // chunk = U * blockDim.x words, contiguous
        w = chunk * (U * (int64_t)blockDim.x) + (int64_t)u * blockDim.x + threadIdx.x;
    } else {
        const int wpb = blockDim.x >> 6;
        const int64_t wave = (it * gridDim.x + blockIdx.x) * wpb + (threadIdx.x >> 6);   // wave-chunk = U * 64 words
        w = wave * (U * 64) + u * 64 + (threadIdx.x & 63);
    }
    return w < n16 ? w : -1;
}

template <int PAT, int U, int ST, int RD>   // RD reads per write (0 = fill)
__global__ __launch_bounds__(256) void sweep_kernel(const v4u* __restrict__ src, v4u* __restrict__ dst, int64_t n16) {
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t iters = (n16 + T * U - 1) / (T * U);
    for (int64_t it = 0; it < iters; ++it) {
        v4u v[U];
        int64_t w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = word_of<PAT, U>(it, u, n16);
        if (RD == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = v4u{1u, 2u, 3u, (uint32_t)w[u]};
        } else if (w[U - 1] >= 0) {   // full step: straight-line loads
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + w[u]);
#pragma unroll
            for (int r = 1; r < RD; ++r)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const v4u x = __builtin_nontemporal_load(src + w[u] + (int64_t)r * n16);
                    v[u].x ^= x.x; v[u].y ^= x.y; v[u].z ^= x.z; v[u].w ^= x.w;
                }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (w[u] >= 0) {
                    v[u] = __builtin_nontemporal_load(src + w[u]);
                    for (int r = 1; r < RD; ++r) {
                        const v4u x = __builtin_nontemporal_load(src + w[u] + (int64_t)r * n16);
                        v[u].x ^= x.x; v[u].y ^= x.y; v[u].z ^= x.z; v[u].w ^= x.w;
                    }
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (w[u] >= 0) store16<ST>(dst + w[u], v[u]);
    }
}

static hipEvent_t ev_a, ev_b;
template <typename F>
static float time_ms(int iters, F&& launch) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev_a));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(ev_b));
    CK(hipEventSynchronize(ev_b));
    float ms;
    CK(hipEventElapsedTime(&ms, ev_a, ev_b));
    return ms / iters;
}

static const char* PN[] = {"GS", "WG", "WV"};
static const char* SN[] = {"plain", "nt", "sc0sc1", "sc1nt"};

template <int PAT, int U, int ST, int RD>
static void one(const v4u* src, v4u* dst, int64_t n16, int wg_per_cu, int it) {
    const int grid = 256 * wg_per_cu;
    const float ms = time_ms(it, [&] { hipLaunchKernelGGL((sweep_kernel<PAT, U, ST, RD>), dim3(grid), dim3(256), 0, 0, src, dst, n16); });
    const double bytes = (double)n16 * 16 * (RD + 1);
    printf("%-5s rd=%d %-3s U=%-2d st=%-7s wg/cu=%-3d %8.3f ms %6.2f TB/s\n", RD == 0 ? "fill" : (RD == 1 ? "copy" : "mix"), RD, PN[PAT], U,
           SN[ST], wg_per_cu, ms, bytes / ms / 1e9);
    fflush(stdout);
}

template <int PAT, int U, int RD>
static void stores(const v4u* src, v4u* dst, int64_t n16, int wg, int it) {
    one<PAT, U, ST_PLAIN, RD>(src, dst, n16, wg, it);
    one<PAT, U, ST_NT, RD>(src, dst, n16, wg, it);
    one<PAT, U, ST_WT, RD>(src, dst, n16, wg, it);
    one<PAT, U, ST_SC1NT, RD>(src, dst, n16, wg, it);
}

template <int RD>
static void shape(const v4u* src, v4u* dst, int64_t n16, int it) {
    const int grids[] = {2, 4, 8, 16, 32};
    for (int g : grids) {
        stores<P_GS, 1, RD>(src, dst, n16, g, it);
        stores<P_GS, 4, RD>(src, dst, n16, g, it);
        stores<P_GS, 8, RD>(src, dst, n16, g, it);
        stores<P_WG, 4, RD>(src, dst, n16, g, it);
        stores<P_WG, 8, RD>(src, dst, n16, g, it);
        stores<P_WG, 16, RD>(src, dst, n16, g, it);
        stores<P_WV, 4, RD>(src, dst, n16, g, it);
        stores<P_WV, 8, RD>(src, dst, n16, g, it);
        stores<P_WV, 16, RD>(src, dst, n16, g, it);
    }
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 4.0;       // bytes WRITTEN per launch
    const int it = argc > 2 ? atoi(argv[2]) : 5;
    const int64_t n16 = (int64_t)(gb * 1e9 / 16) & ~int64_t(0xffff);
    printf("store sweep: %.2f GB written per launch (copy reads the same, mix reads 5x)\n", n16 * 16 / 1e9);
    v4u *src, *dst;
    CK(hipMalloc(&src, (size_t)n16 * 16 * 5));
    CK(hipMalloc(&dst, (size_t)n16 * 16));
    CK(hipMemset(src, 0x5a, (size_t)n16 * 16 * 5));
    CK(hipMemset(dst, 0, (size_t)n16 * 16));
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    {
        const float ms = time_ms(it, [&] { CK(hipMemcpyDtoDAsync((hipDeviceptr_t)dst, (hipDeviceptr_t)src, (size_t)n16 * 16, 0)); });
        printf("hipMemcpyDtoDAsync %8.3f ms %6.2f TB/s (read+write)\n", ms, 2.0 * n16 * 16 / ms / 1e9);
        const float ms2 = time_ms(it, [&] { CK(hipMemsetAsync(dst, 0x11, (size_t)n16 * 16, 0)); });
        printf("hipMemsetAsync     %8.3f ms %6.2f TB/s\n", ms2, 1.0 * n16 * 16 / ms2 / 1e9);
    }
    if (argc > 3 && !strcmp(argv[3], "pmc")) {   // a handful of forms, for a rocprofv3 --pmc pass (kernel names carry <PAT, U, ST, RD>)
        one<P_GS, 4, ST_NT, 1>(src, dst, n16, 16, 1);
        one<P_WG, 4, ST_NT, 1>(src, dst, n16, 16, 1);
        one<P_GS, 8, ST_PLAIN, 0>(src, dst, n16, 16, 1);
        one<P_WG, 8, ST_PLAIN, 0>(src, dst, n16, 16, 1);
        one<P_GS, 4, ST_NT, 5>(src, dst, n16, 16, 1);
        one<P_WG, 4, ST_NT, 5>(src, dst, n16, 16, 1);
        return 0;
    }
    shape<1>(src, dst, n16, it);
    shape<0>(src, dst, n16, it);
    shape<5>(src, dst, n16, it);
    // correctness spot check of the last mix form (src is 0x5a bytes: five xors = 0x5a5a5a5a)
    uint32_t h[4];
    CK(hipMemcpy(h, dst + n16 - 1, 16, hipMemcpyDeviceToHost));
    printf("last word after mix: %08x (expect 5a5a5a5a)\n", h[0]);
    return 0;
}
