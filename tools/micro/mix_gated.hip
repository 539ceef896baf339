// Does a 5 : 1 read : write stream mix run faster when the whole chip writes at the same time and reads at the same time?
// Reads alone reach 6.3-7.0 TB/s and fills 5.6-5.9 on this part, the 5 : 1 mix 5.5-5.9 (profiles/round3_a_store_sweep*.txt):
// done one after the other the same bytes would take ~10 % less time than mixed. The mix kernel of store_sweep.hip
// (workgroup-contiguous chunks, U = 4, nontemporal stores) with its stores and / or loads gated on the chip-wide 100 MHz
// clock (s_memrealtime): stores only while (t mod P) < W, loads only outside that window. A workgroup holds its finished
// chunk in registers while it waits; more workgroups per CU cover the wait.
//   gate 0: none   1: stores inside the window   2: stores inside, loads outside
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mix_gated.hip -o tools/micro/mix_gated.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr int U = 4, RD = 5;

__device__ __forceinline__ bool in_window(uint32_t period, uint32_t window) {
    const uint64_t t = __builtin_amdgcn_s_memrealtime();
    return (uint32_t)(t % period) < window;
}

template <int GATE>
__global__ __launch_bounds__(256) void mix_kernel(const v4u* __restrict__ src, v4u* __restrict__ dst, int64_t nw, uint32_t period,
                                                  uint32_t window) {
    const int64_t chunk_words = (int64_t)U * 256;
    const int64_t chunks = nw / chunk_words;   // whole chunks only
    for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const int64_t w0 = c * chunk_words + threadIdx.x;
        if (GATE == 2) {
            int guard = 0;
            while (in_window(period, window) && ++guard < 100000) __builtin_amdgcn_s_sleep(8);
        }
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + w0 + u * 256);
#pragma unroll
        for (int r = 1; r < RD; ++r)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const v4u x = __builtin_nontemporal_load(src + w0 + u * 256 + (int64_t)r * nw);
                v[u] ^= x;
            }
        if (GATE >= 1) {
            // the loads must have landed before the wait starts, or the wait would only delay their use
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
            int guard = 0;
            while (!in_window(period, window) && ++guard < 100000) __builtin_amdgcn_s_sleep(8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], dst + w0 + u * 256);
    }
}

int main(int argc, char** argv) {
    const int64_t nw = (int64_t)64 << 20;   // 16-B words per stream: 1 GiB
    v4u *src, *dst;
    CK(hipMalloc(&src, (size_t)nw * 16 * RD));
    CK(hipMalloc(&dst, (size_t)nw * 16));
    CK(hipMemset(src, 1, (size_t)nw * 16 * RD));
    CK(hipMemset(dst, 0, (size_t)nw * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = (double)nw * 16 * (RD + 1);
    auto run = [&](int gate, int wg_per_cu, uint32_t period, uint32_t window) {
        const dim3 grid(256 * wg_per_cu), block(256);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (gate == 0) hipLaunchKernelGGL(mix_kernel<0>, grid, block, 0, 0, src, dst, nw, period, window);
            else if (gate == 1) hipLaunchKernelGGL(mix_kernel<1>, grid, block, 0, 0, src, dst, nw, period, window);
            else hipLaunchKernelGGL(mix_kernel<2>, grid, block, 0, 0, src, dst, nw, period, window);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("gate %d  wg/cu %2d  period %5.1f us  window %5.1f us : %7.3f ms  %5.2f TB/s\n", gate, wg_per_cu, period / 100.0,
               window / 100.0, best, bytes / best / 1e9);
        fflush(stdout);
    };
    for (int wg : {8, 16, 32}) run(0, wg, 1000, 200);
    for (int gate : {1, 2})
        for (int wg : {16, 32})
            for (uint32_t period : {400u, 1000u, 2000u, 4000u, 8000u})
                for (uint32_t frac : {15u, 20u, 30u}) run(gate, wg, period, period * frac / 100);
    for (int wg : {8, 16, 32}) run(0, wg, 1000, 200);
    return 0;
}
