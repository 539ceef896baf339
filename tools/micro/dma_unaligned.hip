// Does global_load_lds_dwordx4 (LDS-DMA) accept global addresses that are not 16-B aligned, and at what cost?
// One workgroup copies rows of a matrix whose pitch is 2*L bytes (L odd or L % 8 = 4) into LDS and back out; the host
// compares. Then a bandwidth loop over many workgroups for aligned vs misaligned pitch.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dma_unaligned.hip -o tools/micro/dma_unaligned.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) copy_rows(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int64_t pitch, int rows_per_block,
                                                 int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave: 16 rows x 64 B (4 lanes per row, 16 B each) per instruction -> 1 KiB of LDS at smem + wave * 1024
    for (int it = 0; it < iters; ++it) {
        const int64_t row = (int64_t)blockIdx.x * rows_per_block + ((it * 4 + wave) * 16 + (lane >> 2)) % rows_per_block;
        const uint16_t* p = src + row * pitch + (lane & 3) * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (it == 0) {
            const uint4 v = *reinterpret_cast<const uint4*>(smem + wave * 1024 + lane * 16);
            *reinterpret_cast<uint4*>(dst + ((int64_t)blockIdx.x * 64 + wave * 16 + (lane >> 2)) * 32 + (lane & 3) * 8) = v;
        }
    }
}

int main() {
    const int blocks = 2048, rpb = 64;
    for (int64_t pitch : {8192, 8164, 8165, 8166}) {
        const int64_t rows = (int64_t)blocks * rpb;
        std::vector<uint16_t> h(rows * pitch + 64);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(i * 2654435761u >> 7);
        uint16_t *src, *dst;
        hipMalloc(&src, h.size() * 2);
        hipMalloc(&dst, (size_t)blocks * 64 * 32 * 2);
        hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipMemset(dst, 0, (size_t)blocks * 64 * 32 * 2);
        copy_rows<<<blocks, 256>>>(src, dst, pitch, rpb, 1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("pitch %lld: kernel failed\n", (long long)pitch); return 1; }
        std::vector<uint16_t> o((size_t)blocks * 64 * 32);
        hipMemcpy(o.data(), dst, o.size() * 2, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int b = 0; b < blocks; ++b)
            for (int r = 0; r < 64; ++r)
                for (int c = 0; c < 32; ++c)
                    if (o[((size_t)b * 64 + r) * 32 + c] != h[((size_t)b * rpb + r) * pitch + c]) ++bad;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int iters = 2000;
        copy_rows<<<blocks, 256>>>(src, dst, pitch, rpb, iters);
        hipEventRecord(e0);
        copy_rows<<<blocks, 256>>>(src, dst, pitch, rpb, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("pitch %lld elements (row start %% 16 B = %lld): mismatches %zu; %d x 1 KiB pieces per wave: %.3f ms = %.2f TB/s into LDS\n",
               (long long)pitch, (long long)(pitch * 2 % 16), bad, iters, ms, (double)blocks * 4 * iters * 1024 / ms / 1e9);
        hipFree(src);
        hipFree(dst);
    }
    return 0;
}
