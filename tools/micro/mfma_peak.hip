// Practical MFMA ceiling of the box: v_mfma_f32_16x16x32_bf16 back to back from registers (no memory traffic), every
// SIMD busy. Prints TFLOP/s for 1, 2 and 4 waves per SIMD. Context for gemm.hip's fraction of the 2.5 PF paper peak.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <int ACC>
__global__ void __launch_bounds__(256) mfma_loop(float* sink, int iters, int zero_data) {
    bf16x8 a, b;
    // random operands in [-1, 1): all-zero or small-integer operands let the chip hold a higher clock (MI355X_MICROARCH.md, DVFS)
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        a[i] = zero_data ? (__bf16)0.f : (__bf16)((float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f);
        h = h * 1664525u + 1013904223u;
        b[i] = zero_data ? (__bf16)0.f : (__bf16)((float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f);
    }
    f32x4 c0{0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int it = 0; it < iters; ++it) {   // inline asm: the builtin form made the compiler shuffle AGPRs inside the loop
        MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3);
        if (ACC == 8) { MFMA(c4); MFMA(c5); MFMA(c6); MFMA(c7); }
    }
    f32x4 t = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    float s = t[0] + t[1] + t[2] + t[3];
    if (s == 12345.678f) sink[0] = s;
}

template <int ACC>
static void run(int waves_per_simd, float* sink, int zero_data = 0) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    const int iters = 400000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_loop<ACC><<<blocks, 256>>>(sink, iters, zero_data);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * ACC * 16.0 * 16 * 32 * 2;
        printf("%s CUs %d  waves/SIMD %d  independent accumulators %d  rep %d: %.2f ms  %.1f TFLOP/s\n", zero_data ? "zeros " : "random", cus, waves_per_simd, ACC, rep,
               ms, flop / ms / 1e9);
    }
}

int main() {
    float* sink;
    hipMalloc(&sink, 4);
    run<4>(1, sink);
    run<8>(1, sink);
    run<4>(2, sink);
    run<8>(2, sink);
    run<4>(4, sink);
    run<8>(2, sink, 1);
    return 0;
}
