// What does the memory system give a segment reduction over 512-B rows (config 2: D = 128 fp32)?
//   stream      pure sequential 16-B nontemporal reads, U in flight per lane (the read ceiling)
//   seg_regs    the shape of bucket_reduce_kernel / seg_rows_kernel: a 32-lane group per destination, its rows gathered into
//               registers U at a time, summed in order, one nontemporal 512-B store per destination
//   seg_dma     the same sums with the rows fetched by LDS-DMA (global_load_lds_dwordx4, two rows per wave-instruction) into
//               a per-wave ring of R KiB; every lane owns two columns and adds the rows in order from LDS (ds_read_b64)
// Destinations have DEG rows each (fixed), rows taken through `perm` (identity or a random permutation).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/row_stream.hip -o tools/micro/row_stream.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ROWF = 128;           // floats per row
constexpr int DEG = 5;

template <int U>
__global__ __launch_bounds__(256) void stream_kernel(const v4u* __restrict__ src, int64_t n16, uint32_t* __restrict__ sink) {
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (int64_t i = gtid; i + (U - 1) * stride < n16; i += U * stride) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// STORE: 0 none (one row in 4096 keeps the sums alive), 1 nontemporal, 2 plain; LNT: nontemporal loads
template <int U, int WPS, int STORE = 1, bool LNT = true>
__global__ __launch_bounds__(256, WPS) void seg_regs_kernel(const float* __restrict__ src, const int32_t* __restrict__ perm,
                                                            float* __restrict__ out, int64_t N) {
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> 5;
    const int gl = (int)(gtid & 31);
    for (int64_t d = gtid >> 5; d < N; d += ngroups) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int64_t beg = d * DEG, end = beg + DEG;
        for (int64_t j = beg; j < end; j += U) {
            int32_t e[U];
            v4u rows[U];
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = (j + u < end) ? perm[j + u] : -1;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (e[u] >= 0) {
                    const v4u* rp = reinterpret_cast<const v4u*>(src + (int64_t)e[u] * ROWF) + gl;
                    rows[u] = LNT ? __builtin_nontemporal_load(rp) : *rp;
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (e[u] >= 0) {
                    acc[0] += __uint_as_float(rows[u].x); acc[1] += __uint_as_float(rows[u].y);
                    acc[2] += __uint_as_float(rows[u].z); acc[3] += __uint_as_float(rows[u].w);
                }
        }
        v4u o = {__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
        v4u* op = reinterpret_cast<v4u*>(out + d * ROWF) + gl;
        if (STORE == 1) __builtin_nontemporal_store(o, op);
        else if (STORE == 2) *op = o;
        else if ((d & 4095) == 0) *op = o;
    }
}

// Sequential mix of RD reads per write (16 B per lane each), the byte ratio of the segment reduction, with no index and no
// row structure: what the memory system gives that MIX as plain streams. SNT: nontemporal stores.
template <int RD, bool SNT>
__global__ __launch_bounds__(256) void mix_kernel(const v4u* __restrict__ src, v4u* __restrict__ dst, int64_t nw) {
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = gtid; i < nw; i += stride) {
        v4u v[RD];
#pragma unroll
        for (int u = 0; u < RD; ++u) v[u] = __builtin_nontemporal_load(src + i + (int64_t)u * nw);
        v4u a = v[0];
#pragma unroll
        for (int u = 1; u < RD; ++u) { a.x ^= v[u].x; a.y ^= v[u].y; a.z ^= v[u].z; a.w ^= v[u].w; }
        if (SNT) __builtin_nontemporal_store(a, dst + i); else dst[i] = a;
    }
}
template <bool SNT>
__global__ __launch_bounds__(256) void fill_kernel(v4u* __restrict__ dst, int64_t nw) {
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    const v4u a = {1u, 2u, 3u, (uint32_t)gtid};
    for (int64_t i = gtid; i < nw; i += stride) { if (SNT) __builtin_nontemporal_store(a, dst + i); else dst[i] = a; }
}

// One wave = a contiguous range of destinations; row pairs (2t, 2t+1) of its entry list go into ring slot t % R by ONE
// LDS-DMA instruction (lanes 0-31: row 2t, lanes 32-63: row 2t+1, 16 B each -> 1 KiB). Consumption: lane l owns columns
// 2l, 2l+1 and adds row 2t then row 2t+1 (sequential order), storing a destination when its DEG rows are in.
template <int R, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void seg_dma_kernel(const float* __restrict__ src, const int32_t* __restrict__ perm,
                                                            float* __restrict__ out, int64_t N, int64_t dst_per_wave) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring_all + wave * (R * 1024);
    const int64_t gw = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t d0 = gw * dst_per_wave;
    if (d0 >= N) return;
    const int64_t d1 = (d0 + dst_per_wave < N) ? d0 + dst_per_wave : N;
    const int64_t e0 = d0 * DEG, e1 = d1 * DEG;          // entries [e0, e1)
    const int64_t npairs = (e1 - e0 + 1) / 2;
    const bool hi_half = lane >= 32;
    const int l32 = lane & 31;

    // rows of pair t (wave-uniform scalar loads; clamped past the end: fillers are never consumed)
    auto pair_rows = [&](int64_t t, int32_t& lo, int32_t& hi) {
        const int64_t ea = e0 + 2 * t, eb = ea + 1;
        lo = perm[ea < e1 ? ea : e1 - 1];
        hi = perm[eb < e1 ? eb : e1 - 1];
    };
    auto issue = [&](int64_t t, int32_t lo, int32_t hi) {  // pair t -> slot t % R
        const int32_t r = hi_half ? hi : lo;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)r * ROWF + l32 * 4),
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(ring + (uint32_t)(t & (R - 1)) * 1024), 16, 0, 0);
    };
    int32_t nlo, nhi;
#pragma unroll
    for (int t = 0; t < R; ++t) { pair_rows(t, nlo, nhi); issue(t, nlo, nhi); }
    pair_rows(R, nlo, nhi);   // rows of the next pair to issue, loaded one step ahead
    float a0 = 0.f, a1 = 0.f;
    int64_t d = d0;
    int in_dst = 0;
    for (int64_t t = 0; t < npairs; ++t) {
        // the oldest outstanding DMA is pair t: all but the R-1 younger operations must have completed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory");
        const uint32_t addr = ring + (uint32_t)(t & (R - 1)) * 1024 + lane * 8;
        v2f ra, rb;
        asm volatile("ds_read_b64 %0, %1" : "=v"(ra) : "v"(addr));
        asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(rb) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra), "+v"(rb));
        // the slot's bytes are in registers: refill it with pair t + R
        issue(t + R, nlo, nhi);
        pair_rows(t + R + 1, nlo, nhi);
        a0 += ra.x; a1 += ra.y;
        if (++in_dst == DEG) {
            v2f o = {a0, a1};
            __builtin_nontemporal_store(o, reinterpret_cast<v2f*>(out + d * ROWF) + lane);
            a0 = a1 = 0.f; in_dst = 0; ++d;
        }
        if (e0 + 2 * t + 1 < e1) {
            a0 += rb.x; a1 += rb.y;
            if (++in_dst == DEG) {
                v2f o = {a0, a1};
                __builtin_nontemporal_store(o, reinterpret_cast<v2f*>(out + d * ROWF) + lane);
                a0 = a1 = 0.f; in_dst = 0; ++d;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static float run(const char* name, double bytes, int iters, void (*launch)(void*), void* ctx) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(ctx);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch(ctx);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= iters;
    printf("%-44s %8.3f ms  %6.2f TB/s\n", name, ms, bytes / ms / 1e9);
    fflush(stdout);
    return ms;
}

struct Ctx { const float* src; const int32_t* perm; float* out; int64_t E, N; uint32_t* sink; };

template <int U> void l_stream(void* c) { Ctx* x = (Ctx*)c; hipLaunchKernelGGL(stream_kernel<U>, dim3(256 * 32), dim3(256), 0, 0, (const v4u*)x->src, x->E * 32, x->sink); }
template <int U, int WPS, int STORE = 1, bool LNT = true> void l_regs(void* c) { Ctx* x = (Ctx*)c; hipLaunchKernelGGL((seg_regs_kernel<U, WPS, STORE, LNT>), dim3(256 * 64), dim3(256), 0, 0, x->src, x->perm, x->out, x->N); }
template <int RD, bool SNT> void l_mix(void* c) { Ctx* x = (Ctx*)c; hipLaunchKernelGGL((mix_kernel<RD, SNT>), dim3(256 * 32), dim3(256), 0, 0, (const v4u*)x->src, (v4u*)x->out, x->N * 32); }
template <bool SNT> void l_fill(void* c) { Ctx* x = (Ctx*)c; hipLaunchKernelGGL((fill_kernel<SNT>), dim3(256 * 32), dim3(256), 0, 0, (v4u*)x->out, x->N * 32); }
template <int R, int WAVES, int WG_PER_CU> void l_dma(void* c) {
    Ctx* x = (Ctx*)c;
    // LDS per workgroup padded so that exactly WG_PER_CU workgroups fit a CU's 160 KiB
    int lds = R * WAVES * 1024;
    const int share = (160 * 1024 / WG_PER_CU) & ~1023;
    if (share > lds) lds = share;
    static bool cfg = false;
    if (!cfg) { CK(hipFuncSetAttribute((const void*)&seg_dma_kernel<R, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); cfg = true; }
    const int64_t waves = (int64_t)256 * WG_PER_CU * WAVES * 4;   // a few waves' worth of work per resident wave
    const int64_t dpw = (x->N + waves - 1) / waves;
    const int grid = (int)((x->N + dpw * WAVES - 1) / (dpw * WAVES));
    hipLaunchKernelGGL((seg_dma_kernel<R, WAVES>), dim3(grid), dim3(WAVES * 64), lds, 0, x->src, x->perm, x->out, x->N, dpw);
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 4000000;   // destinations; E = 5 N rows of 512 B
    const int64_t E = N * DEG;
    printf("N=%lld E=%lld src %.2f GB out %.2f GB\n", (long long)N, (long long)E, E * 512 / 1e9, N * 512 / 1e9);
    std::vector<float> h((size_t)E * ROWF);
    std::mt19937 rng(42);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((rng() >> 8) & 0xffff) / 65536.f;
    std::vector<int32_t> ident(E), rnd(E);
    for (int64_t i = 0; i < E; ++i) ident[i] = rnd[i] = (int32_t)i;
    for (int64_t i = E - 1; i > 0; --i) { int64_t j = rng() % (i + 1); std::swap(rnd[i], rnd[j]); }
    float *src, *out; int32_t *p_id, *p_rnd; uint32_t* sink;
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&out, (size_t)N * ROWF * 4));
    CK(hipMalloc(&p_id, E * 4)); CK(hipMalloc(&p_rnd, E * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(p_id, ident.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(p_rnd, rnd.data(), E * 4, hipMemcpyHostToDevice));

    // correctness of the DMA kernel on the random permutation (sampled destinations, exact sequential sums)
    {
        Ctx c{src, p_rnd, out, E, N, sink};
        CK(hipMemset(out, 0xff, (size_t)N * ROWF * 4));
        l_dma<8, 4, 4>(&c);
        CK(hipDeviceSynchronize());
        std::vector<float> o((size_t)N * ROWF);
        CK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (int64_t d = 0; d < N; d += 997) {
            for (int k = 0; k < ROWF; ++k) {
                float a = 0.f;
                for (int j = 0; j < DEG; ++j) a += h[(size_t)rnd[d * DEG + j] * ROWF + k];
                if (a != o[(size_t)d * ROWF + k]) ++bad;
            }
        }
        // every destination written?
        size_t unwritten = 0;
        for (int64_t d = 0; d < N; ++d) { uint32_t u; memcpy(&u, &o[(size_t)d * ROWF], 4); if (u == 0xffffffffu) ++unwritten; }
        printf("seg_dma check: %zu mismatches on sampled rows, %zu unwritten destinations\n", bad, unwritten);
    }

    const double rd = (double)E * 512, all = rd + (double)N * 512 + (double)E * 4;
    Ctx ci{src, p_id, out, E, N, sink}, cr{src, p_rnd, out, E, N, sink};
    const int it = 5;
    run("stream read U=4", rd, it, l_stream<4>, &ci);
    run("stream read U=8", rd, it, l_stream<8>, &ci);
    run("stream read U=16", rd, it, l_stream<16>, &ci);
    run("mix 5 reads : 1 write, nt stores", rd + (double)N * 512, it, l_mix<5, true>, &ci);
    run("mix 5 reads : 1 write, plain stores", rd + (double)N * 512, it, l_mix<5, false>, &ci);
    run("mix 1 read : 1 write (copy), nt stores", 2.0 * N * 512, it, l_mix<1, true>, &ci);
    run("fill nt stores", (double)N * 512, it, l_fill<true>, &ci);
    run("fill plain stores", (double)N * 512, it, l_fill<false>, &ci);
    run("seg_regs U=8 nt loads, nt stores, identity", all, it, l_regs<8, 4, 1, true>, &ci);
    run("seg_regs U=8 nt loads, nt stores, random", all, it, l_regs<8, 4, 1, true>, &cr);
    run("seg_regs U=8 nt loads, NO stores, identity", all - (double)N * 512, it, l_regs<8, 4, 0, true>, &ci);
    run("seg_regs U=8 nt loads, NO stores, random", all - (double)N * 512, it, l_regs<8, 4, 0, true>, &cr);
    run("seg_regs U=8 plain loads, NO stores, random", all - (double)N * 512, it, l_regs<8, 4, 0, false>, &cr);
    run("seg_regs U=8 nt loads, plain stores, random", all, it, l_regs<8, 4, 2, true>, &cr);
    run("seg_regs U=8 plain loads, plain stores, random", all, it, l_regs<8, 4, 2, false>, &cr);
    run("seg_regs U=8 plain loads, nt stores, random", all, it, l_regs<8, 4, 1, false>, &cr);
    run("seg_dma R=8  4 waves x 4 WG/CU random", all, it, l_dma<8, 4, 4>, &cr);
    return 0;
}
