// Which STORE schedule lets a segment reduction over 512-B rows (config 2: D = 128 fp32, mean degree 5, uniform random
// destinations) run at the rate store_sweep.hip found for contiguous-chunk stores (5 reads : 1 write at 5.75-6.0 TB/s)
// instead of the grid-stride rate (5.3)? Plan form (rowptr int32, perm int32 from a real random graph, binomial degrees);
// one 32-lane group per destination row, U = 8 gathered rows in flight, fp32 sums in source order — the loop of
// seg_rows_kernel / bucket_reduce_kernel. Variants differ ONLY in which rows a group takes and when / how it stores:
//   gs          grid-stride destinations (seg_rows_kernel today): a workgroup's 8 groups store 8 adjacent rows, then jump
//   blk         a workgroup owns 256 consecutive rows, group g takes rows g, g+8, ... (bucket_reduce_kernel today)
//   run         a workgroup owns 256 consecutive rows, group g takes the 32 consecutive rows g*32 ...
//   burst<RB>   as blk, but a wave takes 2*RB consecutive rows, keeps the RB finished rows of each half in registers and
//               stores them back to back: store instruction i writes rows 2i, 2i+1 = 1 KiB contiguous, RB KiB per burst
//   tile<TR>    a workgroup stages TR finished rows in LDS (double buffered, one barrier per round) and writes them out
//               as one contiguous TR*512-B run, 16 B per lane, all threads
// store flavours: nt | plain | sc1 nt.   build: hipcc --offload-arch=gfx950 -O3 tools/micro/seg_store.hip -o tools/micro/seg_store.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr int ROWF = 128;   // floats per row
constexpr int U = 8;
enum { ST_NT = 0, ST_PLAIN = 1, ST_SC1NT = 2 };

template <int ST>
__device__ __forceinline__ void store16(v4u* p, v4u v) {
    if (ST == ST_PLAIN) *p = v;
    else if (ST == ST_NT) __builtin_nontemporal_store(v, p);
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// the reduction of one destination row by a 32-lane group: rows perm[beg..end) of src, summed in order
__device__ __forceinline__ v4u reduce_row(const float* __restrict__ src, const int32_t* __restrict__ perm, int32_t beg, int32_t end, int gl) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int32_t j = beg; j < end; j += U) {
        int32_t e[U];
        v4u rows[U];
#pragma unroll
        for (int u = 0; u < U; ++u) e[u] = (j + u < end) ? perm[j + u] : -1;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e[u] >= 0) rows[u] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(src + (int64_t)e[u] * ROWF) + gl);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e[u] >= 0) {
                acc[0] += __uint_as_float(rows[u].x); acc[1] += __uint_as_float(rows[u].y);
                acc[2] += __uint_as_float(rows[u].z); acc[3] += __uint_as_float(rows[u].w);
            }
    }
    return v4u{__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
}

template <int ST>
__global__ __launch_bounds__(256, 4) void gs_kernel(const float* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ perm, float* __restrict__ out, int64_t N) {
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> 5;
    const int gl = (int)(gtid & 31);
    for (int64_t d = gtid >> 5; d < N; d += ngroups) {
        const v4u r = reduce_row(src, perm, rowptr[d], rowptr[d + 1], gl);
        store16<ST>(reinterpret_cast<v4u*>(out + d * ROWF) + gl, r);
    }
}

// MODE 0: group g takes rows g, g+8, ...; MODE 1: group g takes rows g*32 .. g*32+31
template <int ST, int MODE>
__global__ __launch_bounds__(256, 4) void blk_kernel(const float* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ perm, float* __restrict__ out, int64_t N) {
    const int gl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t chunks = (N + 255) >> 8;
    for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        for (int k = 0; k < 32; ++k) {
            const int64_t d = c * 256 + (MODE == 0 ? k * 8 + g : g * 32 + k);
            if (d >= N) continue;
            const v4u r = reduce_row(src, perm, rowptr[d], rowptr[d + 1], gl);
            store16<ST>(reinterpret_cast<v4u*>(out + d * ROWF) + gl, r);
        }
    }
}

template <int ST, int RB>
__global__ __launch_bounds__(256, 4) void burst_kernel(const float* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ perm, float* __restrict__ out, int64_t N) {
    const int lane = threadIdx.x & 63, gl = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    const int64_t chunks = (N + 255) >> 8;
    for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        for (int rb = 0; rb < 256 / (8 * RB); ++rb) {
            const int64_t row0 = c * 256 + rb * 8 * RB + w * 2 * RB;   // the wave's 2*RB consecutive rows
            v4u res[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int64_t d = row0 + 2 * i + h;
                int32_t beg = 0, end = 0;
                if (d < N) { beg = rowptr[d]; end = rowptr[d + 1]; }
                res[i] = reduce_row(src, perm, beg, end, gl);
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int64_t d = row0 + 2 * i + h;
                if (d < N) store16<ST>(reinterpret_cast<v4u*>(out + d * ROWF) + gl, res[i]);
            }
        }
    }
}

template <int ST, int TR>
__global__ __launch_bounds__(256, 4) void tile_kernel(const float* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ perm, float* __restrict__ out, int64_t N) {
    __shared__ v4u tile[2][TR * 32];
    const int gl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t chunks = (N + 255) >> 8;
    int buf = 0;
    for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        for (int rb = 0; rb < 256 / TR; ++rb) {
            const int64_t row0 = c * 256 + rb * TR;
#pragma unroll
            for (int i = 0; i < TR / 8; ++i) {
                const int r = i * 8 + g;
                const int64_t d = row0 + r;
                int32_t beg = 0, end = 0;
                if (d < N) { beg = rowptr[d]; end = rowptr[d + 1]; }
                tile[buf][r * 32 + gl] = reduce_row(src, perm, beg, end, gl);
            }
            __syncthreads();
            v4u* o = reinterpret_cast<v4u*>(out + row0 * ROWF);
            const int64_t lim = (N - row0) * 32;   // 16-B words that exist behind row0
#pragma unroll
            for (int u = 0; u < TR / 8; ++u) {
                const int wd = u * 256 + threadIdx.x;
                if (wd < lim) store16<ST>(o + wd, tile[buf][wd]);
            }
            buf ^= 1;
        }
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

struct Ctx { const float* src; const int32_t *rowptr, *perm; float* out; int64_t N, E; std::vector<float>* h; std::vector<int32_t>*hr, *hp; };

static void verify(const char* name, Ctx& c) {
    std::vector<float> o((size_t)c.N * ROWF);
    CK(hipMemcpy(o.data(), c.out, o.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int64_t d = 0; d < c.N; d += 1009)
        for (int k = 0; k < ROWF; ++k) {
            float a = 0.f;
            for (int32_t j = (*c.hr)[d]; j < (*c.hr)[d + 1]; ++j) a += (*c.h)[(size_t)(*c.hp)[j] * ROWF + k];
            if (a != o[(size_t)d * ROWF + k]) ++bad;
        }
    for (int64_t d = c.N - 300; d < c.N; ++d)   // the ragged tail
        for (int k = 0; k < ROWF; ++k) {
            float a = 0.f;
            for (int32_t j = (*c.hr)[d]; j < (*c.hr)[d + 1]; ++j) a += (*c.h)[(size_t)(*c.hp)[j] * ROWF + k];
            if (a != o[(size_t)d * ROWF + k]) ++bad;
        }
    if (bad) printf("  !! %s: %zu mismatches\n", name, bad);
}

template <typename K>
static void run(const char* name, K kernel, int grid, Ctx& c, int it, bool check) {
    if (check) CK(hipMemset(c.out, 0xff, (size_t)c.N * ROWF * 4));
    const float ms = time_ms(it, [&] { hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, c.src, c.rowptr, c.perm, c.out, c.N); });
    const double bytes = (double)c.E * 512 + (double)c.N * 512 + (double)c.E * 8;   // algorithmic (8-B index per edge)
    printf("%-34s grid=%-6d %8.3f ms  %6.2f TB/s\n", name, grid, ms, bytes / ms / 1e9);
    fflush(stdout);
    if (check) verify(name, c);
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 4000037;   // not a multiple of 256: the tail is exercised
    const int it = argc > 2 ? atoi(argv[2]) : 5;
    const int64_t E = N * 5;
    printf("N=%lld E=%lld src %.2f GB out %.2f GB\n", (long long)N, (long long)E, E * 512 / 1e9, N * 512 / 1e9);
    std::vector<float> h((size_t)E * ROWF);
    std::mt19937_64 rng(42);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((rng() >> 8) & 0xffff) / 65536.f;
    std::vector<int32_t> dst(E), rowptr(N + 1, 0), perm(E);
    for (int64_t e = 0; e < E; ++e) { dst[e] = (int32_t)(rng() % N); ++rowptr[dst[e] + 1]; }
    for (int64_t n = 0; n < N; ++n) rowptr[n + 1] += rowptr[n];
    { std::vector<int32_t> fill(rowptr.begin(), rowptr.end() - 1); for (int64_t e = 0; e < E; ++e) perm[fill[dst[e]]++] = (int32_t)e; }
    float *src, *out; int32_t *d_rowptr, *d_perm;
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&out, (size_t)N * ROWF * 4));
    CK(hipMalloc(&d_rowptr, (N + 1) * 4)); CK(hipMalloc(&d_perm, E * 4));
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rowptr, rowptr.data(), (N + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_perm, perm.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    Ctx c{src, d_rowptr, d_perm, out, N, E, &h, &rowptr, &perm};
    const int chunks = (int)((N + 255) >> 8);

    for (int pass = 0; pass < 2; ++pass) {
        const bool chk = pass == 0;
        printf("--- pass %d\n", pass);
        run("gs nt (seg_rows today)", gs_kernel<ST_NT>, 256 * 16, c, it, chk);
        run("gs plain", gs_kernel<ST_PLAIN>, 256 * 16, c, it, chk);
        run("blk nt (bucket_reduce today)", blk_kernel<ST_NT, 0>, chunks, c, it, chk);
        run("blk nt grid=4096", blk_kernel<ST_NT, 0>, 4096, c, it, chk);
        run("blk plain", blk_kernel<ST_PLAIN, 0>, chunks, c, it, chk);
        run("blk sc1nt", blk_kernel<ST_SC1NT, 0>, chunks, c, it, chk);
        run("run nt", blk_kernel<ST_NT, 1>, chunks, c, it, chk);
        run("burst<2> nt", burst_kernel<ST_NT, 2>, chunks, c, it, chk);
        run("burst<4> nt", burst_kernel<ST_NT, 4>, chunks, c, it, chk);
        run("burst<4> plain", burst_kernel<ST_PLAIN, 4>, chunks, c, it, chk);
        run("burst<4> sc1nt", burst_kernel<ST_SC1NT, 4>, chunks, c, it, chk);
        run("burst<8> nt", burst_kernel<ST_NT, 8>, chunks, c, it, chk);
        run("burst<8> nt grid=4096", burst_kernel<ST_NT, 8>, 4096, c, it, chk);
        run("burst<16> nt", burst_kernel<ST_NT, 16>, chunks, c, it, chk);
        run("tile<16> nt", tile_kernel<ST_NT, 16>, chunks, c, it, chk);
        run("tile<32> nt", tile_kernel<ST_NT, 32>, chunks, c, it, chk);
        run("tile<32> plain", tile_kernel<ST_PLAIN, 32>, chunks, c, it, chk);
        run("tile<32> sc1nt", tile_kernel<ST_SC1NT, 32>, chunks, c, it, chk);
        run("tile<64> nt", tile_kernel<ST_NT, 64>, chunks, c, it, chk);
        run("tile<64> nt grid=4096", tile_kernel<ST_NT, 64>, 4096, c, it, chk);
    }
    return 0;
}
