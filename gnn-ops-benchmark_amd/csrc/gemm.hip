// gemm.hip — torch.addmm(input, mat1, mat2) / torch.matmul(input, other) for 16-bit inputs
// (reference: op_bm_scripts/benchmark_native_addmm.py:13-16, benchmark_native_matmul.py:13-16; fp16 square
// L in [1581, 8164]; BASELINE config 3 asks for bf16). out = input + mat1 @ mat2, fp32 accumulate, one
// rounding. This is the one row of the hot path that is a real contraction, so it runs on the matrix
// cores: v_mfma_f32_16x16x32_{bf16,f16}, 64-lane waves, LDS-staged 128 x 128 x 32 tiles.
//
// Layout (row-major operands, as torch hands them over):
//   A tile [128][32] in LDS with 80-byte rows; a lane's A fragment (row l&15, k = 8*(l>>4)..+7) is one
//     ds_read_b128.
//   B tile [32][128] stays ROW-major in LDS (coalesced 16-B global loads, no transposing writes); the
//     k-strided B fragment comes from two ds_read_b64_tr_b16 hardware-transpose reads (4 k-rows x 16
//     columns per 16-lane group each).
//   4 waves as 2 x 2, each 64 x 64 = 4 x 4 MFMA tiles (64 accumulator VGPRs); next tile's global loads
//     are issued before the MFMA block of the current one.
// First version: single LDS buffer, two barriers per K-step (the guide's "step-3" structure, ~1/3 of
// the MFMA roof); the deeper 256^2 pipeline is the follow-up. Operands whose row length is not a multiple of
// 8 elements (the reference sweeps L = 1581 ... 8164) are first copied into 16-B aligned, zero-padded rows
// (pad_rows_kernel, workspace), so the staging loads are always 16-B vectors.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int AS = 40;   // A row stride in elements (80 B: 16-B aligned, breaks the 64-B power-of-two stride)
constexpr int BS = 136;  // B row stride in elements (272 B)

template <bool IS_BF16>
__device__ inline f32x4 mfma16(const s16x8& a, const s16x8& b, const f32x4& c) {
    if constexpr (IS_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// 8 consecutive 16-bit elements of row `r`, columns c..c+7 of a [rows][cols] row-major matrix (leading dim ld);
// out-of-range elements read as 0. vec_ok: ld % 8 == 0 and 16-B aligned base.
__device__ inline u32x4 load8(const uint16_t* __restrict__ base, int64_t r, int64_t c, int64_t rows, int64_t cols,
                              int64_t ld, bool vec_ok) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r >= rows || c >= cols) return v;
    const uint16_t* p = base + r * ld + c;
    if (vec_ok && c + 8 <= ld) return *reinterpret_cast<const u32x4*>(p);  // ld % 8 == 0; padded tail is zero
    uint16_t e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = (c + i < cols) ? p[i] : (uint16_t)0;
    v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16);
    v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
    return v;
}

template <typename T, bool IS_BF16>
__global__ __launch_bounds__(256) void gemm_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bm,
                                                   const T* __restrict__ addend, T* __restrict__ C, int64_t M,
                                                   int64_t N, int64_t K, int64_t lda, int64_t ldb, bool a_vec,
                                                   bool b_vec) {
    __shared__ __attribute__((aligned(16))) uint16_t sA[BM * AS];
    __shared__ __attribute__((aligned(16))) uint16_t sB[BK * BS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging coordinates: two 16-B pieces of A and of B per thread and K-step
    const int a_row[2] = {tid >> 2, (tid + 256) >> 2};
    const int a_chk = tid & 3;
    const int b_row[2] = {tid >> 4, (tid + 256) >> 4};
    const int b_chk = tid & 15;

    u32x4 ra[2], rb[2];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            ra[p] = load8(A, m0 + a_row[p], k0 + a_chk * 8, M, K, lda, a_vec);
            rb[p] = load8(Bm, k0 + b_row[p], n0 + b_chk * 8, K, N, ldb, b_vec);
        }
    };

    // fragment read addresses (constant over the K loop)
    const int a_off = (wr * 64 + (lane & 15)) * AS + (lane >> 4) * 8;                      // + mi*16*AS
    const int b_off = (8 * (lane >> 4) + ((lane & 15) >> 2)) * BS + wc * 64 + 4 * (lane & 3);  // + ni*16, + 4*BS

    gload(0);
    for (int64_t k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();  // everyone has finished reading the previous tile
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<u32x4*>(&sA[a_row[p] * AS + a_chk * 8]) = ra[p];
            *reinterpret_cast<u32x4*>(&sB[b_row[p] * BS + b_chk * 8]) = rb[p];
        }
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);

        s16x8 af[4], bf[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const s16x8*>(&sA[a_off + mi * 16 * AS]);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(&sB[b_off + ni * 16]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(&sB[b_off + ni * 16 + 4 * BS]));
            bf[ni] = s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16<IS_BF16>(af[mi], bf[ni], acc[mi][ni]);
    }

    // epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wr * 64 + mi * 16 + (lane >> 4) * 4 + r;
                const int64_t col = n0 + wc * 64 + ni * 16 + (lane & 15);
                if (row < M && col < N) {
                    float v = acc[mi][ni][r];
                    if (addend) v += Elem<T>::load(addend + row * N + col);
                    Elem<T>::store(C + row * N + col, v);
                }
            }
}

// Copy a [rows][cols] 16-bit matrix into rows of `ld` elements (ld % 8 == 0), zero-filling the tail, so that every
// row starts 16-B aligned and the GEMM's 16-B staging loads apply (the reference sweeps odd sizes: L = 1581...).
__global__ void pad_rows_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int64_t rows, int64_t cols,
                                int64_t ld) {
    const int64_t chunks = ld / 8, total = rows * chunks;  // one 16-B output chunk per thread
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks, c = (i % chunks) * 8;
        const uint16_t* p = in + r * cols + c;
        uint16_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (c + j < cols) ? p[j] : (uint16_t)0;
        u32x4 v;
        v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16);
        v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
        *reinterpret_cast<u32x4*>(out + r * ld + c) = v;
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int64_t round8(int64_t v) { return (v + 7) / 8 * 8; }

}  // namespace

extern "C" size_t gnnops_addmm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M < 0 || N < 0 || K < 0) return 0;
    size_t b = 0;
    if (K % 8) b += align_up((size_t)M * round8(K) * 2, 256);
    if (N % 8) b += align_up((size_t)K * round8(N) * 2, 256);
    return b;
}

extern "C" int gnnops_addmm(const void* input, const void* mat1, const void* mat2, void* out, int64_t M, int64_t N,
                            int64_t K, int dtype, void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(M >= 0 && N >= 0 && K >= 0, GNNOPS_EINVAL, "addmm: negative size");
    GNNOPS_REQUIRE(dtype == GNNOPS_F16 || dtype == GNNOPS_BF16, GNNOPS_EUNSUPPORTED,
                   "addmm: only float16 / bfloat16 operands are supported (dtype code %d)", dtype);
    if (M * N == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(out && (K == 0 || (mat1 && mat2)), GNNOPS_EINVAL, "addmm: null pointer");
    GNNOPS_REQUIRE(gnnops_cdiv(M, BM) < 65536, GNNOPS_EUNSUPPORTED, "addmm: M too large for the grid");
    const size_t need = gnnops_addmm_workspace_bytes(M, N, K);
    GNNOPS_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), GNNOPS_EWORKSPACE, "addmm: workspace %zu < %zu",
                   workspace_bytes, need);
    int64_t lda = K, ldb = N;
    char* w = (char*)workspace;
    if (K % 8 && K > 0) {
        lda = round8(K);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(gnnops_grid_cap(gnnops_cdiv(M * lda / 8, 256), 256 * 16)), dim3(256), 0, stream,
                           (const uint16_t*)mat1, (uint16_t*)w, M, K, lda);
        mat1 = w;
        w += align_up((size_t)M * lda * 2, 256);
    }
    if (N % 8 && K > 0) {
        ldb = round8(N);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(gnnops_grid_cap(gnnops_cdiv(K * ldb / 8, 256), 256 * 16)), dim3(256), 0, stream,
                           (const uint16_t*)mat2, (uint16_t*)w, K, N, ldb);
        mat2 = w;
    }
    const bool a_vec = (uintptr_t)mat1 % 16 == 0;
    const bool b_vec = (uintptr_t)mat2 % 16 == 0;
    dim3 grid((unsigned)gnnops_cdiv(N, BN), (unsigned)gnnops_cdiv(M, BM));
    if (dtype == GNNOPS_BF16)
        hipLaunchKernelGGL((gemm_kernel<__hip_bfloat16, true>), grid, dim3(256), 0, stream, (const uint16_t*)mat1,
                           (const uint16_t*)mat2, (const __hip_bfloat16*)input, (__hip_bfloat16*)out, M, N, K, lda, ldb, a_vec, b_vec);
    else
        hipLaunchKernelGGL((gemm_kernel<__half, false>), grid, dim3(256), 0, stream, (const uint16_t*)mat1,
                           (const uint16_t*)mat2, (const __half*)input, (__half*)out, M, N, K, lda, ldb, a_vec, b_vec);
    return gnnops_check_launch("addmm");
}
