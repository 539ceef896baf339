// gemm.hip — torch.addmm(input, mat1, mat2) / torch.matmul(input, other) for 16-bit inputs
// (reference: op_bm_scripts/benchmark_native_addmm.py:13-16, benchmark_native_matmul.py:13-16; fp16 square
// L in [1581, 8164]; BASELINE config 3 asks for bf16). out = input + mat1 @ mat2, fp32 accumulate, one
// rounding. This is the one row of the hot path that is a real contraction, so it runs on the matrix
// cores: v_mfma_f32_16x16x32_{bf16,f16}, 64-lane waves, LDS-staged 128 x 128 x 64 tiles.
//
// Layout (row-major operands, as torch hands them over):
//   A tile [128][64] in LDS, 128-B rows, 16-B chunks XOR-swizzled by (row & 7); a lane's A fragment (row l&15,
//     k = 8*(l>>4)..+7) is one conflict-free ds_read_b128.
//   B tile [64][128] stays ROW-major in LDS (coalesced 16-B global loads, no transposing writes) in the
//     guide's swizzled 256-B-row image; the k-strided B fragment comes from two ds_read_b64_tr_b16
//     hardware-transpose reads (4 k-rows x 16 columns per 16-lane group each).
//   4 waves as 2 x 2, each 64 x 64 = 4 x 4 MFMA tiles (64 accumulator VGPRs). The epilogue parks each wave's fp32
//     tile in LDS and writes 16-B row pieces.
// Three kernels (gemm_plan() picks): the register-staged, fully bounds-checked gemm_kernel for small problems; and for
// anything of size the LDS-DMA pipelines gemm_dma4_kernel (128 x 128 tiles) and gemm_dma256_kernel (256 x 256, eight
// waves; gemm_sk256_kernel = the same as persistent workgroups with the last round of tiles cut along K) —
// global_load_lds_dwordx4 into four LDS stages of K = 32 with counted vmcnt waits. The reference sweeps L = 1581 ... 8164:
// rows of any alignment are read where they lie and only the last K-tile comes from zero-filled side copies
// (pad_tail_kernel); large A operands are still copied whole into aligned padded rows (pad_rows_kernel) — gemm_plan().
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_TILE_BYTES = BM * BK * 2;              // 16 KiB, 128-B rows, 8 chunks of 16 B
constexpr int B_TILE_BYTES = BK * BN * 2;              // 16 KiB, 256-B rows, 16 chunks of 16 B
constexpr int STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
constexpr int CS = 68;                                  // epilogue row stride in floats (64 + 4)

template <bool IS_BF16>
__device__ inline f32x4 mfma16(const s16x8& a, const s16x8& b, const f32x4& c) {
    if constexpr (IS_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// XOR swizzles (byte offsets inside a tile). A: chunk ^ (row & 7) makes the ds_read_b128 of 16 rows x one
// k-chunk conflict-free; B: the guide's 256-B-row image (cdna_hip_programming.md T10 (b)), conflict-free for
// ds_read_b64_tr_b16.
__device__ inline int a_off(int row, int ch) { return row * 128 + ((ch ^ (row & 7)) << 4); }
__device__ inline int b_off(int row, int ch) { return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

// Eight consecutive outputs of one row, columns col..col+7: += addend, one rounding, stored as wide as the row
// alignment allows (16 B when N % 8 == 0, 8 B when N % 4 == 0, else per element); rows >= M and columns >= N are dropped.
template <typename T>
__device__ inline void epi_store8(T* __restrict__ C, const T* __restrict__ addend, int64_t row, int64_t col, int64_t M,
                                  int64_t N, float (&f)[8], bool vec_c, bool half_c, int64_t ldadd) {
    if (row >= M || col >= N) return;
    if (vec_c && col + 8 <= N) {
        if (addend) {
            float g[8];
            Elem<T>::unpack(*reinterpret_cast<const u32x4*>(addend + row * ldadd + col), g);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] += g[i];
        }
        *reinterpret_cast<u32x4*>(C + row * N + col) = Elem<T>::pack(f);
    } else if (half_c) {  // rows 8-B aligned (N % 4 == 0): two 4-element pieces, each whole or absent
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (col + 4 * h < N) {
                float* fh = f + 4 * h;
                if (addend) {
                    const uint2 g2 = *reinterpret_cast<const uint2*>(addend + row * ldadd + col + 4 * h);
                    float g[8];
                    Elem<T>::unpack(u32x4{g2.x, g2.y, 0u, 0u}, g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) fh[i] += g[i];
                }
                float tmp[8] = {fh[0], fh[1], fh[2], fh[3], 0.f, 0.f, 0.f, 0.f};
                const u32x4 pk = Elem<T>::pack(tmp);
                *reinterpret_cast<uint2*>(C + row * N + col + 4 * h) = uint2{pk.x, pk.y};
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (col + i < N) {
                float v = f[i];
                if (addend) v += Elem<T>::load(addend + row * ldadd + col + i);
                Elem<T>::store(C + row * N + col + i, v);
            }
        }
    }
}

// 8 consecutive 16-bit elements of row `r`, columns c..c+7 of a [rows][cols] row-major matrix whose rows start
// ALIGN-byte aligned (cols * 2 % ALIGN == 0): ALIGN = 16 -> one dwordx4, 8 -> two dwordx2, 4 -> four dwords, 2 -> elements.
// A piece never straddles the row end; pieces beyond it, and rows beyond `rows`, read as 0.
template <int ALIGN>
__device__ inline u32x4 load8(const uint16_t* __restrict__ base, int64_t r, int64_t c, int64_t rows, int64_t cols) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r >= rows || c >= cols) return v;
    const uint16_t* p = base + r * cols + c;
    if constexpr (ALIGN == 16) {
        v = *reinterpret_cast<const u32x4*>(p);
    } else if constexpr (ALIGN == 8) {
        const uint2 lo = *reinterpret_cast<const uint2*>(p);
        v.x = lo.x; v.y = lo.y;
        if (c + 4 < cols) { const uint2 hi = *reinterpret_cast<const uint2*>(p + 4); v.z = hi.x; v.w = hi.y; }
    } else if constexpr (ALIGN == 4) {
        v.x = *reinterpret_cast<const uint32_t*>(p);
        if (c + 2 < cols) v.y = *reinterpret_cast<const uint32_t*>(p + 2);
        if (c + 4 < cols) v.z = *reinterpret_cast<const uint32_t*>(p + 4);
        if (c + 6 < cols) v.w = *reinterpret_cast<const uint32_t*>(p + 6);
    } else {   // odd row lengths (K = 11 node features): element loads
        uint16_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (c + j < cols) ? p[j] : (uint16_t)0;
        v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16);
        v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
    }
    return v;
}

// out = addend + A @ B. A [M, K] and B [K, N] row-major with ALIGN-byte aligned rows (the host copies an operand
// with an odd row length into padded rows first).
template <typename T, bool IS_BF16, int ALIGN>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bm,
                                                      const T* __restrict__ addend, T* __restrict__ C, int64_t M,
                                                      int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldadd) {
    // lda / ldb = row lengths as stored (== K / N unless the host padded them); they bound the column reads
    constexpr int EPI_BYTES = 4 * 64 * CS * 4;  // four waves' fp32 tiles
    constexpr int SMEM_BYTES = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];  // 68 KiB: two stages; epilogue reuses it

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: 4 chunks of A and 4 of B per thread and K-step; chunk id = tid + 256*p
    int a_st[4], b_st[4];          // swizzled LDS byte offsets (within a stage)
    int64_t a_gr[4], b_gr[4];      // global row of each chunk
    int a_gc[4], b_gc[4];          // global column (elements) inside the tile
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int id = tid + 256 * p;
        const int ar = id >> 3, ac = id & 7;     // A: 128 rows x 8 chunks
        a_st[p] = a_off(ar, ac);
        a_gr[p] = m0 + ar;
        a_gc[p] = ac * 8;
        const int br = id >> 4, bc = id & 15;    // B: 64 rows x 16 chunks
        b_st[p] = A_TILE_BYTES + b_off(br, bc);
        b_gr[p] = br;
        b_gc[p] = bc * 8;
    }
    u32x4 ra[4], rb[4];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p] = load8<ALIGN>(A, a_gr[p], k0 + a_gc[p], M, lda);
            rb[p] = load8<ALIGN>(Bm, k0 + b_gr[p], n0 + b_gc[p], K, ldb);
        }
    };
    auto sstore = [&](int stage) {
        unsigned char* base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<u32x4*>(base + a_st[p]) = ra[p];
            *reinterpret_cast<u32x4*>(base + b_st[p]) = rb[p];
        }
    };

    // fragment addresses. A: row = wr*64 + mi*16 + (lane&15), chunk = ks*4 + (lane>>4).
    // B (transposed read): 16-lane group g = lane>>4 owns k rows ks*32 + 8g (+4 for the second read); lane 4q+p
    // supplies row +q, chunk c0 + (p>>1), +8 bytes for odd p; c0 = 2 * (16-column block) = wc*8 + ni*2.
    const int a_row = wr * 64 + (lane & 15);
    const int a_kc = lane >> 4;
    const int b_q = (lane & 15) >> 2, b_p = lane & 3;
    const int b_row = 8 * (lane >> 4) + b_q;

    const int64_t ksteps = (K + BK - 1) / BK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int64_t kt = 0; kt < ksteps; ++kt) {
        const int cur = (int)(kt & 1);
        const bool more = kt + 1 < ksteps;
        if (more) gload((kt + 1) * BK);
        const unsigned char* sA = smem + cur * STAGE_BYTES;
        const unsigned char* sB = sA + A_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x8 af[4], bf[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                af[mi] = *reinterpret_cast<const s16x8*>(sA + a_off(a_row + mi * 16, ks * 4 + a_kc));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int ch = wc * 8 + ni * 2 + (b_p >> 1);
                const int r_lo = ks * 32 + b_row;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sB + b_off(r_lo, ch) + 8 * (b_p & 1)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sB + b_off(r_lo + 4, ch) + 8 * (b_p & 1)));
                bf[ni] = s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16<IS_BF16>(af[mi], bf[ni], acc[mi][ni]);
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
    }

    // Epilogue through LDS: each wave parks its 64 x 64 fp32 tile (C/D map: col = lane & 15, row = (lane >> 4)*4 + reg),
    // then reads whole row pieces back: 8 lanes x 8 columns per row, so the addend load and the store are 16-B
    // accesses on 128-B row segments. One rounding, after the add.
    float* ctile = reinterpret_cast<float*>(smem) + wave * (64 * CS);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ctile[(mi * 16 + (lane >> 4) * 4 + r) * CS + ni * 16 + (lane & 15)] = acc[mi][ni][r];
    __builtin_amdgcn_wave_barrier();  // the tile is private to this wave; LDS ops of one wave complete in order
    const bool vec_c = (N % 8 == 0) && ((uintptr_t)C % 16 == 0) && (addend == nullptr || (uintptr_t)addend % 16 == 0);
    const bool half_c = (N % 4 == 0) && ((uintptr_t)C % 8 == 0) && (addend == nullptr || (uintptr_t)addend % 8 == 0);
    const int pr = lane >> 3, pc = (lane & 7) * 8;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int rr = pass * 8 + pr;
        const int64_t row = m0 + wr * 64 + rr;
        const int64_t col = n0 + wc * 64 + pc;
        float f[8];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(&ctile[rr * CS + pc]);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(&ctile[rr * CS + pc + 4]);
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        epi_store8<T>(C, addend, row, col, M, N, f, vec_c, half_c, ldadd);
    }
}

// LDS-DMA pipeline: BK = 32 and FOUR LDS stages, staged by global_load_lds_dwordx4 (1 KiB per wave-instruction written
// straight into LDS, the swizzle applied on the per-lane SOURCE address: no staging VGPRs, no ds_write pass), the DMA of
// tile t+3 issued while tile t is multiplied. With two stages and `__syncthreads()` (which waits vmcnt(0)) a K-step
// cannot start before the DMA issued one step earlier has landed, and L2/HBM latency is of the order of a K-step's MFMA
// time (measured: 860 TFLOP/s at 8192^3 bf16, against 1000 here). The wait is COUNTED — `s_waitcnt vmcnt(8)` retires
// tile t and leaves the 4 + 4 DMA instructions of tiles t+1 and t+2 in flight across a raw `s_barrier`
// (cdna_hip_programming.md §5 "Pipelining across barriers").
// A tile [128][32]: 64-B rows, chunk position c' holds chunk c' ^ a4_swz(row) (conflict-free ds_read_b128); B tile
// [32][128]: the 256-B-row image of b_off(). One barrier per K-step; stage (t+3) % 4 is the one every wave finished
// reading before that barrier.
constexpr int BK4 = 32, NST = 4;
constexpr int A4_BYTES = BM * BK4 * 2, B4_BYTES = BK4 * BN * 2, STAGE4_BYTES = A4_BYTES + B4_BYTES;  // 8 + 8 KiB

// ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): with lane = row | chunk << 4 a
// group holds chunk c of rows 0-3 and 12-15 and chunk c^1 of rows 4-11, so the four rows that share a 64-B quarter of a
// bank row (r, r+4, r+8, r+12) need positions c^f, (c^1)^f', ...: f = 3 for rows 8-15 of every 16 makes them distinct
// (tools/lds_conflicts.py: 4 LDS cycles per read, the conflict-free figure).
__device__ inline int a4_swz(int row) { return ((row >> 3) & 1) * 3; }
__device__ inline int a4_off(int row, int ch) { return row * 64 + ((ch ^ a4_swz(row)) << 4); }

// The LDS-DMA kernels read operand rows where they lie, whatever their alignment (global_load_lds_dwordx4 takes any
// 2-byte-aligned source: tools/micro/dma_unaligned.hip). What they cannot read in place is the END of K: the last K-tile
// must hold zeros past K on both operands (0 x garbage is not 0 when the garbage is Inf or NaN) and a B row's last
// 16-B piece may run past the matrix. So the host copies only the last K-tile — A[:, Kmain:K] into At [M][64] and
// B[Kmain:K, :] into Bt [64][ldb], zero-filled — and K-steps at or past Kmain take their pieces from there: the same
// per-lane pointer plus a constant (element offsets; Kmain = K when there is no side copy).
// An operand that IS a whole padded copy has no side copy (At / Bt null): its delta is 0.
__device__ inline int64_t tail_delta_a(const uint16_t* A, const uint16_t* At, int64_t arow, int64_t lda, int64_t Kmain) {
    return At ? (int64_t)(((intptr_t)At - (intptr_t)A) / 2) + arow * (64 - lda) - Kmain : 0;
}
__device__ inline int64_t tail_delta_b(const uint16_t* Bm, const uint16_t* Bt, int64_t ldb, int64_t Kmain) {
    return Bt ? (int64_t)(((intptr_t)Bt - (intptr_t)Bm) / 2) - Kmain * ldb : 0;
}

template <typename T, bool IS_BF16>
__global__ __launch_bounds__(256, 2) void gemm_dma4_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bm,
                                                           const T* __restrict__ addend, T* __restrict__ C, int64_t M,
                                                           int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldadd,
                                                           const uint16_t* __restrict__ At, const uint16_t* __restrict__ Bt,
                                                           int64_t Kmain) {
    constexpr int EPI_BYTES = 4 * 64 * CS * 4;
    constexpr int SMEM_BYTES = (NST * STAGE4_BYTES > EPI_BYTES) ? NST * STAGE4_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's DMA pieces per stage: A pieces wave*2, wave*2+1 (16 rows x 64 B), B pieces wave*2, wave*2+1 (4 rows x 256 B)
    const uint16_t* a_src[2];
    const uint16_t* b_src[2];
    int64_t a_dt[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        // rows past M re-read row M-1 and column chunks past ldb re-read the last chunk: in-bounds filler for outputs
        // the epilogue drops (the host pads K, never M or N)
        const int ar = (wave * 2 + p) * 16 + (lane >> 2);
        const int64_t arow = (m0 + ar < M) ? m0 + ar : M - 1;
        a_src[p] = A + arow * lda + (((lane & 3) ^ a4_swz(ar)) << 3);
        a_dt[p] = tail_delta_a(A, At, arow, lda, Kmain);
        const int br = (wave * 2 + p) * 4 + (lane >> 4);
        int64_t bcol = n0 + (((lane & 15) ^ (((br & 3) << 2) | ((br >> 2) & 3))) << 3);
        if (bcol > ((N - 1) & ~(int64_t)7)) bcol = (N - 1) & ~(int64_t)7;   // chunks past N re-read the last one that holds a column
        b_src[p] = Bm + (int64_t)br * ldb + bcol;
    }
    const int64_t b_dt = tail_delta_b(Bm, Bt, ldb, Kmain);
    auto dma = [&](int stage, int64_t k0) {
        unsigned char* base = smem + stage * STAGE4_BYTES;
        const bool in_tail = k0 >= Kmain;   // uniform: the last K-tile comes from the zero-padded side copies
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[p] + k0 + (in_tail ? a_dt[p] : 0)),
                                             (__attribute__((address_space(3))) void*)(base + (wave * 2 + p) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[p] + k0 * ldb + (in_tail ? b_dt : 0)),
                                             (__attribute__((address_space(3))) void*)(base + A4_BYTES + (wave * 2 + p) * 1024),
                                             16, 0, 0);
        }
    };

    const int a_row = wr * 64 + (lane & 15);
    const int a_kc = lane >> 4;
    const int b_q = (lane & 15) >> 2, b_p = lane & 3;
    const int b_row = 8 * (lane >> 4) + b_q;

    const int64_t ksteps = K / BK4;  // even: K is a multiple of 64
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)smem;

    // Fragments of tile t+1 are read into the second register set while the MFMAs of tile t issue. The transpose reads
    // go through inline asm: as a builtin the compiler orders them after EVERY outstanding LDS-DMA (s_waitcnt
    // vmcnt(0)), which would undo the counted wait. Their lgkmcnt(0) carries the registers as operands so no MFMA is
    // scheduled above it.
    auto read_frags = [&](int64_t t, s16x8 (&af)[4], s16x4 (&blo)[4], s16x4 (&bhi)[4]) {
        const int st = (int)(t & (NST - 1));
        const uint32_t sA_lds = smem_lds + st * STAGE4_BYTES + a4_off(a_row, a_kc);  // rows +16 keep the swizzle term
        const uint32_t sB_lds = smem_lds + st * STAGE4_BYTES + A4_BYTES;
        asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(sA_lds));
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(af[1]) : "v"(sA_lds));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[2]) : "v"(sA_lds));
        asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(af[3]) : "v"(sA_lds));
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int ch = wc * 8 + ni * 2 + (b_p >> 1);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo[ni]) : "v"(sB_lds + b_off(b_row, ch) + 8 * (b_p & 1)));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bhi[ni]) : "v"(sB_lds + b_off(b_row + 4, ch) + 8 * (b_p & 1)));
        }
    };
    // one K-step: `cur` holds tile kt (reads issued one step earlier), `nxt` receives tile kt+1
    auto step = [&](int64_t kt, s16x8 (&caf)[4], s16x4 (&cblo)[4], s16x4 (&cbhi)[4], s16x8 (&naf)[4], s16x4 (&nblo)[4],
                    s16x4 (&nbhi)[4]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(cblo[0]), "+v"(cblo[1]), "+v"(cblo[2]), "+v"(cblo[3]), "+v"(cbhi[0]), "+v"(cbhi[1]),
                       "+v"(cbhi[2]), "+v"(cbhi[3]), "+v"(caf[0]), "+v"(caf[1]), "+v"(caf[2]), "+v"(caf[3]));
        // tile kt+1 must have landed; tiles kt+2 and kt+3 (4 DMA instructions each) stay in flight. Past the last tile
        // the DMA re-fetches tile ksteps-1 into a stage nobody reads again, which keeps this count (and the loop)
        // free of branches.
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave holds tile kt in registers: its stage is free
        const int64_t tn = kt + NST;
        dma((int)(kt & (NST - 1)), (tn < ksteps ? tn : ksteps - 1) * BK4);
        read_frags(kt + 1, naf, nblo, nbhi);  // past the end: a stale stage, never used
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const s16x8 bf = s16x8{cblo[ni].x, cblo[ni].y, cblo[ni].z, cblo[ni].w,
                                       cbhi[ni].x, cbhi[ni].y, cbhi[ni].z, cbhi[ni].w};
                acc[mi][ni] = mfma16<IS_BF16>(caf[mi], bf, acc[mi][ni]);
            }
        __builtin_amdgcn_sched_barrier(0);
    };

#pragma unroll
    for (int t = 0; t < NST; ++t) dma(t, (int64_t)(t < ksteps ? t : ksteps - 1) * BK4);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    s16x8 af0[4], af1[4];
    s16x4 bl0[4], bh0[4], bl1[4], bh1[4];
    read_frags(0, af0, bl0, bh0);
    for (int64_t kt = 0; kt < ksteps; kt += 2) {
        step(kt, af0, bl0, bh0, af1, bl1, bh1);
        step(kt + 1, af1, bl1, bh1, af0, bl0, bh0);
    }
    // the reads issued for the tile past the end still target live registers: retire them before anything is reused
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(bl0[0]), "+v"(bl0[1]), "+v"(bl0[2]), "+v"(bl0[3]), "+v"(bh0[0]), "+v"(bh0[1]), "+v"(bh0[2]),
                   "+v"(bh0[3]), "+v"(af0[0]), "+v"(af0[1]), "+v"(af0[2]), "+v"(af0[3])
                 :
                 : "memory");
    __syncthreads();  // every wave is done with the stages before the epilogue reuses the memory

    float* ctile = reinterpret_cast<float*>(smem) + wave * (64 * CS);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ctile[(mi * 16 + (lane >> 4) * 4 + r) * CS + ni * 16 + (lane & 15)] = acc[mi][ni][r];
    __builtin_amdgcn_wave_barrier();
    const bool vec_c = (N % 8 == 0) && ((uintptr_t)C % 16 == 0) && (addend == nullptr || (uintptr_t)addend % 16 == 0);
    const bool half_c = (N % 4 == 0) && ((uintptr_t)C % 8 == 0) && (addend == nullptr || (uintptr_t)addend % 8 == 0);
    const int pr = lane >> 3, pc = (lane & 7) * 8;
    // interior tile with an addend: its 8 pieces per lane are requested together (one round trip instead of one per pass)
    const bool pre_ok = vec_c && addend != nullptr && m0 + BM <= M && n0 + BN <= N;
    u32x4 pre[8];
    if (pre_ok) {
        const T* ap = addend + (m0 + wr * 64 + pr) * ldadd + n0 + wc * 64 + pc;
#pragma unroll
        for (int i = 0; i < 8; ++i) pre[i] = *reinterpret_cast<const u32x4*>(ap + (int64_t)i * 8 * ldadd);
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int rr = pass * 8 + pr;
        float f[8];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(&ctile[rr * CS + pc]);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(&ctile[rr * CS + pc + 4]);
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        const int64_t row = m0 + wr * 64 + rr, col = n0 + wc * 64 + pc;
        if (pre_ok) {
            float g[8];
            Elem<T>::unpack(pre[pass], g);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] += g[i];
            *reinterpret_cast<u32x4*>(C + row * N + col) = Elem<T>::pack(f);
        } else {
            epi_store8<T>(C, addend, row, col, M, N, f, vec_c, half_c, ldadd);
        }
    }
}

// 256 x 256 block tile, eight waves of 128 x 64 — for problems with at least 128 such tiles (half the CUs busy with
// this kernel already beat 128 x 128 tiles on all of them: gemm_plan). At 128 x 128 / 64 x 64
// per wave the LDS array is as busy as the MFMA pipe (per K-step of 32 and per CU: 8 waves x 8 KiB of fragment reads at
// 256 B/clk plus 32 KiB of DMA stores against 2 x 256 MFMA cycles per SIMD); the larger wave tile reads 12 KiB per 32
// MFMAs instead of 8 KiB per 16 and halves the L2 -> LDS bytes per flop. Same four-stage counted-vmcnt pipeline as
// gemm_dma4_kernel; stage = A [256][32] (a4_off image) + B [32][256] as two [32][128] half images (b_off).
// VAR 2 (default): the two wave groups ping-pong (see the main loop); VAR 1 (GNNOPS_GEMM_VAR=1, kept for A/B runs with
// tools/time_gemm_var.py): all eight waves in lockstep with a half-step software pipeline.
constexpr int BM2 = 256, BN2 = 256;
constexpr int A2_BYTES = BM2 * BK4 * 2, B2_BYTES = BK4 * BN2 * 2, STAGE2_BYTES = A2_BYTES + B2_BYTES;  // 16 + 16 KiB
constexpr int GEMM256_SMEM = NST * STAGE2_BYTES;                                                       // 128 KiB
constexpr int EPI2_ROWS = 32;  // rows of a wave's 128 staged per epilogue round: 8 waves x 32 x CS x 4 B = 68 KiB

template <typename T, bool IS_BF16, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_dma256_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bm,
                                                             const T* __restrict__ addend, T* __restrict__ C, int64_t M,
                                                             int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldadd,
                                                             const uint16_t* __restrict__ At, const uint16_t* __restrict__ Bt,
                                                             int64_t Kmain) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem256[];
    unsigned char* smem = smem256;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;  // 2 x 4 waves: rows wr*128, columns wc*64
    const int64_t m0 = (int64_t)blockIdx.y * BM2, n0 = (int64_t)blockIdx.x * BN2;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DMA pieces of this wave per stage: A pieces 2w, 2w+1 (16 rows x 64 B); B pieces 2w, 2w+1 of 16 (half q>>3, 4 rows x 256 B)
    const uint16_t* a_src[2];
    const uint16_t* b_src[2];
    int64_t a_dt[2];
    int b_dst[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int ar = (wave * 2 + p) * 16 + (lane >> 2);
        const int64_t arow = (m0 + ar < M) ? m0 + ar : M - 1;  // in-bounds filler, as in gemm_dma4_kernel
        a_src[p] = A + arow * lda + (((lane & 3) ^ a4_swz(ar)) << 3);
        a_dt[p] = tail_delta_a(A, At, arow, lda, Kmain);
        const int q = wave * 2 + p, half = q >> 3;
        const int br = (q & 7) * 4 + (lane >> 4);
        int64_t bcol = n0 + half * 128 + (((lane & 15) ^ (((br & 3) << 2) | ((br >> 2) & 3))) << 3);
        if (bcol > ((N - 1) & ~(int64_t)7)) bcol = (N - 1) & ~(int64_t)7;
        b_src[p] = Bm + (int64_t)br * ldb + bcol;
        b_dst[p] = A2_BYTES + half * (B2_BYTES / 2) + (q & 7) * 1024;
    }
    const int64_t b_dt = tail_delta_b(Bm, Bt, ldb, Kmain);
    auto dma_a = [&](int stage, int64_t k0, int p) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[p] + k0 + (k0 >= Kmain ? a_dt[p] : 0)),
                                         (__attribute__((address_space(3))) void*)(smem + stage * STAGE2_BYTES + (wave * 2 + p) * 1024),
                                         16, 0, 0);
    };
    auto dma_b = [&](int stage, int64_t k0, int p) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[p] + k0 * ldb + (k0 >= Kmain ? b_dt : 0)),
                                         (__attribute__((address_space(3))) void*)(smem + stage * STAGE2_BYTES + b_dst[p]), 16, 0, 0);
    };
    auto dma = [&](int stage, int64_t k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            dma_a(stage, k0, p);
            dma_b(stage, k0, p);
        }
    };

    const int a_row = wr * 128 + (lane & 15);
    const int a_kc = lane >> 4;
    const int b_q = (lane & 15) >> 2, b_p = lane & 3;
    const int b_row = 8 * (lane >> 4) + b_q;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)smem;
    const uint32_t a_rd = smem_lds + a4_off(a_row, a_kc);                      // + stage, + mi * 1024
    const uint32_t b_rd = smem_lds + A2_BYTES + (wc >> 1) * (B2_BYTES / 2);    // + stage, + b_off(...)
    uint32_t b_lo_off[4], b_hi_off[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int ch = (wc & 1) * 8 + ni * 2 + (b_p >> 1);
        b_lo_off[ni] = b_rd + b_off(b_row, ch) + 8 * (b_p & 1);
        b_hi_off[ni] = b_rd + b_off(b_row + 4, ch) + 8 * (b_p & 1);
    }

    const int64_t ksteps = K / BK4;  // even: K is a multiple of 64
    // Fragment reads are inline asm: as builtins the transpose reads are ordered after EVERY outstanding LDS-DMA
    // (s_waitcnt vmcnt(0)), which would undo the counted waits. Each s_waitcnt carries the registers it releases as
    // operands, so no MFMA that uses them is scheduled above it.
    auto read_first = [&](int64_t t, s16x4 (&blo)[4], s16x4 (&bhi)[4], s16x8 (&af)[8]) {  // B and rows 0..63 of A
        const uint32_t st = (uint32_t)(t & (NST - 1)) * STAGE2_BYTES;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo[ni]) : "v"(b_lo_off[ni] + st));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bhi[ni]) : "v"(b_hi_off[ni] + st));
        }
        const uint32_t aa = a_rd + st;
        asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(af[1]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[2]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(af[3]) : "v"(aa));
    };
    auto read_second = [&](int64_t t, s16x8 (&af)[8]) {  // rows 64..127 of A
        const uint32_t aa = a_rd + (uint32_t)(t & (NST - 1)) * STAGE2_BYTES;
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[4]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(af[5]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(af[6]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(af[7]) : "v"(aa));
    };
    s16x8 af[8];
    // One K-step, software-pipelined in halves so the LDS reads run under this wave's own MFMAs:
    //   rows 64..127 of tile kt are read under the MFMAs of rows 0..63; B and rows 0..63 of tile kt+1 under the MFMAs of
    //   rows 64..127. B is double-buffered in registers (cur / nxt), A reuses af[0..3] once their MFMAs have issued.
    auto step = [&](int64_t kt, s16x4 (&cblo)[4], s16x4 (&cbhi)[4], s16x4 (&nblo)[4], s16x4 (&nbhi)[4]) {
        read_second(kt, af);
        asm volatile("s_waitcnt lgkmcnt(4)"  // LDS returns in order: everything but the four reads just issued
                     : "+v"(cblo[0]), "+v"(cblo[1]), "+v"(cblo[2]), "+v"(cblo[3]), "+v"(cbhi[0]), "+v"(cbhi[1]),
                       "+v"(cbhi[2]), "+v"(cbhi[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]));
        s16x8 bf[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            bf[ni] = s16x8{cblo[ni].x, cblo[ni].y, cblo[ni].z, cblo[ni].w, cbhi[ni].x, cbhi[ni].y, cbhi[ni].z, cbhi[ni].w};
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16<IS_BF16>(af[mi], bf[ni], acc[mi][ni]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]));
        // tile kt+1 must have landed, tile kt+2 (4 DMA instructions) stays in flight; after the barrier every wave is
        // done with tile kt-1's stage, which takes tile kt+3. Past the last tile the DMA re-fetches tile ksteps-1 into a
        // stage nobody reads again: the count and the loop stay free of branches.
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int64_t tn = kt + NST - 1;
        const int stn = (int)(tn & (NST - 1));
        const int64_t k0n = (tn < ksteps ? tn : ksteps - 1) * BK4;
        read_first(kt + 1, nblo, nbhi, af);  // past the end: a stale stage, never used
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 4; mi < 8; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16<IS_BF16>(af[mi], bf[ni], acc[mi][ni]);
            // one DMA piece behind every four MFMAs (issued right after the barrier they cost 3 % more)
            if (mi & 1) dma_b(stn, k0n, (mi - 4) >> 1);
            else dma_a(stn, k0n, (mi - 4) >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (VAR == 2) {
        // Ping-pong (the default): waves 0-3 and 4-7 — wave w and w + 4 share a SIMD — run half a K-step apart, so one
        // partner's memory phase (16 fragment reads, 4 DMA pieces at ~100 cycles of issue each, their latency) runs under
        // the other's 32 back-to-back MFMAs. Run in lockstep (VAR 1) both partners are in that phase together and the
        // matrix pipe idles: 0.88 -> 0.79 ms at 8192^3 bf16. Still ONE barrier per K-step kt:
        //   waves 0-3 reach it after the MFMAs of tile kt, waves 4-7 after reading tile kt's fragments (reads retired);
        //   every wave has waited for its own DMA pieces of tile kt+1 (vmcnt(8): tiles kt+2, kt+3 stay in flight);
        //   behind it waves 0-3 read tile kt+1 and refill tile kt's stage with tile kt+4, waves 4-7 multiply tile kt,
        //   then read tile kt+1 and refill tile kt's stage too (their pieces of tile kt+4 = (kt+1)+3).
        // Both groups execute 1 + ksteps barriers. Fragments are single-buffered: a wave's MFMAs have all issued before
        // its next reads are.
        s16x4 blo[4], bhi[4];
        auto read_all = [&](int64_t t) {
            const uint32_t st = (uint32_t)(t & (NST - 1)) * STAGE2_BYTES;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo[ni]) : "v"(b_lo_off[ni] + st));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bhi[ni]) : "v"(b_hi_off[ni] + st));
            }
            const uint32_t aa = a_rd + st;
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(af[1]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[2]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(af[3]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[4]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(af[5]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(af[6]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(af[7]) : "v"(aa));
        };
        auto wait_reads = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(blo[0]), "+v"(blo[1]), "+v"(blo[2]), "+v"(blo[3]), "+v"(bhi[0]), "+v"(bhi[1]), "+v"(bhi[2]),
                           "+v"(bhi[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]),
                           "+v"(af[6]), "+v"(af[7]));
        };
        auto compute = [&]() {
            s16x8 bf[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                bf[ni] = s16x8{blo[ni].x, blo[ni].y, blo[ni].z, blo[ni].w, bhi[ni].x, bhi[ni].y, bhi[ni].z, bhi[ni].w};
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16<IS_BF16>(af[mi], bf[ni], acc[mi][ni]);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto dma_tile = [&](int64_t tn) { dma((int)(tn & (NST - 1)), (tn < ksteps ? tn : ksteps - 1) * BK4); };
        dma_tile(0);
        dma_tile(1);
        dma_tile(2);
        if (__builtin_amdgcn_readfirstlane(wave) < 4) {
            dma_tile(3);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // own pieces of tile 0
            __builtin_amdgcn_s_barrier();
            read_all(0);
            for (int64_t kt = 0; kt < ksteps; ++kt) {
                wait_reads();
                compute();
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // own pieces of tile kt+1; kt+2 and kt+3 stay in flight
                __builtin_amdgcn_s_barrier();
                read_all(kt + 1);   // past the end: a stale stage, never used
                dma_tile(kt + 4);   // tile kt's stage: waves 4-7 retired their reads of it before the barrier
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int64_t kt = 0; kt < ksteps; ++kt) {
                read_all(kt);
                dma_tile(kt + 3);   // tile kt-1's stage: everyone read it before the previous barrier
                wait_reads();
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                compute();
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(blo[0]), "+v"(blo[1]), "+v"(blo[2]), "+v"(blo[3]), "+v"(bhi[0]), "+v"(bhi[1]), "+v"(bhi[2]),
                       "+v"(bhi[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]),
                       "+v"(af[6]), "+v"(af[7])
                     :
                     : "memory");
    } else {
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) dma(t, (int64_t)(t < ksteps ? t : ksteps - 1) * BK4);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    s16x4 bl0[4], bh0[4], bl1[4], bh1[4];
    read_first(0, bl0, bh0, af);
    for (int64_t kt = 0; kt < ksteps; kt += 2) {
        step(kt, bl0, bh0, bl1, bh1);
        step(kt + 1, bl1, bh1, bl0, bh0);
    }
    // the reads issued for the tile past the end still target live registers: retire them before anything is reused
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(bl0[0]), "+v"(bl0[1]), "+v"(bl0[2]), "+v"(bl0[3]), "+v"(bh0[0]), "+v"(bh0[1]), "+v"(bh0[2]),
                   "+v"(bh0[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])
                 :
                 : "memory");
    }
    __syncthreads();  // all DMA (including the redundant tail fetches) landed, all reads done: the stages become the epilogue's

    float* ctile = reinterpret_cast<float*>(smem) + wave * (EPI2_ROWS * CS);
    const bool vec_c = (N % 8 == 0) && ((uintptr_t)C % 16 == 0) && (addend == nullptr || (uintptr_t)addend % 16 == 0);
    const bool half_c = (N % 4 == 0) && ((uintptr_t)C % 8 == 0) && (addend == nullptr || (uintptr_t)addend % 8 == 0);
    const int pr = lane >> 3, pc = (lane & 7) * 8;
    // Interior tile with an addend: all 16 addend pieces of this lane (one per staged row it stores) are requested here, in
    // one round trip under the staging below, instead of one dependent global load per store pass (16 round trips).
    const bool pre_ok = vec_c && addend != nullptr && m0 + BM2 <= M && n0 + BN2 <= N;
    u32x4 pre[16];
    if (pre_ok) {
        const T* ap = addend + (m0 + wr * 128 + pr) * ldadd + n0 + wc * 64 + pc;
#pragma unroll
        for (int i = 0; i < 16; ++i) pre[i] = *reinterpret_cast<const u32x4*>(ap + (int64_t)i * 8 * ldadd);
    }
#pragma unroll
    for (int c = 0; c < 128 / EPI2_ROWS; ++c) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ctile[(h * 16 + (lane >> 4) * 4 + r) * CS + ni * 16 + (lane & 15)] = acc[c * 2 + h][ni][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 0; pass < EPI2_ROWS / 8; ++pass) {
            const int rr = pass * 8 + pr;
            float f[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(&ctile[rr * CS + pc]);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(&ctile[rr * CS + pc + 4]);
            f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
            const int64_t row = m0 + wr * 128 + c * EPI2_ROWS + rr, col = n0 + wc * 64 + pc;
            if (pre_ok) {
                float g[8];
                Elem<T>::unpack(pre[c * 4 + pass], g);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] += g[i];
                *reinterpret_cast<u32x4*>(C + row * N + col) = Elem<T>::pack(f);
            } else {
                epi_store8<T>(C, addend, row, col, M, N, f, vec_c, half_c, ldadd);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T, bool IS_BF16, int VAR>
int launch_dma256_var(const void* input, const void* mat1, const void* mat2, void* out, int64_t M, int64_t N, int64_t K,
                  int64_t lda, int64_t ldb, int64_t ldadd, const uint16_t* at, const uint16_t* bt, int64_t Kmain, hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dma256_kernel<T, IS_BF16, VAR>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, GEMM256_SMEM) != hipSuccess)
            return gnnops_check_launch("addmm attribute");
        configured = true;
    }
    hipLaunchKernelGGL((gemm_dma256_kernel<T, IS_BF16, VAR>), dim3((unsigned)gnnops_cdiv(N, BN2), (unsigned)gnnops_cdiv(M, BM2)), dim3(512),
                       GEMM256_SMEM, stream, (const uint16_t*)mat1, (const uint16_t*)mat2, (const T*)input, (T*)out, M, N, K,
                       lda, ldb, ldadd, at, bt, Kmain);
    return gnnops_check_launch("addmm");
}

template <typename T, bool IS_BF16>
int launch_dma256(const void* input, const void* mat1, const void* mat2, void* out, int64_t M, int64_t N, int64_t K,
                  int64_t lda, int64_t ldb, int64_t ldadd, const uint16_t* at, const uint16_t* bt, int64_t Kmain, hipStream_t stream) {
    const char* var = getenv("GNNOPS_GEMM_VAR");
    if (var && var[0] == '1') return launch_dma256_var<T, IS_BF16, 1>(input, mat1, mat2, out, M, N, K, lda, ldb, ldadd, at, bt, Kmain, stream);
    return launch_dma256_var<T, IS_BF16, 2>(input, mat1, mat2, out, M, N, K, lda, ldb, ldadd, at, bt, Kmain, stream);
}

// ---- split-K tail over the same 256 x 256 tiles: ONE persistent workgroup per CU --------------------------------------
// A grid of T tiles on G CUs runs in ceil(T / G) rounds whatever T is: the reference's sweep (benchmark_native_addmm.py:23-27,
// L = 1581 ... 8164, whatever int(sqrt(x)) gives) makes 17 x 17 = 289 tiles cost what 19 x 19 = 361 cost (two rounds, the
// second with 33 tiles: measured 248.6 against 244.8 us). Here the whole rounds run as before (workgroup v takes tiles v,
// v + G, ... — every workgroup of an XCD at the same K-step of neighbouring tiles, which is what lets one L2 serve 12
// operand panels to 32 tiles), and the r = T mod G tiles of the last round are cut along K into s = 8, 4 or 2 pieces
// (r <= 32, 64, 128), one workgroup each. Classic stream-K (one sequence of K-steps cut into G equal ranges) was built
// first and measured: ranges start at arbitrary K offsets, no two workgroups of an XCD read the same panel rows at the
// same time, every operand byte crosses the fabric once per tile and the main loop runs at HALF speed
// (profiles/round3_f_gemm_streamk.txt) — the pieces must be in phase. So: XCD x multiplies piece x mod s of the tail tiles
// (8 / s interleaved subsets of them): all its workgroups walk the same K range of neighbouring tiles.
// The s workgroups of a tile are s CONSECUTIVE block ids (they sit on s different XCDs). Each leaves its fp32
// accumulators in its own 256 KiB slot of the workspace (register order: 32 x 16 B per lane, coalesced), raises its
// flag, waits for its s - 1 partners' flags and finishes 1 / s of the tile: the 16-row blocks [j 8 / s, (j + 1) 8 / s) of
// every wave — its own registers plus that part of the partners' slots — through the ordinary epilogue. Every workgroup
// has written everything others wait for BEFORE it waits itself, and a tile's workgroups are adjacent in dispatch order:
// with in-order dispatch any eight free CUs are enough for progress, whatever else is running on the chip.
// Hand-off as MI355X_MICROARCH.md "Valid forms" prescribes: plain stores, every wave's vmcnt(0), workgroup barrier, lane 0
// agent release + vmcnt(0) + relaxed agent flag store | relaxed polls, one agent acquire, vmcnt(0), barrier, plain loads.
// Tile ids run down bands of eight tile-rows (column after column inside a band) and, in the whole rounds, workgroup
// v = the v-th of its XCD's contiguous share: the 32 workgroups of an XCD sit on an 8 x 4 block of tiles.
constexpr int SK_SLOT_FLOATS = BM2 * BN2;  // 256 KiB of fp32 per workgroup
constexpr unsigned SK_SPIN_LIMIT = 1u << 25;  // polls of ~0.1 us each before an owner gives up and poisons its tile

__device__ inline void sk_tile_rc(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int band = id / (8 * tiles_n);
    const int rows = (tiles_m - band * 8 < 8) ? tiles_m - band * 8 : 8;
    const int local = id - band * 8 * tiles_n;
    tn = local / rows;
    tm = band * 8 + local % rows;
}

template <typename T, bool IS_BF16>
__global__ __launch_bounds__(512, 2) void gemm_sk256_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bm,
                                                            const T* __restrict__ addend, T* __restrict__ C, int64_t M,
                                                            int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldadd,
                                                            const uint16_t* __restrict__ At, const uint16_t* __restrict__ Bt,
                                                            int64_t Kmain, float* __restrict__ slots,
                                                            unsigned* __restrict__ flags, int tiles_m, int tiles_n, int dp_tiles,
                                                            int tail_tiles, int split, int order) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem256[];
    unsigned char* smem = smem256;
    __shared__ int poisoned;

    const int G = (int)gridDim.x;
    const int v = ((order & 1) && (G & 7) == 0) ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int64_t S = K / BK4;
    int dp_next = v;
    // tail: block b = 8 i + x (XCD x): piece x mod split of tail tile i (8 / split) + x / split
    const int tail_j = (int)blockIdx.x & (split - 1);
    const int tail_t = ((int)blockIdx.x >> 3) * (8 / split) + (((int)blockIdx.x & 7) / split);
    bool tail_left = split > 1 && tail_t < tail_tiles && !(order & 8);   // order bits 4, 8: timing-only builds of the A/B tool
    if (threadIdx.x == 0) poisoned = 0;
    const bool vec_c = (N % 8 == 0) && ((uintptr_t)C % 16 == 0) && (addend == nullptr || (uintptr_t)addend % 16 == 0);
    const bool half_c = (N % 4 == 0) && ((uintptr_t)C % 8 == 0) && (addend == nullptr || (uintptr_t)addend % 8 == 0);

    for (;;) {
        int t;
        int64_t kb, ke;
        bool is_tail = false;
        if (dp_next < dp_tiles) {
            t = dp_next;
            dp_next += G;
            kb = 0;
            ke = S;
        } else if (tail_left) {
            tail_left = false;
            is_tail = true;
            t = dp_tiles + tail_t;
            kb = S * tail_j / split;
            ke = S * (tail_j + 1) / split;
        } else {
            break;
        }
        int tm, tn;
        if (order & 2) sk_tile_rc(t, tiles_m, tiles_n, tm, tn);
        else { tm = t / tiles_n; tn = t - tm * tiles_n; }
        const int64_t m0 = (int64_t)tm * BM2, n0 = (int64_t)tn * BN2;
        const int64_t ksteps = ke - kb;
        // per-lane constants are rebuilt for every piece from a value the compiler cannot see through: hoisted out of the
        // persistent loop they (and the epilogue's addresses) stay live across the main loop and spill
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const int wr = wave >> 2, wc = wave & 3;
        const int a_row = wr * 128 + (lane & 15);
        const int a_kc = lane >> 4;
        const int b_q = (lane & 15) >> 2, b_p = lane & 3;
        const int b_row = 8 * (lane >> 4) + b_q;
        const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)smem;
        const uint32_t a_rd = smem_lds + a4_off(a_row, a_kc);
        const uint32_t b_rd = smem_lds + A2_BYTES + (wc >> 1) * (B2_BYTES / 2);
        uint32_t b_lo_off[4], b_hi_off[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int ch = (wc & 1) * 8 + ni * 2 + (b_p >> 1);
            b_lo_off[ni] = b_rd + b_off(b_row, ch) + 8 * (b_p & 1);
            b_hi_off[ni] = b_rd + b_off(b_row + 4, ch) + 8 * (b_p & 1);
        }

        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        const uint16_t* a_src[2];
        const uint16_t* b_src[2];
        int64_t a_dt[2];
        int b_dst[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ar = (wave * 2 + p) * 16 + (lane >> 2);
            const int64_t arow = (m0 + ar < M) ? m0 + ar : M - 1;
            a_src[p] = A + arow * lda + (((lane & 3) ^ a4_swz(ar)) << 3);
            a_dt[p] = tail_delta_a(A, At, arow, lda, Kmain);
            const int q = wave * 2 + p, half = q >> 3;
            const int br = (q & 7) * 4 + (lane >> 4);
            int64_t bcol = n0 + half * 128 + (((lane & 15) ^ (((br & 3) << 2) | ((br >> 2) & 3))) << 3);
            if (bcol > ((N - 1) & ~(int64_t)7)) bcol = (N - 1) & ~(int64_t)7;
            b_src[p] = Bm + (int64_t)br * ldb + bcol;
            b_dst[p] = A2_BYTES + half * (B2_BYTES / 2) + (q & 7) * 1024;
        }
        const int64_t b_dt = tail_delta_b(Bm, Bt, ldb, Kmain);
        auto dma_tile = [&](int64_t tn_) {
            const int stage = (int)(tn_ & (NST - 1));
            const int64_t k0 = (kb + (tn_ < ksteps ? tn_ : ksteps - 1)) * BK4;
            const bool in_tail = k0 >= Kmain;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[p] + k0 + (in_tail ? a_dt[p] : 0)),
                                                 (__attribute__((address_space(3))) void*)(smem + stage * STAGE2_BYTES + (wave * 2 + p) * 1024),
                                                 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[p] + k0 * ldb + (in_tail ? b_dt : 0)),
                                                 (__attribute__((address_space(3))) void*)(smem + stage * STAGE2_BYTES + b_dst[p]), 16, 0, 0);
            }
        };
        s16x8 af[8];
        s16x4 blo[4], bhi[4];
        auto read_all = [&](int64_t tt) {
            const uint32_t st = (uint32_t)(tt & (NST - 1)) * STAGE2_BYTES;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo[ni]) : "v"(b_lo_off[ni] + st));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bhi[ni]) : "v"(b_hi_off[ni] + st));
            }
            const uint32_t aa = a_rd + st;
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(af[1]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[2]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(af[3]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[4]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(af[5]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(af[6]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(af[7]) : "v"(aa));
        };
        auto wait_reads = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(blo[0]), "+v"(blo[1]), "+v"(blo[2]), "+v"(blo[3]), "+v"(bhi[0]), "+v"(bhi[1]), "+v"(bhi[2]),
                           "+v"(bhi[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]),
                           "+v"(af[6]), "+v"(af[7]));
        };
        auto compute = [&]() {
            s16x8 bf[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                bf[ni] = s16x8{blo[ni].x, blo[ni].y, blo[ni].z, blo[ni].w, bhi[ni].x, bhi[ni].y, bhi[ni].z, bhi[ni].w};
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16<IS_BF16>(af[mi], bf[ni], acc[mi][ni]);
            __builtin_amdgcn_sched_barrier(0);
        };
        // the ping-pong main loop of gemm_dma256_kernel (VAR 2), over K-steps kb .. ke-1 of this tile
        dma_tile(0);
        dma_tile(1);
        dma_tile(2);
        if (__builtin_amdgcn_readfirstlane(wave) < 4) {
            dma_tile(3);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            read_all(0);
            for (int64_t kt = 0; kt < ksteps; ++kt) {
                wait_reads();
                compute();
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                read_all(kt + 1);
                dma_tile(kt + 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int64_t kt = 0; kt < ksteps; ++kt) {
                read_all(kt);
                dma_tile(kt + 3);
                wait_reads();
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                compute();
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(blo[0]), "+v"(blo[1]), "+v"(blo[2]), "+v"(blo[3]), "+v"(bhi[0]), "+v"(bhi[1]), "+v"(bhi[2]),
                       "+v"(bhi[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]),
                       "+v"(af[6]), "+v"(af[7])
                     :
                     : "memory");
        __syncthreads();  // every DMA landed, every fragment read: the stages are free (epilogue staging, next piece)

        int mi_lo = 0, mi_hi = 8;   // the 16-row blocks of every wave's 128 rows this workgroup stores
        if (is_tail && !(order & 4)) {
            const int b0 = (int)blockIdx.x - tail_j;   // the tile's workgroups: blocks b0 .. b0 + split - 1
            mi_lo = tail_j * (8 / split);
            mi_hi = mi_lo + 8 / split;
            f32x4* slot = reinterpret_cast<f32x4*>(slots + (size_t)blockIdx.x * SK_SLOT_FLOATS) + tid;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < mi_lo || i >= mi_hi) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) slot[(i * 4 + j) * 512] = acc[i][j];
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int p = 0; p < split; ++p) {
                    if (p == tail_j) continue;
                    unsigned spins = 0;
                    while (__hip_atomic_load(flags + b0 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > SK_SPIN_LIMIT) {
                            poisoned = 1;
                            break;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            for (int p = 0; p < split; ++p) {
                if (p == tail_j) continue;
                const f32x4* ps = reinterpret_cast<const f32x4*>(slots + (size_t)(b0 + p) * SK_SLOT_FLOATS) + tid;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i >= mi_lo && i < mi_hi) {
                        f32x4 part[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) part[j] = ps[(i * 4 + j) * 512];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] += part[j];
                    }
            }
            if (poisoned) {   // a partner never showed up: make the failure visible in the output instead of hanging
                const float bad = __builtin_nanf("");
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{bad, bad, bad, bad};
            }
        }
        // the epilogue of gemm_dma256_kernel
        const int pr = lane >> 3, pc = (lane & 7) * 8;
        float* ctile = reinterpret_cast<float*>(smem) + wave * (EPI2_ROWS * CS);
        const bool pre_ok = vec_c && addend != nullptr && m0 + BM2 <= M && n0 + BN2 <= N;
        u32x4 pre[16];
        if (pre_ok) {
            const T* ap = addend + (m0 + wr * 128 + pr) * ldadd + n0 + wc * 64 + pc;
#pragma unroll
            for (int i = 0; i < 16; ++i) pre[i] = *reinterpret_cast<const u32x4*>(ap + (int64_t)i * 8 * ldadd);
        }
#pragma unroll
        for (int c = 0; c < 128 / EPI2_ROWS; ++c) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ctile[(h * 16 + (lane >> 4) * 4 + r) * CS + ni * 16 + (lane & 15)] = acc[c * 2 + h][ni][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int pass = 0; pass < EPI2_ROWS / 8; ++pass) {
                if (c * 2 + pass / 2 < mi_lo || c * 2 + pass / 2 >= mi_hi) continue;
                const int rr = pass * 8 + pr;
                float f[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&ctile[rr * CS + pc]);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(&ctile[rr * CS + pc + 4]);
                f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
                const int64_t row = m0 + wr * 128 + c * EPI2_ROWS + rr, col = n0 + wc * 64 + pc;
                if (pre_ok) {
                    float g[8];
                    Elem<T>::unpack(pre[c * 4 + pass], g);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] += g[i];
                    *reinterpret_cast<u32x4*>(C + row * N + col) = Elem<T>::pack(f);
                } else {
                    epi_store8<T>(C, addend, row, col, M, N, f, vec_c, half_c, ldadd);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();  // the staging rows of every wave are read: the next piece's DMA may overwrite them
    }
}

inline int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        n = v;
    }
    return n;
}

// The K split of the last round's tiles (1: plain grid launch — whole rounds already, more than half a round left, or off).
inline int sk_split_of(int64_t T, int G) {
    const char* sw = getenv("GNNOPS_GEMM_SK");  // A/B (tools/time_gemm_sk.py): 0 = off, 3 = persistent loop for whole rounds too
    if (sw && sw[0] == '0') return 1;
    // T < G (every tile a "last round" tile, cut so that most CUs get a piece) was tried and lost: 49 tiles of L = 1581 as
    // 196 pieces 54 vs 41 us, 64 tiles of L = 2000 58 vs 37, 100 tiles as 200 pieces 75 vs 75 — three to seven 192-KiB partial
    // tiles per workgroup cost more than the short K loops save (profiles/round3_f_gemm_streamk.txt)
    if ((G & 7) != 0 || T < G) return 1;
    const int64_t r = T % G;
    if (r == 0) return (sw && sw[0] == '3') ? 8 : 1;
    const int per_xcd = G / 8;   // tail tiles one XCD can take per piece
    if (r <= per_xcd) return 8;
    if (r <= 2 * per_xcd) return 4;
    if (r <= 4 * per_xcd) return 2;
    return 1;
}
inline size_t sk_flag_bytes(int G) { return ((size_t)G * 4 + 1023) / 1024 * 1024; }
inline size_t sk_workspace_bytes(int G) { return (size_t)G * SK_SLOT_FLOATS * 4 + sk_flag_bytes(G); }

template <typename T, bool IS_BF16>
int launch_sk256(const void* input, const void* mat1, const void* mat2, void* out, int64_t M, int64_t N, int64_t K, int64_t lda,
                 int64_t ldb, int64_t ldadd, const uint16_t* at, const uint16_t* bt, int64_t Kmain, void* sk_ws, int split,
                 hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sk256_kernel<T, IS_BF16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, GEMM256_SMEM) != hipSuccess)
            return gnnops_check_launch("addmm attribute");
        configured = true;
    }
    const int G = cu_count();
    unsigned* flags = reinterpret_cast<unsigned*>((char*)sk_ws + (size_t)G * SK_SLOT_FLOATS * 4);
    // cleared by a kernel, not a memset node (common.h gnnops_memset_async)
    if (gnnops_memset_async(flags, 0, sk_flag_bytes(G), stream) != hipSuccess) return gnnops_check_launch("addmm flags");
    const int tiles_m = (int)gnnops_cdiv(M, BM2), tiles_n = (int)gnnops_cdiv(N, BN2);
    const char* od = getenv("GNNOPS_GEMM_SK_ORDER");  // A/B: bit 0 = workgroup id by XCD share, bit 1 = tile ids down bands of 8 rows
    int order = od ? atoi(od) : 3;
    // bits 4 and 8 (no hand-off / no tail) leave the last round's tiles WRONG: they exist for tools/time_gemm_sk_parts.py, which also
    // sets GNNOPS_GEMM_SK_TIMING_ONLY — without that they are ignored
    if (!getenv("GNNOPS_GEMM_SK_TIMING_ONLY")) order &= 3;
    const int tail_tiles = (int)((int64_t)tiles_m * tiles_n % G), dp_tiles = tiles_m * tiles_n - tail_tiles;
    hipLaunchKernelGGL((gemm_sk256_kernel<T, IS_BF16>), dim3((unsigned)G), dim3(512), GEMM256_SMEM, stream, (const uint16_t*)mat1,
                       (const uint16_t*)mat2, (const T*)input, (T*)out, M, N, K, lda, ldb, ldadd, at, bt, Kmain, (float*)sk_ws, flags, tiles_m,
                       tiles_n, dp_tiles, tail_tiles, split, order);
    return gnnops_check_launch("addmm stream-K");
}

// ---- fp32 operands: v_mfma_f32_16x16x4_f32 (exact fp32 products and sums, 1/16 of the bf16 MFMA rate = the fp32
// vector peak, MI355X_MICROARCH.md "Matrix cores"). Same 128 x 128 block / 2 x 2 waves / 4 x 4 MFMA tiles; BK = 16.
// A tile [128][16] with 20-float rows and B tile [16][128] with 144-float rows: both fragment reads (one ds_read_b32
// per lane: A[row l&15][k l>>4], B[k l>>4][col l&15]) are bank-conflict-free, and B needs no transpose at all.
constexpr int FBK = 16, FAS = 20, FBS = 144;

__device__ inline f32x4 load4f(const float* __restrict__ base, int64_t r, int64_t c, int64_t rows, int64_t cols, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r >= rows || c >= cols) return v;
    const float* p = base + r * cols + c;
    if (vec && c + 4 <= cols) return *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (c + i < cols) v[i] = p[i];
    return v;
}

__global__ __launch_bounds__(256, 4) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                          const float* __restrict__ addend, float* __restrict__ C,
                                                          int64_t M, int64_t N, int64_t K, bool a_vec, bool b_vec, int64_t ldadd) {
    constexpr int STAGE_F = BM * FAS + FBK * FBS;       // floats per stage (2560 + 2304)
    constexpr int EPI_F = 4 * 32 * CS;
    constexpr int SMEM_F = (2 * STAGE_F > EPI_F) ? 2 * STAGE_F : EPI_F;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_F];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: A 128 x 16 floats = 512 float4 chunks (2 per thread), B 16 x 128 = 512 chunks (2 per thread)
    f32x4 ra[2], rb[2];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int id = tid + 256 * p;
            ra[p] = load4f(A, m0 + (id >> 2), k0 + (id & 3) * 4, M, K, a_vec);
            rb[p] = load4f(Bm, k0 + (id >> 5), n0 + (id & 31) * 4, K, N, b_vec);
        }
    };
    auto sstore = [&](int stage) {
        float* sA = smem + stage * STAGE_F;
        float* sB = sA + BM * FAS;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int id = tid + 256 * p;
            *reinterpret_cast<f32x4*>(&sA[(id >> 2) * FAS + (id & 3) * 4]) = ra[p];
            *reinterpret_cast<f32x4*>(&sB[(id >> 5) * FBS + (id & 31) * 4]) = rb[p];
        }
    };

    const int64_t ksteps = (K + FBK - 1) / FBK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int64_t kt = 0; kt < ksteps; ++kt) {
        const int cur = (int)(kt & 1);
        const bool more = kt + 1 < ksteps;
        if (more) gload((kt + 1) * FBK);
        const float* sA = smem + cur * STAGE_F;
        const float* sB = sA + BM * FAS;
#pragma unroll
        for (int ks = 0; ks < FBK / 4; ++ks) {
            float af[4], bf[4];
            const int kk = ks * 4 + (lane >> 4);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = sA[(wr * 64 + mi * 16 + (lane & 15)) * FAS + kk];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bf[ni] = sB[kk * FBS + wc * 64 + ni * 16 + (lane & 15)];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue: each wave stages its 64 x 64 tile through LDS in two rounds of 32 rows, so the staging area (34 KiB) stays
    // below the main loop's two stages (38 KiB) and three workgroups share a CU (one round of 64 rows: 68 KiB, two).
    // The K loop ended on a barrier: every wave is done with the stages.
    float* ctile = smem + wave * (32 * CS);
    const bool vec_c = (N % 4 == 0) && ((uintptr_t)C % 16 == 0) && (addend == nullptr || (uintptr_t)addend % 16 == 0);
    const int pr = lane >> 4, pc = (lane & 15) * 4;  // 16 lanes x 4 columns per row, 4 rows per pass
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ctile[(h * 16 + (lane >> 4) * 4 + r) * CS + ni * 16 + (lane & 15)] = acc[c * 2 + h][ni][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int rr = pass * 4 + pr;
            const int64_t row = m0 + wr * 64 + c * 32 + rr;
            const int64_t col = n0 + wc * 64 + pc;
            f32x4 v = *reinterpret_cast<const f32x4*>(&ctile[rr * CS + pc]);
            if (row >= M || col >= N) continue;
            if (vec_c && col + 4 <= N) {
                if (addend) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(addend + row * ldadd + col);
                    v[0] += g[0]; v[1] += g[1]; v[2] += g[2]; v[3] += g[3];
                }
                *reinterpret_cast<f32x4*>(C + row * N + col) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (col + i < N) C[row * N + col + i] = v[i] + (addend ? addend[row * ldadd + col + i] : 0.f);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- fp32, big tiles: 256 x 256 block tile, eight waves of 128 x 64, operands staged by LDS-DMA --------------------
// The 128 x 128 kernel above reads 16 KiB of operands per 0.5 MFLOP; at the reference's (48000)^3 (data/native_addmm.csv:2)
// that is ~4.7 TB/s of L2 -> LDS traffic and 32 fragment reads per 64 MFMAs, and the chip holds a lower clock under it than
// under a kernel that moves half of that. Here: BK = 16, stage = A [256][16] (64-B rows: the a4_off image of the 16-bit
// kernels, a lane's fragment = ONE ds_read_b128 = its row's k = 4q .. 4q+3) + B [16][256] with a row pitch of 1040 B (the
// four k-rows a 32-lane half reads fall on disjoint banks). The four MFMA steps of a K-step take k = 4q + j from lane
// group q (any order of k is a valid order of the sum: the SAME permutation on both operands), so A needs no scalar
// reads at all. Three stages; the DMA of tile t+2 is issued right after the fragments of tile t are in registers, so at
// the top of a K-step the only outstanding DMA is one whole K-step (>= 4096 MFMA cycles) old: a plain vmcnt(0) there
// costs nothing and no hand-counted wait is needed. Tiles are handed out XCD-contiguously in 8-wide column strips:
// the 32 workgroups of an XCD work on 4 x 8 neighbouring tiles (4 A panels + 8 B panels through one L2).
// Needs K % 16 == 0, N % 4 == 0 and 16-B aligned operands (the host falls back to the 128 x 128 kernel otherwise);
// any M and N (filler rows / columns, guarded epilogue).
constexpr int F2_BK = 16, F2_NST = 3;
constexpr int F2_A_BYTES = 256 * F2_BK * 4;   // 16 KiB
constexpr int F2_BROW = 1040;                  // B row pitch in bytes: 256 floats + 16 B
constexpr int F2_B_BYTES = F2_BK * F2_BROW;    // 16640
constexpr int F2_STAGE = F2_A_BYTES + F2_B_BYTES;
constexpr int F2_SMEM = F2_NST * F2_STAGE;     // 99072 B (the epilogue's 8 x 32 x 68 floats = 69632 B fit inside)

// DBG (timing-only builds behind GNNOPS_GEMM_F32_DBG, wrong results): 1 no DMA inside the loop, 2 no fragment reads inside
// the loop, 3 no barrier / wait at the top of a K-step — which part of a K-step the matrix pipe waits for.
template <bool STAGGER, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_f32_dma256_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                                 const float* __restrict__ addend, float* __restrict__ C,
                                                                 int64_t M, int64_t N, int64_t K, int tiles_m, int tiles_n, int64_t ldadd) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smemf[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;   // 2 x 4 waves: rows wr*128, columns wc*64

    // tile of this workgroup: XCD-contiguous order over column strips of 8 tiles (row-major inside a strip)
    constexpr int W = 8;
    const int64_t t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int full_strips = tiles_n / W;
    const int64_t in_full = (int64_t)full_strips * W * tiles_m;
    int by, bx;
    if (t < in_full) {
        const int64_t strip = t / ((int64_t)W * tiles_m), r = t % ((int64_t)W * tiles_m);
        by = (int)(r / W);
        bx = (int)(strip * W + r % W);
    } else {
        const int wl = tiles_n - full_strips * W;
        const int64_t r = t - in_full;
        by = (int)(r / wl);
        bx = full_strips * W + (int)(r % wl);
    }
    const int64_t m0 = (int64_t)by * 256, n0 = (int64_t)bx * 256;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's DMA pieces per stage: A pieces 2w, 2w+1 (16 rows x 64 B each), B rows 2w, 2w+1 (1 KiB each)
    const float* a_src[2];
    const float* b_src[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int ar = (wave * 2 + p) * 16 + (lane >> 2);
        const int64_t arow = (m0 + ar < M) ? m0 + ar : M - 1;            // in-bounds filler: the epilogue drops those rows
        a_src[p] = A + arow * K + (((lane & 3) ^ a4_swz(ar)) << 2);
        int64_t bcol = n0 + lane * 4;
        if (bcol > N - 4) bcol = N - 4;                                    // filler columns, dropped by the epilogue
        b_src[p] = Bm + (int64_t)(wave * 2 + p) * N + bcol;
    }
    auto dma = [&](int stage, int64_t k0) {
        unsigned char* base = smemf + stage * F2_STAGE;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[p] + k0),
                                             (__attribute__((address_space(3))) void*)(base + (wave * 2 + p) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[p] + k0 * N),
                                             (__attribute__((address_space(3))) void*)(base + F2_A_BYTES + (wave * 2 + p) * F2_BROW),
                                             16, 0, 0);
        }
    };

    const int a_row = wr * 128 + (lane & 15);
    const int q = lane >> 4;
    const int64_t ksteps = K / F2_BK;
    dma(0, 0);
    if (ksteps > 1) dma(1, F2_BK);
    f32x4 af[8], af2[8];
    float bf[4][4], bf2[4][4];
    auto read_into = [&](int64_t kt, f32x4 (&fa)[8], float (&fb)[4][4]) {
        const unsigned char* sA = smemf + (int)(kt % F2_NST) * F2_STAGE;
        const unsigned char* sB = sA + F2_A_BYTES;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) fa[mi] = *reinterpret_cast<const f32x4*>(sA + a4_off(a_row + mi * 16, q));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                fb[ni][j] = *reinterpret_cast<const float*>(sB + (4 * q + j) * F2_BROW + (wc * 64 + ni * 16 + (lane & 15)) * 4);
    };
    auto mfma_from = [&](int j0, const f32x4 (&fa)[8], const float (&fb)[4][4]) {
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi][j], fb[ni][j], acc[mi][ni], 0, 0, 0);
    };
    auto read_frags = [&](int64_t kt) { read_into(kt, af, bf); };
    auto mfma_half = [&](int j0) { mfma_from(j0, af, bf); };
    if constexpr (DBG == 5) {
        // REGISTER-STAGED operands: global_load_dwordx4 into registers right after the fragment reads, ds_write_b128 after the
        // MFMAs. Same LDS image (the swizzle is on the source address, the write goes to piece * 1 KiB + lane * 16 B). A
        // global_load_lds instruction costs the issuing wave ~100 cycles of issue during which it feeds no MFMA
        // (MI355X_MICROARCH.md "LDS-DMA piece issue cost"); a dwordx4 load ~4 and a ds_write_b128 ~13.
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 ra[2], rb[2];
        auto gload = [&](int64_t k0) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                ra[p] = *reinterpret_cast<const f4*>(a_src[p] + k0);
                rb[p] = *reinterpret_cast<const f4*>(b_src[p] + k0 * N);
            }
        };
        auto swrite = [&](int stage) {
            unsigned char* base = smemf + stage * F2_STAGE;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                *reinterpret_cast<f4*>(base + (wave * 2 + p) * 1024 + lane * 16) = ra[p];
                *reinterpret_cast<f4*>(base + F2_A_BYTES + (wave * 2 + p) * F2_BROW + lane * 16) = rb[p];
            }
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two tiles the DMA prologue fetched
        for (int64_t kt = 0; kt < ksteps; ++kt) {
            __syncthreads();
            read_frags(kt);
            const bool more = kt + 2 < ksteps;
            if (more) gload((kt + 2) * F2_BK);
            mfma_half(0);
            mfma_half(2);
            if (more) swrite((int)((kt + 2) % F2_NST));
        }
    } else if constexpr (DBG == 4) {
        // SOFTWARE-PIPELINED fragments: the barrier for tile kt+1 and its fragment reads (into the second register set) sit
        // in the MIDDLE of tile kt's MFMA block, so the burst of LDS reads all eight waves issue behind a barrier
        // (~600 LDS cycles for 8 x 16 reads) runs under the second half's 64 MFMAs instead of in front of an idle pipe.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        read_into(0, af, bf);
        auto step = [&](int64_t kt, f32x4 (&ca)[8], float (&cb)[4][4], f32x4 (&na)[8], float (&nb)[4][4]) {
            mfma_from(0, ca, cb);
            if (kt + 1 < ksteps) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kt+1: issued one K-step ago
                __syncthreads();
                read_into(kt + 1, na, nb);
                if (kt + 2 < ksteps) dma((int)((kt + 2) % F2_NST), (kt + 2) * F2_BK);   // stage of tile kt-1: read a K-step ago
            }
            mfma_from(2, ca, cb);
        };
        int64_t kt = 0;
        for (; kt + 1 < ksteps; kt += 2) {
            step(kt, af, bf, af2, bf2);
            step(kt + 1, af2, bf2, af, bf);
        }
        if (kt < ksteps) step(kt, af, bf, af2, bf2);
    } else
    if constexpr (!STAGGER) {
        for (int64_t kt = 0; kt < ksteps; ++kt) {
            if (DBG != 3) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tiles kt (and kt+1: a whole K-step old)
                __syncthreads();                                    // everyone's pieces landed; everyone is done reading tile kt-1
            }
            if (DBG != 2 || kt == 0) read_frags(kt);
            if (DBG != 1 && kt + 2 < ksteps) dma((int)((kt + 2) % F2_NST), (kt + 2) * F2_BK);   // its stage was read in step kt-1
            mfma_half(0);
            mfma_half(2);
        }
    } else {
        // PING-PONG: wave w and wave w+4 share a SIMD. Waves 4-7 run half a K-step behind waves 0-3: in every phase one
        // of the two does its memory work (wait, barrier, fragment reads, four DMA instructions — each ~100 cycles of
        // issue during which the wave feeds no MFMA) and the first 64 MFMAs of its tile, while its partner issues the last
        // 64 MFMAs of ITS tile back to back; in lockstep both partners stalled at the same time and the matrix pipe idled
        // ~12 % of a K-step. One barrier per phase. A wave waits for its own DMA pieces (vmcnt(0)) before the barrier
        // that opens its memory phase: pieces of tile t are issued two of the wave's memory phases earlier and are
        // visible to everyone from the barrier after the wave's NEXT memory phase on — before any wave reads tile t.
        // The two groups run the same loop body, shifted by one barrier: b0 | G0: mem(0) + first half | b1 | G0: second
        // half, G1: mem(0) + first half | b2 | G0: mem(1) + first half, G1: second half | ... Every wave passes the same
        // number of barriers (2 * ksteps + 1).
        const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
        if (grp == 1) {                                             // b0: waves 4-7 start one phase late
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (their pieces of tiles 0 and 1 are waited for here)
            __syncthreads();
        }
        for (int64_t kt = 0; kt < ksteps; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            read_frags(kt);
            if (kt + 2 < ksteps) dma((int)((kt + 2) % F2_NST), (kt + 2) * F2_BK);
            mfma_half(0);
            // phase boundary only: no wait here (a `__syncthreads()` would wait vmcnt(0) for the DMA issued half a K-step
            // ago). What the protocol needs from this barrier — this wave's fragment reads of tile kt are complete — holds:
            // the MFMAs above consumed them.
            asm volatile("s_barrier" ::: "memory");
            mfma_half(2);
        }
        if (grp == 0) asm volatile("s_barrier" ::: "memory");      // the barrier waves 4-7 opened their last half with
    }
    __syncthreads();   // the stages are free: the epilogue reuses them

    // epilogue: each wave stages its 128 x 64 tile through LDS in four rounds of 32 rows and writes 16-B row pieces
    float* ctile = reinterpret_cast<float*>(smemf) + wave * (32 * CS);
    const bool vec_c = (N % 4 == 0) && ((uintptr_t)C % 16 == 0) && (addend == nullptr || (uintptr_t)addend % 16 == 0);
    const int pr = lane >> 4, pc = (lane & 15) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ctile[(h * 16 + (lane >> 4) * 4 + r) * CS + ni * 16 + (lane & 15)] = acc[c * 2 + h][ni][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int rr = pass * 4 + pr;
            const int64_t row = m0 + wr * 128 + c * 32 + rr;
            const int64_t col = n0 + wc * 64 + pc;
            f32x4 v = *reinterpret_cast<const f32x4*>(&ctile[rr * CS + pc]);
            if (row >= M || col >= N) continue;
            if (vec_c && col + 4 <= N) {
                if (addend) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(addend + row * ldadd + col);
                    v[0] += g[0]; v[1] += g[1]; v[2] += g[2]; v[3] += g[3];
                }
                *reinterpret_cast<f32x4*>(C + row * N + col) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (col + i < N) C[row * N + col + i] = v[i] + (addend ? addend[row * ldadd + col + i] : 0.f);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- fp32, one wave per SIMD: 256 x 256 block tile, FOUR waves of 128 x 128 --------------------------------------
// The eight-wave kernel above keeps the matrix pipe busy 87.7 % of the cycles (PMC, profiles/round2_d_gemm_f32_pmc.txt): all
// eight waves read their fragments in one burst behind each barrier (~600 LDS cycles per 8192 MFMA cycles) and two waves
// of 256 registers have no room to double-buffer them. Here a wave owns a whole SIMD and its 512-register file: 256
// accumulator registers (8 x 8 MFMA tiles), two sets of fragments (A 8 x b128, B 8 x 4 scalars: 64 registers each). The
// barrier for tile t+1, its fragment reads and the DMA of tile t+2 sit in the MIDDLE of tile t's 256 MFMAs, the DMA
// instructions spread between MFMA groups, so nothing the wave waits for is on the critical path of the matrix pipe.
// Same stages, LDS images, k-permutation and tile order as gemm_f32_dma256_kernel; LDS traffic per K-step is 64 KiB
// instead of 96 (128 x 128 wave tiles).
constexpr int W4_CS = 132;   // epilogue row stride in floats (128 + 4)

__global__ __launch_bounds__(256, 1) void gemm_f32_w4_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                             const float* __restrict__ addend, float* __restrict__ C,
                                                             int64_t M, int64_t N, int64_t K, int tiles_m, int tiles_n, int64_t ldadd) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smemw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;   // 2 x 2 waves: rows wr*128, columns wc*128

    constexpr int W = 8;
    const int64_t t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int full_strips = tiles_n / W;
    const int64_t in_full = (int64_t)full_strips * W * tiles_m;
    int by, bx;
    if (t < in_full) {
        const int64_t strip = t / ((int64_t)W * tiles_m), r = t % ((int64_t)W * tiles_m);
        by = (int)(r / W);
        bx = (int)(strip * W + r % W);
    } else {
        const int wl = tiles_n - full_strips * W;
        const int64_t r = t - in_full;
        by = (int)(r / wl);
        bx = full_strips * W + (int)(r % wl);
    }
    const int64_t m0 = (int64_t)by * 256, n0 = (int64_t)bx * 256;

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's DMA pieces per stage: A pieces 4w .. 4w+3 (16 rows x 64 B each), B rows 4w .. 4w+3 (1 KiB each)
    const float* a_src[4];
    const float* b_src[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int ar = (wave * 4 + p) * 16 + (lane >> 2);
        const int64_t arow = (m0 + ar < M) ? m0 + ar : M - 1;
        a_src[p] = A + arow * K + (((lane & 3) ^ a4_swz(ar)) << 2);
        int64_t bcol = n0 + lane * 4;
        if (bcol > N - 4) bcol = N - 4;
        b_src[p] = Bm + (int64_t)(wave * 4 + p) * N + bcol;
    }
    // LDS-DMA as ONE asm statement (M0 = the wave-uniform LDS destination, written in the statement that reads it —
    // cdna_hip_programming.md §5.7): as a builtin the compiler drains every outstanding LDS read before it (lgkmcnt(0)) and
    // every outstanding DMA before the next LDS read (vmcnt(0)); here the protocol below owns both orders.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smemw;
    auto glds = [&](const float* gsrc, uint32_t lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    };
    auto dma_a = [&](int stage, int64_t k0, int p) { glds(a_src[p] + k0, lds0 + stage * F2_STAGE + (wave * 4 + p) * 1024); };
    auto dma_b = [&](int stage, int64_t k0, int p) {
        glds(b_src[p] + k0 * N, lds0 + stage * F2_STAGE + F2_A_BYTES + (wave * 4 + p) * F2_BROW);
    };
    auto dma_all = [&](int stage, int64_t k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { dma_a(stage, k0, p); dma_b(stage, k0, p); }
    };

    const int a_row = wr * 128 + (lane & 15);
    const int q = lane >> 4;
    const int64_t ksteps = K / F2_BK;
    // ONE set of fragment registers in two halves that roll: (a_lo, b_lo) = k-steps j = 0, 1 of a tile, (a_hi, b_hi) =
    // j = 2, 3. While the MFMAs of one half run, the other half's registers are free and receive what comes next.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a_lo[8], a_hi[8];
    float b_lo[8][2], b_hi[8][2];
    auto read_half = [&](int64_t kt, int h, f32x2 (&fa)[8], float (&fb)[8][2]) {
        const unsigned char* sA = smemw + (int)(kt % F2_NST) * F2_STAGE;
        const unsigned char* sB = sA + F2_A_BYTES;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) fa[mi] = *reinterpret_cast<const f32x2*>(sA + a4_off(a_row + mi * 16, q) + h * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni)
                fb[ni][j] = *reinterpret_cast<const float*>(sB + (4 * q + 2 * h + j) * F2_BROW + (wc * 128 + ni * 16 + (lane & 15)) * 4);
    };
    // The 64 accumulator tiles fill the accumulator half of the register file exactly (256 AGPRs). As a builtin the
    // compiler rotates them through spare registers (D != C plus hundreds of v_accvgpr moves per K-step); as an asm
    // statement with a read-write "a" operand each tile stays where it is (accumulate chain: no wait states needed
    // between an MFMA and the next one that takes its D whole as C — cdna_hip_programming.md §5.7 item 2).
    auto mfma_j = [&](int j, const f32x2 (&fa)[8], const float (&fb)[8][2]) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni)
                asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[mi][ni]) : "v"(fa[mi][j]), "v"(fb[ni][j]));
    };

    dma_all(0, 0);
    if (ksteps > 1) dma_all(1, F2_BK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_half(0, 0, a_lo, b_lo);
    for (int64_t kt = 0; kt < ksteps; ++kt) {
        read_half(kt, 1, a_hi, b_hi);           // second half of THIS tile: lands under the 128 MFMAs below
        mfma_j(0, a_lo, b_lo);
        mfma_j(1, a_lo, b_lo);
        const bool next = kt + 1 < ksteps, more = kt + 2 < ksteps;
        const int st2 = (int)((kt + 2) % F2_NST);
        const int64_t k2 = (kt + 2) * F2_BK;
        if (next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kt+1: its DMA was issued one K-step ago
            __syncthreads();                                    // everyone's pieces landed; everyone has read all of tile kt
            read_half(kt + 1, 0, a_lo, b_lo);                   // first half of the NEXT tile, under the MFMAs below
        }
        // the DMA of tile kt+2 (its stage held tile kt-1) between the MFMA groups of the second half. (Staging these
        // operands through registers instead — dwordx4 loads here, ds_write_b128 after the last MFMA — measured SLOWER:
        // 117.6 vs 122.5 TFLOP/s at 8192^3, profiles/round2_d_gemm_f32_variants.txt.)
        if (more) { dma_a(st2, k2, 0); dma_b(st2, k2, 0); dma_a(st2, k2, 1); dma_b(st2, k2, 1); }
        mfma_j(0, a_hi, b_hi);
        if (more) { dma_a(st2, k2, 2); dma_b(st2, k2, 2); dma_a(st2, k2, 3); dma_b(st2, k2, 3); }
        mfma_j(1, a_hi, b_hi);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // the stages are free: the epilogue reuses them

    // epilogue: each wave stages its 128 x 128 tile through LDS in four rounds of 32 rows and writes 16-B row pieces
    float* ctile = reinterpret_cast<float*>(smemw) + wave * (32 * W4_CS);
    const bool vec_c = (N % 4 == 0) && ((uintptr_t)C % 16 == 0) && (addend == nullptr || (uintptr_t)addend % 16 == 0);
    const int pr = lane >> 5, pc = (lane & 31) * 4;   // 32 lanes x 4 columns per row, 2 rows per pass
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ctile[(h * 16 + (lane >> 4) * 4 + r) * W4_CS + ni * 16 + (lane & 15)] = acc[c * 2 + h][ni][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) {
            const int rr = pass * 2 + pr;
            const int64_t row = m0 + wr * 128 + c * 32 + rr;
            const int64_t col = n0 + wc * 128 + pc;
            f32x4 v = *reinterpret_cast<const f32x4*>(&ctile[rr * W4_CS + pc]);
            if (row >= M || col >= N) continue;
            if (vec_c && col + 4 <= N) {
                if (addend) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(addend + row * ldadd + col);
                    v[0] += g[0]; v[1] += g[1]; v[2] += g[2]; v[3] += g[3];
                }
                *reinterpret_cast<f32x4*>(C + row * N + col) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (col + i < N) C[row * N + col + i] = v[i] + (addend ? addend[row * ldadd + col + i] : 0.f);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

template <int ALIGN>
__global__ void pad_rows_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int64_t rows, int64_t cols,
                                int64_t ld, int64_t rows_out) {
    const int64_t chunks = ld / 8, total = rows_out * chunks;  // one 16-B output chunk per thread; rows >= `rows` are zero
    // each workgroup writes ONE contiguous span (DESIGN.md "store shape": grid-strided 16-B stores run 10-25 % below this)
    const int64_t span = (total + gridDim.x - 1) / gridDim.x;
    const int64_t i_end = ((int64_t)blockIdx.x + 1) * span < total ? ((int64_t)blockIdx.x + 1) * span : total;
    for (int64_t i = (int64_t)blockIdx.x * span + threadIdx.x; i < i_end; i += blockDim.x) {
        const int64_t r = i / chunks, c = (i % chunks) * 8;
        u32x4 v;
        if constexpr (ALIGN >= 4) {
            v = load8<ALIGN>(in, r, c, rows, cols);
        } else {
            // rows of odd length (the reference's L = int(sqrt(x))): every other row starts 2 bytes off a dword. Whole
            // dwords are read from the dword at or below the piece and shifted by 0 or 2 bytes — five loads (one dwordx4 +
            // one dword) instead of eight 2-byte ones; a dword is read only if it holds a valid element, so nothing past
            // the last row is touched, and elements past the row end are cleared after the shift.
            v = u32x4{0u, 0u, 0u, 0u};
            if (r < rows && c < cols) {
                const int64_t left = cols - c;
                const int n = left < 8 ? (int)left : 8;
                const uint16_t* p = in + r * cols + c;
                const int odd = (int)((reinterpret_cast<uintptr_t>(p) >> 1) & 1);
                const uint32_t* q = reinterpret_cast<const uint32_t*>(p - odd);
                const int nd = (odd + n + 1) >> 1;   // dwords holding elements odd .. odd + n - 1
                uint32_t d[5] = {0u, 0u, 0u, 0u, 0u};
                if (nd >= 4) {
                    const u32x4_a4 w = *reinterpret_cast<const u32x4_a4*>(q);
                    d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
                    if (nd == 5) d[4] = q[4];
                } else {
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (j < nd) d[j] = q[j];
                }
                if (odd) {
                    v.x = (d[0] >> 16) | (d[1] << 16); v.y = (d[1] >> 16) | (d[2] << 16);
                    v.z = (d[2] >> 16) | (d[3] << 16); v.w = (d[3] >> 16) | (d[4] << 16);
                } else {
                    v.x = d[0]; v.y = d[1]; v.z = d[2]; v.w = d[3];
                }
                if (n < 8) {   // the dwords may carry the next row's first elements
                    uint32_t* vw = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int keep = n - 2 * j;   // valid elements in word j
                        vw[j] = keep >= 2 ? vw[j] : keep == 1 ? (vw[j] & 0xffffu) : 0u;
                    }
                }
            }
        }
        *reinterpret_cast<u32x4*>(out + r * ld + c) = v;
    }
}

inline void launch_pad(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld, int64_t rows_out, hipStream_t stream) {
    const dim3 grid(gnnops_grid_cap(gnnops_cdiv(rows_out * ld / 8, 256), 256 * 16));
    const uint16_t* i = (const uint16_t*)in;
    uint16_t* o = (uint16_t*)out;
    if (cols % 4 == 0 && (uintptr_t)in % 8 == 0)
        hipLaunchKernelGGL(pad_rows_kernel<8>, grid, dim3(256), 0, stream, i, o, rows, cols, ld, rows_out);
    else if (cols % 2 == 0 && (uintptr_t)in % 4 == 0)
        hipLaunchKernelGGL(pad_rows_kernel<4>, grid, dim3(256), 0, stream, i, o, rows, cols, ld, rows_out);
    else
        hipLaunchKernelGGL(pad_rows_kernel<2>, grid, dim3(256), 0, stream, i, o, rows, cols, ld, rows_out);
}

// The last K-tile of both operands, zero-filled (see tail_delta_a): At[r][c] = A[r][Kmain + c] for c < K - Kmain, Bt[k][c] =
// B[Kmain + k][c] for k < K - Kmain; everything else 0. One launch; 2-byte loads (a megabyte or two in all).
__global__ void pad_tail_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bm, uint16_t* __restrict__ At,
                                uint16_t* __restrict__ Bt, int64_t na, int64_t N, int64_t K, int64_t Kmain, int64_t bt_elems) {
    const int64_t total = na + bt_elems;   // na = M * 64, or 0 when A is a whole padded copy; likewise bt_elems
    const int tail = (int)(K - Kmain);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < na) {
            const int64_t r = i >> 6;
            const int c = (int)(i & 63);
            At[i] = c < tail ? A[r * K + Kmain + c] : (uint16_t)0;
        } else {
            const int64_t j = i - na, k = j / N, c = j - k * N;
            Bt[j] = k < tail ? Bm[(Kmain + k) * N + c] : (uint16_t)0;   // k >= 64: the slack behind the last row
        }
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// How a 16-bit problem is run. The LDS-DMA kernels take any M and N (filler rows / columns, guarded epilogue) but whole
// K-tiles of 64 with zeros past K on BOTH operands (0 x Inf would poison valid outputs), and 16-B aligned rows; the
// register-staged kernel bounds-checks everything and only needs the 16-B aligned rows. Operands that do not comply
// are copied into the workspace with zero fill (pad_rows_kernel) — the reference sweeps odd sizes (L = 1581 ... 8164),
// for which the copies cost ~10 % of the product and the DMA kernels gain more than that.
struct GemmPlan {
    int path;             // 0 register-staged 128 x 128, 1 LDS-DMA 128 x 128, 2 LDS-DMA 256 x 256
    int64_t Kp, lda, ldb; // K as the kernel sees it, row lengths of the operands as the kernel sees them
    int64_t Kmain;        // K-steps at or past this come from the side copies of the last K-tile (== Kp: there are none)
    bool copy_a, copy_b;  // whole padded copies
    bool tail_a, tail_b;  // side copies of the last K-tile only
    size_t a_bytes, b_bytes, at_bytes, bt_bytes;
    int sk_split;         // path 2: K pieces of a last-round tile (<= 1: plain grid launch)
    size_t sk_bytes;      // slots + flags of gemm_sk256_kernel
};

// Whole padded copies or in-place reads? Measured over the reference's sweep (tools/time_gemm_pad.py,
// profiles/round3_f_gemm_pad_modes.txt): B read in place costs nothing at any size (256-B row pieces); A in place (64-B
// row pieces at a pitch that is no multiple of a cache line) costs 2 % of the product at L = 4684 and 15-20 % at L = 8045,
// so from ~30 M elements on A is copied into padded rows (a copy is L^2, the penalty L^3). The 128 x 128 kernel moves twice
// the bytes per flop: once there is more than one workgroup per CU both operands are copied.
inline void gemm_copy_whole(int64_t M, int64_t N, int64_t K, int path, bool& a, bool& b) {
    if (path == 1) {
        a = b = gnnops_cdiv(M, BM) * gnnops_cdiv(N, BN) > cu_count();
        return;
    }
    a = M * K >= 30000000;
    b = false;
}

inline GemmPlan gemm_plan(int64_t M, int64_t N, int64_t K) {
    GemmPlan g{};
    const char* sw = getenv("GNNOPS_GEMM_NO_DMA");  // A/B switch for tools/time_gemm.py: 1 = register staging, 3 = 128 x 128 DMA only
    const bool aligned = M % BM == 0 && N % BN == 0 && K % BK == 0;
    const bool dma = K > 0 && !(sw && sw[0] == '1') && (aligned || (M >= 512 && N >= 512 && K >= 256));
    if (dma) {
        const char* mn = getenv("GNNOPS_GEMM_MIN256");  // A/B: least number of 256 x 256 tiles that takes the big-tile kernel
        const int64_t min256 = mn ? atoll(mn) : 128;  // measured: 144 tiles 0.062 vs 0.086 ms, 100 tiles 0.056 vs 0.046 (tools/time_gemm_tiles.py)
        g.path = (gnnops_cdiv(M, BM2) * gnnops_cdiv(N, BN2) >= min256 && !(sw && sw[0] == '3')) ? 2 : 1;
        // Operands are read in place where that is cheaper, with only the last K-tile copied (zeros past K; a B row's last
        // 16-B piece may run into the next row, and behind B's very last row there is no next row), or copied whole into
        // aligned zero-padded rows. GNNOPS_GEMM_PAD = full | a | b | none forces a choice (tools/time_gemm_pad.py).
        const bool need = K % 64 != 0 || N % 8 != 0;
        const char* pd = getenv("GNNOPS_GEMM_PAD");
        bool whole_a = false, whole_b = false;
        if (pd && pd[0] == 'f') whole_a = whole_b = true;
        else if (pd && pd[0] == 'a') whole_a = true;
        else if (pd && pd[0] == 'b') whole_b = true;
        else if (pd && pd[0] == 'n') whole_a = whole_b = false;
        else gemm_copy_whole(M, N, K, g.path, whole_a, whole_b);
        g.Kp = round_up(K, 64);
        g.copy_a = whole_a && (K % 64 != 0 || K % 8 != 0);
        g.copy_b = whole_b && (K % 64 != 0 || N % 8 != 0);
        g.lda = g.copy_a ? g.Kp : K;
        g.ldb = g.copy_b ? round_up(N, 8) : N;
        g.tail_a = need && !g.copy_a;
        g.tail_b = need && !g.copy_b;
        g.Kmain = (g.tail_a || g.tail_b) ? (K % 64 ? K - K % 64 : K - 64) : g.Kp;
        g.a_bytes = g.copy_a ? align_up((size_t)M * g.lda * 2, 256) : 0;
        g.b_bytes = g.copy_b ? align_up((size_t)g.Kp * g.ldb * 2, 256) : 0;
        g.at_bytes = g.tail_a ? align_up((size_t)M * 64 * 2, 256) : 0;
        g.bt_bytes = g.tail_b ? align_up(((size_t)64 * N + 8) * 2, 256) : 0;
        if (g.path == 2) {
            g.sk_split = sk_split_of(gnnops_cdiv(M, BM2) * gnnops_cdiv(N, BN2), cu_count());
            g.sk_bytes = g.sk_split > 1 ? sk_workspace_bytes(cu_count()) : 0;
        }
    } else {
        // register staging bounds-checks every piece and reads rows of any length in the widest pieces their alignment
        // allows (load8<ALIGN>): no copies — these are the small, launch-bound problems (a batch of small graphs:
        // [9134, 11] @ [11, 44]), where two pad launches cost more than the product
        g.path = 0;
        g.Kp = K;
        g.Kmain = K;
        g.copy_a = g.copy_b = false;
        g.lda = K;
        g.ldb = N;
        g.a_bytes = g.b_bytes = 0;
    }
    return g;
}

}  // namespace

// the row-tile index travels in gridDim.y (< 65536): taller problems (a node-feature matrix of BASELINE config 2's 10M rows)
// run as slabs of whole tiles, one after the other on the stream, sharing the workspace
constexpr int64_t GEMM_SLAB = (int64_t)65280 * 128;

static size_t plan_bytes(int64_t M, int64_t N, int64_t K) {
    const GemmPlan g = gemm_plan(M, N, K);  // 16-bit operands only; fp32 needs none
    return g.a_bytes + g.b_bytes + g.at_bytes + g.bt_bytes + g.sk_bytes;
}

extern "C" size_t gnnops_addmm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M < 0 || N < 0 || K < 0) return 0;
    if (M <= GEMM_SLAB) return plan_bytes(M, N, K);
    // every slab plans for itself (tile count, copies, split-K tail): the largest need of the two slab heights that occur
    const size_t whole = plan_bytes(GEMM_SLAB, N, K), rest = M % GEMM_SLAB ? plan_bytes(M % GEMM_SLAB, N, K) : 0;
    return whole > rest ? whole : rest;
}

extern "C" int gnnops_addmm(const void* input, const void* mat1, const void* mat2, void* out, int64_t M, int64_t N,
                            int64_t K, int dtype, void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    return gnnops_addmm_ld(input, N, mat1, mat2, out, M, N, K, dtype, workspace, workspace_bytes, s);
}

extern "C" int gnnops_addmm_ld(const void* input, int64_t ldadd, const void* mat1, const void* mat2, void* out, int64_t M,
                               int64_t N, int64_t K, int dtype, void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(M >= 0 && N >= 0 && K >= 0, GNNOPS_EINVAL, "addmm: negative size");
    GNNOPS_REQUIRE(dtype == GNNOPS_F16 || dtype == GNNOPS_BF16 || dtype == GNNOPS_F32, GNNOPS_EUNSUPPORTED,
                   "addmm: unknown dtype code %d", dtype);
    GNNOPS_REQUIRE(ldadd == 0 || ldadd >= N, GNNOPS_EINVAL, "addmm: input row pitch must be 0 (one row for all) or >= N");
    // the epilogues' 16-B loads of the addend test its POINTER and N only: a pitch that is neither N nor a multiple of 16 bytes
    // would put rows 1.. on misaligned addresses
    GNNOPS_REQUIRE(ldadd == 0 || ldadd == N || (ldadd * (dtype == GNNOPS_F32 ? 4 : 2)) % 16 == 0, GNNOPS_EINVAL,
                   "addmm: an input row pitch other than 0 or N must be a multiple of 16 bytes (got %lld elements)", (long long)ldadd);
    if (M * N == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(out && (K == 0 || (mat1 && mat2)), GNNOPS_EINVAL, "addmm: null pointer");
    constexpr int64_t SLAB = GEMM_SLAB;
    if (M > SLAB) {
        const size_t es = dtype == GNNOPS_F32 ? 4 : 2;
        for (int64_t r0 = 0; r0 < M; r0 += SLAB) {
            const int64_t rows = M - r0 < SLAB ? M - r0 : SLAB;
            const int rc = gnnops_addmm_ld(input ? (const char*)input + (size_t)r0 * ldadd * es : nullptr, ldadd,
                                           (const char*)mat1 + (size_t)r0 * K * es, mat2, (char*)out + (size_t)r0 * N * es, rows, N, K,
                                           dtype, workspace, workspace_bytes, s);
            if (rc != GNNOPS_OK) return rc;
        }
        return GNNOPS_OK;
    }
    if (dtype == GNNOPS_F32) {
        const bool a_vec = (K % 4 == 0) && ((uintptr_t)mat1 % 16 == 0);
        const bool b_vec = (N % 4 == 0) && ((uintptr_t)mat2 % 16 == 0);
        // big problems with whole K-steps and 16-B aligned rows: 256 x 256 tiles staged by LDS-DMA (GNNOPS_GEMM_F32_BIG=0: off)
        const int64_t tm = gnnops_cdiv(M, 256), tn = gnnops_cdiv(N, 256);
        const char* big = getenv("GNNOPS_GEMM_F32_BIG");
        if (a_vec && b_vec && K % F2_BK == 0 && K >= 2 * F2_BK && N >= 4 && tm * tn >= 128 && tm * tn < ((int64_t)1 << 31) &&
            !(big && big[0] == '0')) {
            static bool configured = false;
            if (!configured) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_dma256_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, F2_SMEM) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_dma256_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, F2_SMEM) != hipSuccess)
                    return gnnops_check_launch("addmm f32 attribute");
                configured = true;
            }
            if (big && big[0] == '4') {   // one wave per SIMD: four waves of 128 x 128 (A/B: tools/time_gemm_f32_dbg.py)
                static bool cfg4 = false;
                if (!cfg4) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_w4_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, F2_SMEM) != hipSuccess)
                        return gnnops_check_launch("addmm f32 w4 attribute");
                    cfg4 = true;
                }
                hipLaunchKernelGGL(gemm_f32_w4_kernel, dim3((unsigned)(tm * tn)), dim3(256), F2_SMEM, stream, (const float*)mat1,
                                   (const float*)mat2, (const float*)input, (float*)out, M, N, K, (int)tm, (int)tn, ldadd);
                return gnnops_check_launch("addmm f32 w4");
            }
            const char* dbg = getenv("GNNOPS_GEMM_F32_DBG");
            if (dbg && dbg[0] >= '1' && dbg[0] <= '5') {
                auto kfn = dbg[0] == '1' ? &gemm_f32_dma256_kernel<false, 1> : dbg[0] == '2' ? &gemm_f32_dma256_kernel<false, 2>
                           : dbg[0] == '3' ? &gemm_f32_dma256_kernel<false, 3> : dbg[0] == '4' ? &gemm_f32_dma256_kernel<false, 4>
                           : &gemm_f32_dma256_kernel<false, 5>;
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, F2_SMEM) != hipSuccess)
                    return gnnops_check_launch("addmm f32 attribute");
                hipLaunchKernelGGL(kfn, dim3((unsigned)(tm * tn)), dim3(512), F2_SMEM, stream,
                                   (const float*)mat1, (const float*)mat2, (const float*)input, (float*)out, M, N, K, (int)tm, (int)tn, ldadd);
            } else if (big && big[0] == '1')   // A/B: all eight waves in lockstep (tools/time_gemm_f32.py)
                hipLaunchKernelGGL(gemm_f32_dma256_kernel<false>, dim3((unsigned)(tm * tn)), dim3(512), F2_SMEM, stream,
                                   (const float*)mat1, (const float*)mat2, (const float*)input, (float*)out, M, N, K, (int)tm, (int)tn, ldadd);
            else
                hipLaunchKernelGGL(gemm_f32_dma256_kernel<true>, dim3((unsigned)(tm * tn)), dim3(512), F2_SMEM, stream,
                                   (const float*)mat1, (const float*)mat2, (const float*)input, (float*)out, M, N, K, (int)tm, (int)tn, ldadd);
            return gnnops_check_launch("addmm f32 256");
        }
        dim3 fgrid((unsigned)gnnops_cdiv(N, BN), (unsigned)gnnops_cdiv(M, BM));
        hipLaunchKernelGGL(gemm_f32_kernel, fgrid, dim3(256), 0, stream, (const float*)mat1, (const float*)mat2,
                           (const float*)input, (float*)out, M, N, K, a_vec, b_vec, ldadd);
        return gnnops_check_launch("addmm f32");
    }
    const GemmPlan g = gemm_plan(M, N, K);
    const size_t need = g.a_bytes + g.b_bytes + g.at_bytes + g.bt_bytes + g.sk_bytes;
    GNNOPS_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), GNNOPS_EWORKSPACE, "addmm: workspace %zu < %zu",
                   workspace_bytes, need);
    int64_t lda = g.lda, ldb = g.ldb;
    char* w = (char*)workspace;
    const uint16_t* at = nullptr;   // null: the operand is a whole padded copy (or there is no tail at all)
    const uint16_t* bt = nullptr;
    const void* a_orig = mat1;
    const void* b_orig = mat2;
    if (g.copy_a) {
        launch_pad(mat1, w, M, K, lda, M, stream);
        mat1 = w;
        w += g.a_bytes;
    }
    if (g.copy_b) {
        launch_pad(mat2, w, K, N, ldb, g.Kp, stream);  // rows K .. Kp-1 zero
        mat2 = w;
        w += g.b_bytes;
    }
    if (g.path != 0 && (g.tail_a || g.tail_b)) {
        if (g.tail_a) at = (const uint16_t*)w;
        if (g.tail_b) bt = (const uint16_t*)(w + g.at_bytes);
        const int64_t at_elems = g.tail_a ? M * 64 : 0, bt_elems = g.tail_b ? 64 * N + 8 : 0;
        hipLaunchKernelGGL(pad_tail_kernel, dim3(gnnops_grid_cap(gnnops_cdiv(at_elems + bt_elems, 256), 256 * 8)), dim3(256), 0, stream,
                           (const uint16_t*)a_orig, (const uint16_t*)b_orig, (uint16_t*)w, (uint16_t*)(w + g.at_bytes), at_elems, N, K,
                           g.Kmain, bt_elems);
        w += g.at_bytes + g.bt_bytes;
    }
    if (g.path == 2 && g.sk_split > 1)
        return dtype == GNNOPS_BF16
                   ? launch_sk256<__hip_bfloat16, true>(input, mat1, mat2, out, M, N, g.Kp, lda, ldb, ldadd, at, bt, g.Kmain, w, g.sk_split, stream)
                   : launch_sk256<__half, false>(input, mat1, mat2, out, M, N, g.Kp, lda, ldb, ldadd, at, bt, g.Kmain, w, g.sk_split, stream);
    if (g.path == 2)
        return dtype == GNNOPS_BF16
                   ? launch_dma256<__hip_bfloat16, true>(input, mat1, mat2, out, M, N, g.Kp, lda, ldb, ldadd, at, bt, g.Kmain, stream)
                   : launch_dma256<__half, false>(input, mat1, mat2, out, M, N, g.Kp, lda, ldb, ldadd, at, bt, g.Kmain, stream);
    dim3 grid((unsigned)gnnops_cdiv(N, BN), (unsigned)gnnops_cdiv(M, BM));
    if (g.path == 1) {
        if (dtype == GNNOPS_BF16)
            hipLaunchKernelGGL((gemm_dma4_kernel<__hip_bfloat16, true>), grid, dim3(256), 0, stream, (const uint16_t*)mat1,
                               (const uint16_t*)mat2, (const __hip_bfloat16*)input, (__hip_bfloat16*)out, M, N, g.Kp, lda, ldb, ldadd,
                               at, bt, g.Kmain);
        else
            hipLaunchKernelGGL((gemm_dma4_kernel<__half, false>), grid, dim3(256), 0, stream, (const uint16_t*)mat1,
                               (const uint16_t*)mat2, (const __half*)input, (__half*)out, M, N, g.Kp, lda, ldb, ldadd, at, bt, g.Kmain);
        return gnnops_check_launch("addmm");
    }
    // path 0: the widest piece both operands' rows allow
    int align = 16;
    for (; align > 2; align >>= 1)
        if ((K * 2) % align == 0 && (N * 2) % align == 0 && (uintptr_t)mat1 % align == 0 && (uintptr_t)mat2 % align == 0) break;
#define GNNOPS_GEMM0(AL)                                                                                                        \
    do {                                                                                                                        \
        if (dtype == GNNOPS_BF16)                                                                                               \
            hipLaunchKernelGGL((gemm_kernel<__hip_bfloat16, true, AL>), grid, dim3(256), 0, stream, (const uint16_t*)mat1,     \
                               (const uint16_t*)mat2, (const __hip_bfloat16*)input, (__hip_bfloat16*)out, M, N, K, lda, ldb, ldadd); \
        else                                                                                                                    \
            hipLaunchKernelGGL((gemm_kernel<__half, false, AL>), grid, dim3(256), 0, stream, (const uint16_t*)mat1,             \
                               (const uint16_t*)mat2, (const __half*)input, (__half*)out, M, N, K, lda, ldb, ldadd);             \
    } while (0)
    switch (align) {
        case 16: GNNOPS_GEMM0(16); break;
        case 8: GNNOPS_GEMM0(8); break;
        case 4: GNNOPS_GEMM0(4); break;
        default: GNNOPS_GEMM0(2); break;
    }
#undef GNNOPS_GEMM0
    return gnnops_check_launch("addmm");
}
