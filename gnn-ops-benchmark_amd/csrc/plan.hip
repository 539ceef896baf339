// plan.hip — inverted index ("plan") of a destination index, built by a stable LSD radix sort.
//
// Replaces the implicit "who writes row n" structure behind torch_scatter.scatter_* /
// Tensor.index_add_ (reference call sites: op_bm_scripts/benchmark_scatter_add.py:15-19,
// benchmark_native_index_add_.py:13-16). The reference reaches those through atomics
// (ops_to_kernels.md:5,7); on MI355X float atomics are capped at ~1.3 TB/s of added bytes
// (MI355X_MICROARCH.md "Global float atomics"), so we sort once and reduce per destination.
//
// Sort engine: 8-bit digits, 8192-key tiles, three launches per pass:
//   radix_hist    per-tile digit counts                  (reads keys)
//   radix_scan    per-digit exclusive scan over tiles    (256 blocks)
//   radix_scatter stable in-tile ranking by wave ballots, reorder through LDS, coalesced run writes
// The same engine sorts fp32 keys for torch.sort (sort.hip) and 64-bit COO keys for coalesce.
#include "common.h"
#include "sort_engine.h"
#include <stdlib.h>

namespace {

// max over an int64 index: a streaming read — 16-B nontemporal lane loads, eight of them in flight per lane
typedef long long ll2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void index_max_kernel(const int64_t* __restrict__ index, int64_t E, int64_t* d_max) {
    int64_t m = INT64_MIN;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = ((uintptr_t)index & 15) == 0;
    const int64_t n2 = vec ? E / 2 : 0;
    const ll2_t* p = reinterpret_cast<const ll2_t*>(index);
    int64_t i = gtid;
    for (; i + 7 * stride < n2; i += 8 * stride) {
        ll2_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) m = max(m, max((int64_t)v[u].x, (int64_t)v[u].y));
    }
    for (; i < n2; i += stride) {
        const ll2_t a = __builtin_nontemporal_load(p + i);
        m = max(m, max((int64_t)a.x, (int64_t)a.y));
    }
    for (int64_t j = 2 * n2 + gtid; j < E; j += stride) m = max(m, index[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int64_t t = __shfl_xor(m, o);
        m = t > m ? t : m;
    }
    // ONE atomic per workgroup and few workgroups: atomics on one address are served one after the other (~12 ns each) —
    // with an atomic per wave of 2048 workgroups they were 100 of the kernel's 120 us on the reference's (6708)^2 index
    __shared__ int64_t s_m[4];
    if (lane_id() == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m != INT64_MIN) atomicMax((long long*)d_max, (long long)m);
    }
}

// rowptr[n] = first sorted position whose key is >= n. One thread per boundary i in [0, E].
// Long runs of empty destinations (gap > GAP_INLINE) are queued and filled by fill_gaps_kernel so
// that a skewed index (all edges on one node) does not serialise on one lane.
constexpr int GAP_INLINE = 32;

// One boundary i in [0, E]: destinations in (key[i-1], key[i]] start at sorted position i.
__device__ inline void rowptr_boundary(int64_t i, int64_t prev, int64_t cur, int64_t N, int32_t* __restrict__ rowptr,
                                       int32_t* __restrict__ gap_list, unsigned int* __restrict__ gap_count) {
    if (cur > N) cur = N;  // out-of-range keys cannot push writes past rowptr[N]
    const int64_t gap = cur - prev;
    if (gap <= 0) return;
    if (gap <= GAP_INLINE) {
        for (int64_t n = prev + 1; n <= cur; ++n) rowptr[n] = (int32_t)i;
    } else {
        const unsigned int slot = atomicAdd(gap_count, 1u);
        gap_list[3 * (int64_t)slot + 0] = (int32_t)(prev + 1);
        gap_list[3 * (int64_t)slot + 1] = (int32_t)cur;
        gap_list[3 * (int64_t)slot + 2] = (int32_t)i;
    }
}

// Four boundaries per thread (one 16-B load of u32 keys plus the key before them).
template <typename KeyT>
__global__ void rowptr_kernel(const KeyT* __restrict__ sorted_keys, int64_t E, int64_t N,
                              int32_t* __restrict__ rowptr, int32_t* __restrict__ gap_list,
                              unsigned int* __restrict__ gap_count) {
    const int64_t nquads = (E + 1 + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i0 = q * 4;
        int64_t k[5];  // k[j] = key[i0 - 1 + j]
        k[0] = (i0 == 0) ? -1 : (int64_t)sorted_keys[i0 - 1];
        if (sizeof(KeyT) == 4 && i0 + 3 < E && ((uintptr_t)sorted_keys % 16 == 0)) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(sorted_keys + i0);
            k[1] = v.x; k[2] = v.y; k[3] = v.z; k[4] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) k[1 + j] = (i0 + j < E) ? (int64_t)sorted_keys[i0 + j] : N;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j <= E) rowptr_boundary(i0 + j, k[j], (i0 + j == E) ? N : k[1 + j], N, rowptr, gap_list, gap_count);
    }
}

__global__ void fill_gaps_kernel(int32_t* __restrict__ rowptr, const int32_t* __restrict__ gap_list,
                                 const unsigned int* __restrict__ gap_count) {
    unsigned int cnt = *gap_count;
    for (unsigned int g = blockIdx.x; g < cnt; g += gridDim.x) {
        int64_t lo = gap_list[3 * (int64_t)g + 0], hi = gap_list[3 * (int64_t)g + 1];
        int32_t v = gap_list[3 * (int64_t)g + 2];
        for (int64_t n = lo + threadIdx.x; n <= hi; n += blockDim.x) rowptr[n] = v;
    }
}


// ---- small inputs: the whole plan in ONE launch -----------------------------------------------------------------------
// The radix build above is ten launches (memset, 2 x (hist, scan, scatter), rowptr, fill_gaps + the caller's permute of the
// companion column): ~45 us of kernels but ~90 us of launch overhead, which is all a batch of small graphs costs
// (app_bm/benchmark_convs.py: 512 QM9 molecules = 9 134 nodes, 18 744 edges). Up to 65 536 positions and 40 000
// destinations ONE workgroup does a counting sort with the counters in LDS:
//   count (LDS atomics) -> exclusive scan = rowptr -> placement in chunks of 1024 positions, in order.
// Stable without a sort: before a chunk claims its slots every position reads where its destination's run stands (prev);
// the claims (LDS atomicAdd) hand out the slots [prev, prev + k) of the chunk's k positions with that destination in SOME
// order; a position's stable slot is prev + (how many of those k are smaller than it), counted by reading the k claimed
// entries back — k is 1-2 on real graphs, and earlier chunks hold only smaller positions, later ones only larger.
// `companion` (optional): a second int64 array in position order (the edge list's source row); col[slot] = companion[e]
// comes out of the same launch, so the CSR column array needs no permute pass.
constexpr int SMALL_THREADS = 1024;
constexpr int64_t SMALL_MAX_E = 24576, SMALL_MAX_N = 40000;   // (N + 1) counters of 4 B (+ the scan scratch) in 160 KiB of LDS

__global__ __launch_bounds__(SMALL_THREADS) void plan_small_kernel(const int64_t* __restrict__ index,
                                                                   const int64_t* __restrict__ companion, int E, int N,
                                                                   int32_t* __restrict__ rowptr, int32_t* __restrict__ perm,
                                                                   int64_t* __restrict__ col) {
    extern __shared__ uint32_t s_cur[];   // [N + 1]
    __shared__ uint32_t s_tmp[SMALL_THREADS / 64];
    const int t = threadIdx.x;
    for (int n = t; n <= N; n += SMALL_THREADS) s_cur[n] = 0u;
    __syncthreads();
    for (int e = t; e < E; e += SMALL_THREADS) {
        const int64_t d = index[e];
        if (d >= 0 && d < N) atomicAdd(&s_cur[d], 1u);
    }
    __syncthreads();
    // exclusive scan over [0, N]: a contiguous run per thread, block scan of the run sums
    const int per = (N + 1 + SMALL_THREADS - 1) / SMALL_THREADS;
    const int lo = min(t * per, N + 1), hi = min(lo + per, N + 1);
    uint32_t run = 0;
    for (int n = lo; n < hi; ++n) run += s_cur[n];
    uint32_t off = block_excl_scan_u32<SMALL_THREADS / 64>(run, s_tmp, nullptr);
    for (int n = lo; n < hi; ++n) {
        const uint32_t c = s_cur[n];
        s_cur[n] = off;
        rowptr[n] = (int32_t)off;
        off += c;
    }
    __syncthreads();
    // per chunk of 1024 positions, two barriers:  [claim c]  |  [rank c, read where chunk c+1 starts]  |  [place c, claim c+1] ...
    auto fetch = [&](int e, int64_t& d, int64_t& comp) {
        d = -1; comp = 0;
        if (e < E) {
            d = index[e];
            if (d < 0 || d >= N) d = -1;   // out of range: belongs to no destination (the radix build sorts these past rowptr[N])
            if (companion) comp = companion[e];
        }
    };
    int64_t d, comp, d_next = -1, comp_next = 0;
    fetch(t, d, comp);
    uint32_t prev = d >= 0 ? s_cur[d] : 0u;
    __syncthreads();
    for (int base = 0; base < E; base += SMALL_THREADS) {
        const int e = base + t;
        if (d >= 0) perm[atomicAdd(&s_cur[d], 1u)] = e;                  // claim: slots [prev, prev + k) in some order
        const bool more = base + SMALL_THREADS < E;
        if (more) fetch(e + SMALL_THREADS, d_next, comp_next);          // the next chunk's keys travel under this chunk's barriers
        __syncthreads();                                                 // the chunk's claims are in `perm` and in s_cur
        uint32_t rank = 0;
        if (d >= 0) {
            const uint32_t end = s_cur[d];
            // read at device scope (past this CU's L1: the line may have been cached before a neighbouring slot was claimed)
            for (uint32_t j = prev; j < end; ++j)
                rank += (__hip_atomic_load(&perm[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) ? 1u : 0u;
        }
        const uint32_t prev_next = (more && d_next >= 0) ? s_cur[d_next] : 0u;   // s_cur is final for this chunk
        __syncthreads();                                                 // everyone has read the unordered entries and s_cur
        if (d >= 0) {
            perm[prev + rank] = e;
            if (col) col[prev + rank] = comp;
        }
        d = d_next; comp = comp_next; prev = prev_next;
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int key_bits_for(int64_t N) {
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < N) ++bits;
    return bits;
}

}  // namespace

extern "C" int gnnops_index_max(const int64_t* index, int64_t E, int64_t* d_max, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(d_max != nullptr, GNNOPS_EINVAL, "index_max: d_max is null");
    GNNOPS_REQUIRE(E >= 0 && (E == 0 || index != nullptr), GNNOPS_EINVAL, "index_max: bad index/E");
    // d_max = -1 (all bits set), then atomicMax over the data (stream-ordered, no sync)
    if (gnnops_memset_async(d_max, 0xff, sizeof(int64_t), stream) != hipSuccess)
        return gnnops_check_launch("index_max memset");
    if (E > 0) {
        int grid = gnnops_grid_cap(gnnops_cdiv(E, 256 * 8), 256);  // one workgroup per CU: 55 us against 75 us with 1024 or more (tools/time_index_max.py)
        if (const char* g = getenv("GNNOPS_IMAX_GRID")) grid = gnnops_grid_cap(gnnops_cdiv(E, 256 * 8), atoi(g));
        hipLaunchKernelGGL(index_max_kernel, dim3(grid), dim3(256), 0, stream, index, E, d_max);
    }
    return gnnops_check_launch("index_max");
}

// Workspace layout (all 256-B aligned):
//   keys_a[E] u32 | keys_b[E] u32 | vals_x[E] u32 | tile_hist[256*tiles] u32 | digit_total[256] u32 |
//   gap_count u32 | gap_list[3*(N/32+2)] i32
extern "C" size_t gnnops_plan_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return 0;
    size_t tiles = (size_t)gnnops_cdiv(E > 0 ? E : 1, sortengine::TILE);
    size_t b = 0;
    b += 3 * align_up((size_t)E * 4, 256);
    b += align_up(256 * tiles * 4, 256);
    b += 256 * 4;
    b += 256;
    b += align_up(3 * ((size_t)N / GAP_INLINE + 2) * 4, 256);
    return b;
}

extern "C" int gnnops_plan_build(const int64_t* index, int64_t E, int64_t N, int32_t* rowptr, int32_t* perm,
                                 void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(E >= 0 && N >= 0, GNNOPS_EINVAL, "plan_build: negative size");
    GNNOPS_REQUIRE(E < ((int64_t)1 << 31) && N < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED,
                   "plan_build: E and N must be < 2^31 (got E=%lld N=%lld)", (long long)E, (long long)N);
    GNNOPS_REQUIRE(rowptr != nullptr, GNNOPS_EINVAL, "plan_build: rowptr is null");
    if (E == 0) {
        if (gnnops_memset_async(rowptr, 0, (size_t)(N + 1) * 4, stream) != hipSuccess)
            return gnnops_check_launch("plan_build memset");
        return GNNOPS_OK;
    }
    GNNOPS_REQUIRE(index != nullptr && perm != nullptr, GNNOPS_EINVAL, "plan_build: null pointer");
    GNNOPS_REQUIRE(N > 0, GNNOPS_EINVAL, "plan_build: E > 0 needs N > 0");
    GNNOPS_REQUIRE(workspace_bytes >= gnnops_plan_workspace_bytes(E, N) && workspace != nullptr,
                   GNNOPS_EWORKSPACE, "plan_build: workspace %zu < %zu", workspace_bytes,
                   gnnops_plan_workspace_bytes(E, N));

    const size_t tiles = (size_t)gnnops_cdiv(E, sortengine::TILE);
    char* w = (char*)workspace;
    uint32_t* keys_a = (uint32_t*)w; w += align_up((size_t)E * 4, 256);
    uint32_t* keys_b = (uint32_t*)w; w += align_up((size_t)E * 4, 256);
    uint32_t* vals_x = (uint32_t*)w; w += align_up((size_t)E * 4, 256);
    uint32_t* tile_hist = (uint32_t*)w; w += align_up(256 * tiles * 4, 256);
    uint32_t* digit_total = (uint32_t*)w; w += 256 * 4;
    unsigned int* gap_count = (unsigned int*)w; w += 256;
    int32_t* gap_list = (int32_t*)w;

    const int bits = key_bits_for(N);
    const int passes = (bits + 7) / 8;

    // Ping-pong so that the last pass lands in (keys_?, perm).
    uint32_t* vals_y = (uint32_t*)perm;
    const uint32_t* kin = nullptr;
    const uint32_t* vin = nullptr;
    uint32_t* sorted_keys = nullptr;
    for (int p = 0; p < passes; ++p) {
        const bool to_y = ((passes - 1 - p) % 2) == 0;  // last pass writes perm
        uint32_t* kout = (p % 2 == 0) ? keys_a : keys_b;
        uint32_t* vout = to_y ? vals_y : vals_x;
        int rc;
        if (p == 0)
            rc = sortengine::pass_first_i64(index, kout, vout, E, 0, tile_hist, digit_total, (int)tiles, stream);
        else
            rc = sortengine::pass_u32(kin, vin, kout, vout, E, 8 * p, tile_hist, digit_total, (int)tiles, stream);
        if (rc != GNNOPS_OK) return rc;
        kin = kout; vin = vout; sorted_keys = kout;
    }

    if (gnnops_memset_async(gap_count, 0, sizeof(unsigned int), stream) != hipSuccess)
        return gnnops_check_launch("plan_build memset gap_count");
    {
        int grid = gnnops_grid_cap(gnnops_cdiv(E + 4, 1024));
        hipLaunchKernelGGL(rowptr_kernel<uint32_t>, dim3(grid), dim3(256), 0, stream, sorted_keys, E, N, rowptr, gap_list, gap_count);
        hipLaunchKernelGGL(fill_gaps_kernel, dim3(512), dim3(256), 0, stream, rowptr, gap_list, gap_count);
    }
    return gnnops_check_launch("plan_build");
}


// One-launch plan of a small index (see plan_small_kernel). gnnops_plan_small_fits says whether (E, N) qualify.
extern "C" int gnnops_plan_small_fits(int64_t E, int64_t N) {
    const char* sw = getenv("GNNOPS_PLAN_SMALL");   // A/B and tests: 0 = always the radix build
    if (sw && sw[0] == '0') return 0;
    return E > 0 && E <= SMALL_MAX_E && N > 0 && N <= SMALL_MAX_N;
}

extern "C" int gnnops_plan_build_small(const int64_t* index, const int64_t* companion, int64_t E, int64_t N, int32_t* rowptr,
                                       int32_t* perm, int64_t* col, gnnops_stream_t s) {
    GNNOPS_REQUIRE(gnnops_plan_small_fits(E, N), GNNOPS_EUNSUPPORTED, "plan_build_small: needs 0 < E <= %lld and 0 < N <= %lld",
                   (long long)SMALL_MAX_E, (long long)SMALL_MAX_N);
    GNNOPS_REQUIRE(index && rowptr && perm && (!companion == !col), GNNOPS_EINVAL, "plan_build_small: null pointer");
    const size_t lds = (size_t)(N + 1) * 4;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&plan_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((SMALL_MAX_N + 1) * 4)) != hipSuccess)
            return gnnops_check_launch("plan_build_small attribute");
        configured = true;
    }
    hipLaunchKernelGGL(plan_small_kernel, dim3(1), dim3(SMALL_THREADS), lds, (hipStream_t)s, index, companion, (int)E, (int)N,
                       rowptr, perm, col);
    return gnnops_check_launch("plan_build_small");
}

// rowptr of an already sorted int64 index (torch_scatter.segment_coo's input contract; also CSR <- sorted COO rows).
extern "C" size_t gnnops_rowptr_workspace_bytes(int64_t N) {
    if (N < 0) return 0;
    return 256 + align_up(3 * ((size_t)N / GAP_INLINE + 2) * 4, 256);
}

extern "C" int gnnops_rowptr_from_sorted(const int64_t* sorted_index, int64_t E, int64_t N, int32_t* rowptr,
                                         void* workspace, size_t workspace_bytes, gnnops_stream_t s) {
    hipStream_t stream = (hipStream_t)s;
    GNNOPS_REQUIRE(E >= 0 && N >= 0 && rowptr, GNNOPS_EINVAL, "rowptr_from_sorted: bad arguments");
    GNNOPS_REQUIRE(E < ((int64_t)1 << 31) && N < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "rowptr_from_sorted: E, N must be < 2^31");
    GNNOPS_REQUIRE(workspace && workspace_bytes >= gnnops_rowptr_workspace_bytes(N), GNNOPS_EWORKSPACE,
                   "rowptr_from_sorted: workspace too small");
    GNNOPS_REQUIRE(E == 0 || sorted_index, GNNOPS_EINVAL, "rowptr_from_sorted: null index");
    unsigned int* gap_count = (unsigned int*)workspace;
    int32_t* gap_list = (int32_t*)((char*)workspace + 256);
    if (gnnops_memset_async(gap_count, 0, sizeof(unsigned int), stream) != hipSuccess)
        return gnnops_check_launch("rowptr_from_sorted memset");
    const int grid = gnnops_grid_cap(gnnops_cdiv(E + 4, 1024));
    hipLaunchKernelGGL(rowptr_kernel<int64_t>, dim3(grid), dim3(256), 0, stream, sorted_index, E, N, rowptr, gap_list, gap_count);
    hipLaunchKernelGGL(fill_gaps_kernel, dim3(512), dim3(256), 0, stream, rowptr, gap_list, gap_count);
    return gnnops_check_launch("rowptr_from_sorted");
}
