/*
 * gnnops.h — C ABI of the MI355X (gfx950) operator library behind the gnn-ops-benchmark op API.
 *
 * This is the drop-in boundary. The reference (ryienh/gnn-ops-benchmark) has no FFI of its own:
 * its hot path is the body of each `op_*` function in op_bm_scripts/benchmark_*.py, which calls a
 * third-party CUDA op (torch_scatter / torch_sparse / ATen). Each entry point below replaces ONE such
 * call; the reference call site it stands behind is cited as (file:line under the reference root).
 *
 * Conventions
 *   - Plain pointers and sizes only. Every pointer is a DEVICE pointer unless marked "host".
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream). All work is enqueued on it;
 *     nothing here synchronises the device, allocates device memory or retains a pointer after return.
 *   - Tensors are viewed around the reduced/indexed dimension as [B, E, K] (source side) and [B, N, K]
 *     (destination side): B = product of sizes before `dim`, K = product of sizes after it.
 *     A 2-D [E, D] tensor indexed along dim 0 is B=1, K=D; along dim 1 it is B=E', K=1.
 *   - `index` is int64 as in the reference (benchmark_scatter_add.py:78-84). Layout R = one index per
 *     position along E (1-D row index, broadcast over B and K). Layout F = index of the full [B,E,K]
 *     shape (what the reference scripts build).
 *   - dtype codes: GNNOPS_F32/F16/BF16. 16-bit types are accumulated in fp32 and rounded once.
 *   - Return value: 0 on success, a GNNOPS_E* code otherwise; gnnops_last_error() gives the text
 *     (thread-local). Functions never throw and never call exit().
 */
#ifndef GNNOPS_H
#define GNNOPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNOPS_ABI_VERSION 2   /* 2: gnnops_narrow_index takes the id bound; round-2 exports (layers, spline, cluster) */

enum gnnops_status {
    GNNOPS_OK = 0,
    GNNOPS_EINVAL = 1,    /* bad argument (shape, dtype, alignment, null pointer) */
    GNNOPS_EWORKSPACE = 2,/* workspace too small */
    GNNOPS_ELAUNCH = 3,   /* hipGetLastError() after a launch */
    GNNOPS_EUNSUPPORTED = 4
};

enum gnnops_dtype { GNNOPS_F32 = 0, GNNOPS_F16 = 1, GNNOPS_BF16 = 2 };

/* torch_scatter reduce names: scatter_add/scatter_sum, scatter_mean, scatter_min, scatter_max, scatter_mul */
enum gnnops_reduce { GNNOPS_SUM = 0, GNNOPS_MEAN = 1, GNNOPS_MIN = 2, GNNOPS_MAX = 3, GNNOPS_MUL = 4 };

typedef void* gnnops_stream_t;

int gnnops_version(void);
const char* gnnops_last_error(void);

/* Measurement aid for bench.py's roofline leg (no reference counterpart): `reads` sequential nontemporal read streams of
 * `pieces` 16-B pieces each (src holds reads * pieces pieces) combined into ONE nontemporal write stream (dst: `pieces`
 * pieces, or NULL for reads only). The rate of the 5 : 1 mix is what the memory system offers the config-2 segment
 * reduction (5 source rows and the index per output row) on the box at hand. */
int gnnops_diag_stream_mix(const void* src, void* dst, int64_t pieces, int reads, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * dim_size discovery. torch_scatter computes `int(index.max()) + 1` when dim_size is None
 * (behind benchmark_scatter_add.py:18, benchmark_scatter_min.py:17). Writes max(index) (or -1 when
 * E == 0) to *d_max (device int64). The caller decides when to read it back.
 * ------------------------------------------------------------------------------------------- */
int gnnops_index_max(const int64_t* index, int64_t E, int64_t* d_max, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Inverted index ("plan") of a destination index: a stable counting sort of index[0..E) by value.
 *   rowptr[n] .. rowptr[n+1]  = range of `perm` holding the positions e with index[e] == n,
 *   in ascending e (stable) — so a per-destination sequential reduction visits contributions in the
 *   same order as a sequential CPU loop over e.
 * rowptr: int32[N+1], perm: int32[E] (E < 2^31). index values must lie in [0, N).
 * Shared by scatter_* / index_add_ (segment reduce), index_select (push form), spmm (COO->CSR),
 * coalesce and sparse transpose.
 * ------------------------------------------------------------------------------------------- */
size_t gnnops_plan_workspace_bytes(int64_t E, int64_t N);
int gnnops_plan_build(const int64_t* index, int64_t E, int64_t N,
                      int32_t* rowptr, int32_t* perm,
                      void* workspace, size_t workspace_bytes, gnnops_stream_t stream);
/* The same plan in ONE launch for small inputs (gnnops_plan_small_fits(E, N) != 0: E <= 24576 positions, N <= 40000
 * destinations — a batch of small graphs, app_bm/benchmark_convs.py): one workgroup, counters in LDS, stable. `companion`
 * (optional, int64 [E], e.g. the edge list's source row) comes back in plan order as col [E] from the same launch. */
int gnnops_plan_small_fits(int64_t E, int64_t N);
int gnnops_plan_build_small(const int64_t* index, const int64_t* companion, int64_t E, int64_t N, int32_t* rowptr,
                            int32_t* perm, int64_t* col, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Segment reduce over a plan — the kernel behind, with layout R:
 *   torch_scatter.scatter_add / scatter_mean / scatter_min / scatter_max / scatter_mul
 *     (benchmark_scatter_add.py:15-19, benchmark_scatter_mean.py:15-18,
 *      benchmark_scatter_min.py:15-18, benchmark_scatter_max.py:15-18), and
 *   Tensor.index_add_ (benchmark_native_index_add_.py:13-16) when init_from_out != 0.
 * src [B,E,K] -> out [B,N,K].  out[b,n,k] = reduce over e in segment n of src[b,e,k].
 *   init_from_out == 0 : out is overwritten; empty segments give 0 (SUM/MEAN/MIN/MAX) or 1 (MUL).
 *   init_from_out != 0 : the reduction starts from the current out[b,n,k] (index_add_, `out=` given);
 *                        MIN/MAX keep out where no contribution beats it; MEAN is not allowed.
 * arg_out (MIN/MAX only, may be NULL): int64 [B,N,K]; position e of the first extremal contribution,
 *   E where the segment is empty (torch_scatter's convention).
 * ------------------------------------------------------------------------------------------- */
int gnnops_segment_reduce(const void* src, const int32_t* rowptr, const int32_t* perm,
                          void* out, int64_t* arg_out,
                          int64_t B, int64_t E, int64_t K, int64_t N,
                          int dtype, int reduce, int init_from_out, gnnops_stream_t stream);
/* The same with heavy destinations ("hubs": more than 8192 contributions) set aside and reduced piecewise by whole
 * workgroups instead of serially by one lane group (csrc/hub.h; B == 1, rows of whole 16-B lanes). min / max stay exact;
 * sums / means / products of a hub are re-associated (deterministic, not bit-identical to the sequential loop).
 * hub_workspace: gnnops_hub_workspace_bytes(E, K, reduce) bytes, or NULL for the plain form. */
size_t gnnops_hub_workspace_bytes(int64_t E, int64_t K, int reduce);
int gnnops_segment_reduce_hubs(const void* src, const int32_t* rowptr, const int32_t* perm,
                               void* out, int64_t* arg_out,
                               int64_t B, int64_t E, int64_t K, int64_t N,
                               int dtype, int reduce, int init_from_out,
                               void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * One-shot row scatter — the same reference call sites as gnnops_segment_reduce (layout R, B == 1), for
 * a call that has no plan to reuse: src [E,K], index [E] int64 in [0,N), out [N,K] (arg_out as above).
 * Equivalent to gnnops_plan_build + gnnops_segment_reduce (same order of operations, bit-identical
 * results) but the last radix pass, the row-pointer kernel and the perm/rowptr round trip through HBM
 * are folded into the reduction (bucket.hip). Returns GNNOPS_EUNSUPPORTED — take the plan path — when
 * K * elem is not a whole number of 16-B lanes, pointers are not 16-B aligned, N <= 256, E == 0, or
 * E / N >= 2^31.
 * ------------------------------------------------------------------------------------------- */
int gnnops_scatter_rows_oneshot(const void* src, const int64_t* index, void* out, int64_t* arg_out,
                                int64_t E, int64_t K, int64_t N, int dtype, int reduce, int init_from_out,
                                void* workspace, size_t workspace_bytes, gnnops_stream_t stream);
/* The two stages separately: gnnops_bucket_partition groups the E positions by bucket = index >> 8 (stable) into the
 * workspace — which then is a reusable, coarser plan — and gnnops_bucket_reduce runs the reduction from it (the on-chip
 * finish of the sort is repeated per call; the indices never travel through HBM as rowptr / perm). */
size_t gnnops_bucket_workspace_bytes(int64_t E, int64_t N);
int gnnops_bucket_partition(const int64_t* index, int64_t E, int64_t N, void* workspace, size_t workspace_bytes,
                            gnnops_stream_t stream);
/* Windowed partition (the per-GPU step of the destination-partitioned scatter, BASELINE config 5): positions with
 * index[e] in [lo, lo+N) are partitioned under the local id index[e]-lo; the others are set aside in position order
 * behind the last bucket (pairs bptr[NB] .. E of the layout) and are ignored by reduce / select. */
int gnnops_bucket_partition_window(const int64_t* index, int64_t E, int64_t lo, int64_t N, void* workspace,
                                   size_t workspace_bytes, gnnops_stream_t stream);
/* Byte offsets inside a partitioned workspace: keys u32[E] (local id, sentinel ceil(N/256)*256 for set-aside positions),
 * positions u32[E], bptr int32[ceil(N/256)+1]. Host-only, no device work. */
int gnnops_bucket_layout(int64_t E, int64_t N, size_t* keys_offset, size_t* vals_offset, size_t* bptr_offset);
int gnnops_bucket_reduce(const void* src, const void* workspace, void* out, int64_t* arg_out,
                         int64_t E, int64_t K, int64_t N, int dtype, int reduce, int init_from_out,
                         gnnops_stream_t stream);
/* gnnops_bucket_reduce with hubs set aside, as gnnops_segment_reduce_hubs (hub_workspace: gnnops_hub_workspace_bytes). */
int gnnops_bucket_reduce_hubs(const void* src, const void* workspace, void* out, int64_t* arg_out,
                              int64_t E, int64_t K, int64_t N, int dtype, int reduce, int init_from_out,
                              void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t stream);
/* torch.index_select(input [N,K], 0, index [E]) -> out [E,K] (benchmark_native_index_select.py:14), push form, from a
 * workspace gnnops_bucket_partition filled for (index, E, N): each selected input row is read once and stored to every
 * output row that selects it. Rows must be a multiple of 16 bytes and 16-B aligned. */
int gnnops_bucket_select(const void* input, const void* workspace, void* out, int64_t N, int64_t K, int64_t E,
                         int elem_bytes, gnnops_stream_t stream);
/* Push-form index_select with hot rows (selected by more than 8192 outputs) set aside and written by whole workgroups
 * instead of one lane group (csrc/hub.h): hub_workspace = gnnops_hub_workspace_bytes(E, 0, 0) bytes, or NULL. */
int gnnops_bucket_select_hubs(const void* input, const void* workspace, void* out, int64_t N, int64_t K, int64_t E,
                              int elem_bytes, void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t stream);
int gnnops_index_select_planned_hubs(const void* input, const int32_t* rowptr, const int32_t* perm, void* out,
                                     int64_t B, int64_t N, int64_t K, int64_t E, int elem_bytes,
                                     void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Element-wise scatter, layout F (index has the shape of src) — what the reference scripts build
 * (benchmark_scatter_add.py:67,78-84; native form `zeros_like(src).scatter_add_(dim, idx, src)`
 * benchmark_scatter_add.py:22-25; `scatter_(-1, idx, src, reduce="multiply")`
 * benchmark_scatter_multiply.py:42-45).  out[b, index[b,e,k], k] (op)= src[b,e,k].
 *   init_from_out == 0 : the call initialises out itself (0 / identity) and, for MIN/MAX, sets groups
 *                        nothing reached to 0 (needs arg_out); MEAN divides by max(count, 1).
 *   init_from_out != 0 : contributions are combined into the existing out (`out=` / in-place forms).
 * SUM/MEAN on 16-bit types use an fp32 scratch of B*N*K floats in `workspace` (MEAN: one more for
 * the counts). MIN/MAX: value pass, then an arg pass choosing the smallest e among ties.
 * ------------------------------------------------------------------------------------------- */
size_t gnnops_scatter_elementwise_workspace_bytes(int64_t B, int64_t N, int64_t K, int dtype, int reduce);
int gnnops_scatter_elementwise(const void* src, const int64_t* index, void* out, int64_t* arg_out,
                               int64_t B, int64_t E, int64_t K, int64_t N,
                               int dtype, int reduce, int init_from_out,
                               void* workspace, size_t workspace_bytes, gnnops_stream_t stream);
/* The same with the index stored in `index_bytes` bytes per element: 8 (int64, what the reference builds), 4 (int32) or 2
 * (uint16, N <= 65536) — SURVEY.md 8(f) rank 2: at the reference's layout-F shapes (benchmark_scatter_add.py:78-84) the
 * int64 index is 8 of every 10 bytes the op reads; a narrowed copy (gnnops_narrow_index) is made once per index tensor and
 * reused. A narrowed index is taken by the LDS-strip form only: GNNOPS_EUNSUPPORTED otherwise (pass the int64 index). */
int gnnops_scatter_elementwise_ix(const void* src, const void* index, int index_bytes, void* out, int64_t* arg_out,
                                  int64_t B, int64_t E, int64_t K, int64_t N, int dtype, int reduce, int init_from_out,
                                  void* workspace, size_t workspace_bytes, gnnops_stream_t stream);
/* The same with the arg rows of min / max stored as int32 (arg_bytes 4; E < 2^31; the LDS-strip form only, GNNOPS_EUNSUPPORTED
 * otherwise) or int64 (arg_bytes 8 = gnnops_scatter_elementwise_ix): the dim-0 route of large full-index scatters
 * (data/scatter_max.csv:32-33, (38000, 38000)) reduces along the last dim of TRANSPOSED operands and widens the positions inside
 * the transpose back (gnnops_transpose2d_cvt), so neither the int64 index nor an int64 arg crosses a transpose. */
int gnnops_scatter_elementwise_ixa(const void* src, const void* index, int index_bytes, void* out, void* arg_out, int arg_bytes,
                                   int64_t B, int64_t E, int64_t K, int64_t N, int dtype, int reduce, int init_from_out,
                                   void* workspace, size_t workspace_bytes, gnnops_stream_t stream);
/* out[i] = (int32 / uint16) index[i], i < n, for ids in [0, bound); an id outside becomes all ones (-1 / 0xFFFF — never a valid
 * id: bound <= 65535 for out_bytes 2, < 2^31 for 4), which the element kernels drop as they drop it in the int64 index. */
int gnnops_narrow_index(const int64_t* index, void* out, int64_t n, int out_bytes, int64_t bound, gnnops_stream_t stream);

/* torch_scatter.scatter_min / scatter_max of a LONG 1-D tensor (benchmark_scatter_min.py:15-18 at the reference's ">= 95 % of
 * memory" shapes, data/scatter_min.csv:2: 1 472 353 280 fp32 elements): src [E], index [E] int64, out [N], arg_out [N] int64.
 * The value travels with its destination through radix passes over the destination bits above the low 15, and a workgroup
 * finishes each bucket of 32768 destinations in LDS (csrc/scatter1d.hip) — no complete sort, no random gather. Same result
 * as gnnops_segment_reduce (out 0 / arg E where nothing arrives, smallest position on ties, NaNs and the reduce's identity
 * never win). GNNOPS_EUNSUPPORTED unless 32768 < N < 2^31 - 32768 and 0 < E < 2^31 and reduce is MIN or MAX. */
size_t gnnops_scatter1d_workspace_bytes(int64_t E, int64_t N);
int gnnops_scatter1d_minmax(const void* src, const int64_t* index, void* out, int64_t* arg_out, int64_t E, int64_t N,
                            int dtype, int reduce, void* workspace, size_t workspace_bytes, gnnops_stream_t stream);
/* The same shapes for reduce SUM / MEAN / MUL (benchmark_scatter_mean.py:15-18 at data/scatter_mean.csv:2): three passes to buckets
 * of 256 destinations, finished by a stable on-chip sort and a sum in source position order — bit-identical to the sequential
 * loop (16-bit types: fp32 accumulator, rounded once). Same workspace and shape limits as gnnops_scatter1d_minmax. */
int gnnops_scatter1d_sum(const void* src, const int64_t* index, void* out, int64_t E, int64_t N, int dtype, int reduce,
                         void* workspace, size_t workspace_bytes, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * torch.index_select (benchmark_native_index_select.py:12-15; also the first half of
 * benchmark_fused_index_select_reduce.py:12-20).  input [B,N,K], index int64 [E] -> out [B,E,K],
 * out[b,e,k] = input[b,index[e],k].  Bit-exact copy of opaque elements; `elem_bytes` in {1,2,4,8}.
 * ------------------------------------------------------------------------------------------- */
int gnnops_index_select(const void* input, const int64_t* index, void* out,
                        int64_t B, int64_t N, int64_t K, int64_t E,
                        int elem_bytes, gnnops_stream_t stream);

/* Push form of the same op over a plan of `index` (each input row is read once and stored to every
 * output row that selects it). Same result, different HBM traffic; see DESIGN.md. */
int gnnops_index_select_planned(const void* input, const int32_t* rowptr, const int32_t* perm, void* out,
                                int64_t B, int64_t N, int64_t K, int64_t E,
                                int elem_bytes, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * torch.gather (benchmark_native_gather.py:14-17). input [B,N,K], index int64 [B,E,K] ->
 * out[b,e,k] = input[b, index[b,e,k], k]. Bit-exact copy.
 * ------------------------------------------------------------------------------------------- */
int gnnops_gather(const void* input, const int64_t* index, void* out,
                  int64_t B, int64_t N, int64_t K, int64_t E,
                  int elem_bytes, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused index_select + full sum (benchmark_fused_index_select_reduce.py:12-20):
 *   *d_sum_f32 = sum_{b,e,k} input[b,index[e],k], accumulated in fp32 (the reference's fp16 result
 *   overflows to inf at its own sizes; the fp32 accumulator is the comparable quantity).
 * The [B,E,K] intermediate is never written. Deterministic: per-block partials are combined in a
 * fixed order by a second launch. workspace: gnnops_fused_select_sum_workspace_bytes().
 * ------------------------------------------------------------------------------------------- */
size_t gnnops_fused_select_sum_workspace_bytes(void);
int gnnops_fused_index_select_sum(const void* input, const int64_t* index, float* d_sum_f32,
                                  int64_t B, int64_t N, int64_t K, int64_t E,
                                  int dtype, void* workspace, size_t workspace_bytes,
                                  gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * torch_sparse.spmm(index, value, m, n, matrix) / torch.sparse.mm(COO, dense)
 * (benchmark_sparse_spmm.py:12-14; BASELINE config 3). Row-split over a CSR view of the sparse operand:
 *   out[i,:] = sum_{j in [rowptr[i], rowptr[i+1])} value[e_j] * mat[col[e_j], :],  e_j = perm ? perm[j] : j
 * so a CSR matrix passes perm = NULL and a COO matrix passes the plan of its row index (rowptr, perm).
 * value == NULL means all ones. mat [mat_rows, D], out [M, D]; fp32 accumulation, one rounding.
 * ------------------------------------------------------------------------------------------- */
int gnnops_spmm(const int32_t* rowptr, const int32_t* perm, const int64_t* col, const void* value,
                const void* mat, void* out, int64_t M, int64_t D, int64_t nnz, int64_t mat_rows, int dtype,
                gnnops_stream_t stream);
/* The same with rows of more than 8192 nonzeros set aside and multiplied out piecewise by whole workgroups (csrc/hub.h;
 * re-associated fp32 sums for those rows only): hub_workspace = gnnops_hub_workspace_bytes(nnz, D, 0) bytes, or NULL. */
int gnnops_spmm_hubs(const int32_t* rowptr, const int32_t* perm, const int64_t* col, const void* value,
                     const void* mat, void* out, int64_t M, int64_t D, int64_t nnz, int64_t mat_rows, int dtype,
                     void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t stream);

/* out[j] = in[perm[j]] (elements of 2, 4 or 8 bytes): materialises the CSR column / value arrays of a plan-ordered COO
 * operand once, so gnnops_spmm can be called with perm == NULL and stream them. */
int gnnops_permute(const void* in, const int32_t* perm, void* out, int64_t n, int elem_bytes, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * torch.sort(input, dim, descending, stable) (benchmark_native_sort.py:28-30 times fp32). input viewed
 * [B,E,K], sorted along E; values [B,E,K] of the input dtype, indices int64 [B,E,K] (position along E).
 * Always stable. sort_dtype: 0 f32, 1 f16, 2 bf16, 3 i32, 4 i64, 5 f64 (64-bit types: B*K == 1 only).
 * -0.0 is ordered (and returned) as +0.0; NaNs last (first when descending).
 * ------------------------------------------------------------------------------------------- */
size_t gnnops_sort_workspace_bytes(int64_t B, int64_t E, int64_t K, int sort_dtype);
int gnnops_sort(const void* input, void* values, int64_t* indices, int64_t B, int64_t E, int64_t K,
                int sort_dtype, int descending, void* workspace, size_t workspace_bytes, gnnops_stream_t stream);

/* On-chip form of the same op for fp32 rows that fit in LDS (E <= gnnops_sort_rows_max_len()): input / values
 * [rows, E], indices int64 [rows, E], sorted along E; one HBM read and one write per element instead of 6+ passes. */
int64_t gnnops_sort_rows_max_len(void);
int gnnops_sort_rows_f32(const float* input, float* values, int64_t* indices, int64_t rows, int64_t E,
                         int descending, gnnops_stream_t stream);
int64_t gnnops_sort_rows2_max_len(void);
/* Rows longer than gnnops_sort_rows_max_len(), up to gnnops_sort_rows2_max_len() (the reference's (28200, 28200) sort): halves sorted on chip into the caller's
 * (tmp_values, tmp_indices) — each [rows, E] — then rank-merged into (values, indices). Same ordering conventions. */
int gnnops_sort_rows2_f32(const float* input, float* values, int64_t* indices, float* tmp_values, int64_t* tmp_indices,
                          int64_t rows, int64_t E, int descending, gnnops_stream_t stream);
/* Both with the positions as int32 rows (they fit: a row is at most gnnops_sort_rows2_max_len() long), for a caller that widens
 * them in a later pass of its own (torch.sort along dim 0 of a matrix: gnnops_transpose2d_cvt mode 1 on the way back). */
int gnnops_sort_rows_f32_i32(const float* input, float* values, int32_t* indices, int64_t rows, int64_t E,
                             int descending, gnnops_stream_t stream);
int gnnops_sort_rows2_f32_i32(const float* input, float* values, int32_t* indices, float* tmp_values, int32_t* tmp_indices,
                              int64_t rows, int64_t E, int descending, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * torch_sparse.coalesce(index, value, m, n, op="add") / Tensor.coalesce()
 * (benchmark_sparse_coalesce.py:35-42); torch_sparse.transpose = the same call with row/col swapped
 * and (n, m). Entries are sorted row-major by (row, col); duplicates are summed in sorted (stable)
 * order. value [nnz, C] of `dtype` or NULL. Outputs are sized nnz; *d_count (device int64) receives
 * the number of distinct entries.
 * ------------------------------------------------------------------------------------------- */
size_t gnnops_coalesce_workspace_bytes(int64_t nnz);
int gnnops_coalesce(const int64_t* row, const int64_t* col, const void* value, int64_t nnz,
                    int64_t m, int64_t n, int64_t C, int dtype,
                    int64_t* out_row, int64_t* out_col, void* out_value, int64_t* d_count,
                    void* workspace, size_t workspace_bytes, gnnops_stream_t stream);

/* Dense 2-D transpose copy: torch.transpose(matA, 0, 1).contiguous() (benchmark_sparse_transpose.py:13-16).
 * in [R, C] -> out [C, R]; elem_bytes in {1, 2, 4, 8}; bit-exact. */
int gnnops_transpose2d(const void* in, void* out, int64_t R, int64_t C, int elem_bytes,
                       gnnops_stream_t stream);
/* [R, C] int64 -> [C, R] int32 (mode 0: every value must fit in 31 bits) or [R, C] int32 -> [C, R] int64 (mode 1). */
int gnnops_transpose2d_cvt(const void* in, void* out, int64_t R, int64_t C, int mode, gnnops_stream_t stream);
/* mode 0 with the largest element read left in *max_out (-1 for an empty matrix; device memory, written on the stream):
 * torch_scatter's implicit dim_size = index.max() + 1 from the index's own transpose. Values >= 2^31 make the int32 copy
 * meaningless — the caller checks *max_out before using it. */
int gnnops_transpose2d_cvt_max(const void* in, void* out, int64_t R, int64_t C, int64_t* max_out, gnnops_stream_t stream);
/* batch of independent [R, C] -> [C, R] copies (in / out [batch, R, C] / [batch, C, R]). */
int gnnops_transpose_batched(const void* in, void* out, int64_t batch, int64_t R, int64_t C, int elem_bytes,
                             gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * torch.addmm(input, mat1, mat2) / torch.matmul(input, other) on 16-bit operands
 * (benchmark_native_addmm.py:13-16, benchmark_native_matmul.py:13-16): out[M,N] = input[M,N] + mat1[M,K] @
 * mat2[K,N] (input == NULL: plain matmul). Row-major; dtype F16 / BF16 (fp32 MFMA accumulation, one rounding) or
 * F32 (exact-fp32 MFMA).
 * ------------------------------------------------------------------------------------------- */
/* Workspace of the 16-bit product (fp32 needs none; passing it is harmless): zero-filled copies of the last K-tile of each
 * operand when K is not a multiple of 64 or N not a multiple of 8 (a whole padded copy of a large mat1 instead), plus — when more than one round of 256 x 256 tiles leaves a last round of
 * at most half the CUs' worth — one 256 KiB fp32 slot per CU and a flag word each for the split-K tail (64 MiB on MI355X;
 * contents need not survive the call, the flags are cleared on the stream by the call itself). 0 for small aligned problems. */
size_t gnnops_addmm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gnnops_addmm(const void* input, const void* mat1, const void* mat2, void* out,
                 int64_t M, int64_t N, int64_t K, int dtype,
                 void* workspace, size_t workspace_bytes, gnnops_stream_t stream);
/* The same with a row pitch for `input` (elements): ld_input = N is gnnops_addmm; ld_input = 0 adds ONE row [N] to every
 * output row — the bias of a Linear layer (app_bm/groq_script.py:75-76 lin_f / lin_s) without materialising [M, N]. Any other
 * pitch must be >= N and a multiple of 16 bytes (EINVAL otherwise). */
int gnnops_addmm_ld(const void* input, int64_t ld_input, const void* mat1, const void* mat2, void* out, int64_t M,
                    int64_t N, int64_t K, int dtype, void* workspace, size_t workspace_bytes, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused index_select(index_add(input, dim, index, other), dim, index).sum(dim)
 * (benchmark_fused_index_add_reduce.py:12-20). input [B,N,K], other [B,E,K], plan of index over N.
 * out_f32 [B,K] (fp32: the reference's fp16 result overflows at its own sizes). Nothing of size [B,N,K]
 * is materialised. workspace only when K > 1.
 * ------------------------------------------------------------------------------------------- */
size_t gnnops_fused_index_add_select_sum_workspace_bytes(int64_t B, int64_t K);
int gnnops_fused_index_add_select_sum(const void* input, const void* other,
                                      const int32_t* rowptr, const int32_t* perm, float* out_f32,
                                      int64_t B, int64_t N, int64_t E, int64_t K, int dtype,
                                      void* workspace, size_t workspace_bytes, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * torch_sparse.spspmm / torch.sparse.mm(COO, COO) (benchmark_sparse_spspmm.py:12-14,94-95):
 * C = A(m x k) @ B(k x n) by expand - sort - compress. B is given as the plan (rowptrB, permB) of its row
 * index plus colB/valB. Phase 1 counts the products (*d_total, device int64) so the caller can size the
 * expansion; phase 2 writes them in (nonzero of A, nonzero of B's row) order; gnnops_coalesce over
 * (out_row, out_col, out_val, m, n) then yields the coalesced result. `workspace` is shared by both phases.
 * ------------------------------------------------------------------------------------------- */
size_t gnnops_spspmm_workspace_bytes(int64_t nnzA);
int gnnops_spspmm_count(const int64_t* colA, int64_t nnzA, const int32_t* rowptrB, int64_t* d_total,
                        void* workspace, size_t workspace_bytes, gnnops_stream_t stream);
int gnnops_spspmm_expand(const int64_t* rowA, const int64_t* colA, const void* valA, int64_t nnzA,
                         const int32_t* rowptrB, const int32_t* permB, const int64_t* colB, const void* valB,
                         int64_t* out_row, int64_t* out_col, void* out_val, int dtype,
                         const void* workspace, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Widening per SURVEY.md 8(f) rank 1: what PyG layers on the reference's OpProfiler path call.
 *
 * gnnops_segment_reduce with perm == NULL treats rowptr as a CSR pointer over src itself
 * (torch_scatter.segment_csr); gnnops_rowptr_from_sorted turns a sorted int64 index into that pointer
 * (torch_scatter.segment_coo's input contract).
 *
 * gnnops_segment_composite: torch_scatter.composite over a plan / CSR pointer (perm may be NULL).
 *   mode 0 softmax      out[B,E,K] = exp(x - max) / sum exp(x - max)
 *   mode 1 log_softmax  out[B,E,K] = (x - max) - log(sum + param)         param = eps (upstream 1e-12)
 *   mode 2 logsumexp    out[B,N,K] = max + log(sum + param)               param = eps
 *   mode 3 std          out[B,N,K] = sqrt(sum (x - mean)^2 / (cnt' + 1e-6)), param != 0: unbiased (cnt' = max(cnt-1,1))
 * ------------------------------------------------------------------------------------------- */
/* Backward-pass pieces (gnnops/autograd.py; the rest of every backward is one of the forward entry points above):
 *   gnnops_rowptr_expand  index[e] = n with rowptr[n] <= e < rowptr[n+1], or N where no segment holds e (E positions):
 *                         torch_scatter.gather_csr's addressing and the gather behind segment_csr's gradient.
 *   gnnops_sddmm          out[k] = sum_d a[rows_a[k], d] * b[rows_b[k], d]: d(value) of torch_sparse.spmm
 *                         (benchmark_sparse_spmm.py:12-14 under autograd); fp32 accumulation, one rounding. */
int gnnops_rowptr_expand(const int32_t* rowptr, int64_t N, int64_t E, int64_t* index, gnnops_stream_t stream);
int gnnops_sddmm(const int64_t* rows_a, const int64_t* rows_b, const void* a, const void* b, void* out,
                 int64_t nnz, int64_t D, int dtype, gnnops_stream_t stream);
/* Destination-partitioned scatter over the GPUs of a node (BASELINE config 5; gnnops/dist.py): counts[g] = number of
 * positions e with g * rows_per_owner <= index[e] < (g + 1) * rows_per_owner, g < owners <= 64 — what sizes the one
 * exchange of the step (the only value the host reads back). counts: device int64[owners], zeroed here. An id outside
 * [0, owners * rows_per_owner) is counted for no owner, so sum(counts) < E tells the caller about it. */
int gnnops_owner_counts(const int64_t* index, int64_t E, int64_t rows_per_owner, int owners, int64_t* counts,
                        gnnops_stream_t stream);
size_t gnnops_rowptr_workspace_bytes(int64_t N);
int gnnops_rowptr_from_sorted(const int64_t* sorted_index, int64_t E, int64_t N, int32_t* rowptr,
                              void* workspace, size_t workspace_bytes, gnnops_stream_t stream);
int gnnops_segment_composite(const void* src, const int32_t* rowptr, const int32_t* perm, void* out,
                             int64_t B, int64_t E, int64_t K, int64_t N, int dtype, int mode, double param,
                             gnnops_stream_t stream);
/* The same with groups of more than 8192 members set aside and processed piecewise by whole workgroups (csrc/hub.h);
 * hub_workspace = gnnops_hub_workspace_bytes(E, K, GNNOPS_MIN) bytes, or NULL. B == 1, rows of whole 16-B lanes. */
int gnnops_segment_composite_hubs(const void* src, const int32_t* rowptr, const int32_t* perm, void* out, int64_t B,
                                  int64_t E, int64_t K, int64_t N, int dtype, int mode, double param,
                                  void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Widening per SURVEY.md 8(f) rank 4: the message + aggregate of one message-passing layer in one pass
 * (app_bm/benchmark_convs.py:146-246 times FiLMConv, GINConv, CGConv, PNAConv, SAGEConv forward;
 * app_bm/groq_script.py:91-109 is CGConv.forward / .message). Replaces, per layer, the chain
 * x[edge_index[0]] / x[edge_index[1]] -> message -> torch_scatter.scatter(..., reduce) of MessagePassing.propagate.
 *
 *   out[i, (s * n_aggr + a) * K + k] = scaler_s(deg_i) * AGGR_a over the edges (j -> i) of f(p[i], q[j], w[e])[k]
 *
 * Edges in destination-sorted (plan) order: rowptr int32 [N+1]; col int64 [E] = source row j of sorted edge position;
 * perm int32 [E] = original edge id of that position (only read when w is given; NULL = identity).
 * functor 0 COPY    f = q                                             rows: q [K]
 *         1 ADD     f = p + q + w   (w optional)                            q, p, w [K]
 *         2 CGCONV  f = sigmoid(p_f + q_f + w_f) * softplus(p_s + q_s + w_s) q, p, w [f part K | s part K]; w optional
 *         3 FILM    f = relu(gamma * q + beta)                               q [K], p [beta K | gamma K]
 * aggr[]: 0 sum, 1 mean (divide by max(deg, 1)), 2 min, 3 max (0 for rows without edges), 4 std =
 *   sqrt(relu(mean(f^2) - mean(f)^2) + 1e-5) — PNAConv.aggregate; scalers[] (n_scalers == 0: none): 0 identity,
 *   1 amplification log(deg+1)/avg_deg_log, 2 attenuation avg_deg_log/log(deg+1), 3 linear deg/avg_deg_lin,
 *   4 inverse_linear avg_deg_lin/deg, deg clamped to >= 1 — PNAConv's degree scalers.
 * add (optional, [N, >=K]): added to the first output block (CGConv's `out += x[1]`).
 * ld* = row pitches in elements, so operands may be column blocks of one GEMM result and `out` a column block of the
 * buffer the layer would torch.cat into. fp32 arithmetic, one rounding on store.
 * ------------------------------------------------------------------------------------------- */
int gnnops_edge_reduce(int functor, const void* q, int64_t ldq, const void* p, int64_t ldp, const void* w, int64_t ldw,
                       const void* add, int64_t ldadd, const int32_t* rowptr, const int32_t* perm, const int64_t* col,
                       void* out, int64_t ldo, int64_t N, int64_t E, int64_t K, const int* aggr, int n_aggr,
                       const int* scalers, int n_scalers, float avg_deg_log, float avg_deg_lin, int dtype,
                       gnnops_stream_t stream);
/* The same with destinations of more than 8192 edges ("hubs", as csrc/hub.h) set aside by the main kernel and reduced in
 * pieces of 2048 edges by separate lane groups (fp32 partial accumulators, combined in piece order: the same result every
 * run, sums re-associated for those rows only). hub_workspace = gnnops_edge_reduce_hub_workspace_bytes(E, K) bytes, or NULL. */
size_t gnnops_edge_reduce_hub_workspace_bytes(int64_t E, int64_t K);
int gnnops_edge_reduce_hubs(int functor, const void* q, int64_t ldq, const void* p, int64_t ldp, const void* w, int64_t ldw,
                            const void* add, int64_t ldadd, const int32_t* rowptr, const int32_t* perm, const int64_t* col,
                            void* out, int64_t ldo, int64_t N, int64_t E, int64_t K, const int* aggr, int n_aggr,
                            const int* scalers, int n_scalers, float avg_deg_log, float avg_deg_lin, int dtype,
                            void* hub_workspace, size_t hub_workspace_bytes, gnnops_stream_t stream);

/* Backward of the edge pass for sum / mean aggregation — what lets GINConv / SAGEConv / CGConv / FiLMConv of gnnops.conv sit
 * in a training loop (graph_benchmark/profile/OpProfiler.py:259-292 profiles one). Per edge e = (j -> i), in EDGE order,
 * the gradient of the message with respect to the rows it was made from, given g[i] = d loss / d out[i] (for mean: already
 * divided by max(deg_i, 1)):
 *   CGCONV  z_f = p_f[i] + q_f[j] (+ w_f[e]), z_s likewise:
 *           gp[e] = [ g * softplus(z_s) * sig(z_f) (1 - sig(z_f)) | g * sig(z_f) * sig(z_s) ]   (2K; gq: pass NULL — d q and d w
 *           are the same rows)        d p[i] = sum of gp over the edges into i, d q[j] = sum over the edges out of j, d w = gp
 *   FILM    a = gamma[i] * q[j] + beta[i], m = [a > 0]:  gp[e] = [ g m | g m q[j] ]  (2K = d beta | d gamma),  gq[e] = g m gamma[i]  (K)
 * COPY / ADD messages need no call (their per-edge gradient is g[i] itself). The two sums are gnnops_segment_reduce over the
 * destination plan and over the plan of the source ids. ld* = row pitches in elements; gp [E, 2K], gq [E, K] dense. */
int gnnops_edge_grad(int functor, const void* p, int64_t ldp, const void* q, int64_t ldq, const void* w, int64_t ldw,
                     const void* g, int64_t ldg, const int64_t* src, const int64_t* dst, void* gp, void* gq, int64_t E,
                     int64_t K, int dtype, gnnops_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The remaining ops of the reference's list (ops.txt:17-19, 29-41) - SURVEY.md 8(f) rank 4. Neither package is in the
 * reference tree (torch-spline-conv 1.2.1, torch-cluster 1.5.9: requirements.txt:214, :210) and the reference has no
 * script or output for them: the packages' published definitions, parity unpinned (oracle/spatial_oracle.py).
 *
 * torch_spline_conv  (csrc/spline.hip has the formulas)
 *   gnnops_spline_basis      pseudo [E, D] -> basis [E, S] (dtype), weight_index [E, S] int64, S = (degree + 1)^D <= 64;
 *                            kernel_size int64 [D] and is_open_spline uint8 [D] are HOST arrays
 *   gnnops_spline_weighting  out[e, o] = sum_s basis[e, s] * sum_i x[e, i] * weight[weight_index[e, s], i, o]
 *   gnnops_spline_conv       the whole layer per destination row in one pass: (rowptr, perm) = plan of edge_index[0],
 *                            src int64 [E] = edge_index[1] in plan order, pseudo [E, D] in edge order; norm != 0 divides by
 *                            the row's degree; root_weight [Min, Mout] and bias [Mout] optional
 * torch_cluster  (batches as CSR pointers ptr [batches + 1] int64 over points sorted by batch; all arrays on the device)
 *   gnnops_grid_cluster  voxel id per point; size / start / end: device double [D]
 *   gnnops_knn           col [Ny, k] = the k nearest x of every y within its batch, ascending (distance, index); -1 = none
 *   gnnops_radius        col [Ny, max] = the first max x (ascending index) with |x - y|^2 < r^2; -1 = none
 *   gnnops_fps           farthest point sampling per batch: out[out_ptr[b] .. out_ptr[b+1]) starting from start[b];
 *                        dist_workspace float [N]
 *   gnnops_random_walk   out [walkers, walk_length + 1]: uniform next neighbour over a CSR adjacency (int64 rowptr, col)
 *   gnnops_graclus_rounds  `rounds` rounds of handshake matching over a CSR adjacency (weight in CSR order or NULL): cluster
 *                        int64 [N] (-1 = unmatched, initialised by the caller) gets min(n, partner) for matched pairs;
 *                        *d_active (device int) = proposals in the last round (0: the matching is maximal); finish != 0
 *                        gives every node still alone its own id. proposal: int64 [N] scratch.
 * ------------------------------------------------------------------------------------------- */
int gnnops_spline_basis(const void* pseudo, const int64_t* kernel_size, const uint8_t* is_open_spline, int64_t E, int D,
                        int degree, void* basis, int64_t* weight_index, int dtype, gnnops_stream_t stream);
int gnnops_spline_weighting(const void* x, const void* weight, const void* basis, const int64_t* weight_index, void* out,
                            int64_t E, int64_t Min, int64_t Mout, int64_t S, int dtype, gnnops_stream_t stream);
int gnnops_spline_conv(const void* x, const int32_t* rowptr, const int32_t* perm, const int64_t* src, const void* pseudo,
                       const void* weight, const int64_t* kernel_size, const uint8_t* is_open_spline, int D, int degree,
                       const void* root_weight, const void* bias, void* out, int64_t N, int64_t E, int64_t Min, int64_t Mout,
                       int norm, int dtype, gnnops_stream_t stream);
int gnnops_grid_cluster(const void* pos, int64_t N, int D, const double* d_size, const double* d_start, const double* d_end,
                        int64_t* cluster, int dtype, gnnops_stream_t stream);
int gnnops_knn(const void* x, const void* y, const int64_t* ptr_x, const int64_t* ptr_y, int64_t batches, int64_t Ny, int D,
               int k, int cosine, int64_t* col, int dtype, gnnops_stream_t stream);
/* torch_cluster.knn through a uniform grid — one cloud, Euclidean, D <= 3, fp32, k <= 64; the pairs of the exhaustive kernel.
 * Step 1: bounding box (24 bytes at `box`) and cell ids of x on a G^D grid; then gnnops_plan_build(cell, Nx, G^D, rowptr, perm);
 * step 2: the queries walk the cells around their own in shells until the k-th distance is inside the block searched. */
int gnnops_knn_grid_cells(const void* x, int64_t Nx, int D, int G, void* box, int64_t* cell, gnnops_stream_t stream);
int gnnops_knn_grid_query(const void* x, const void* y, int64_t Ny, int D, int k, int G, const void* box, const int32_t* rowptr,
                          const int32_t* perm, int64_t* col, gnnops_stream_t stream);
/* torch_cluster.radius over the same cells / plan: the max_num_neighbors (<= 64) smallest indices inside the ball, ascending. */
int gnnops_radius_grid_query(const void* x, const void* y, int64_t Ny, int D, double r, int max_num_neighbors, int G, const void* box,
                             const int32_t* rowptr, const int32_t* perm, int64_t* col, gnnops_stream_t stream);
int gnnops_radius(const void* x, const void* y, const int64_t* ptr_x, const int64_t* ptr_y, int64_t batches, int64_t Ny, int D,
                  double r, int max_num_neighbors, int64_t* col, int dtype, gnnops_stream_t stream);
int gnnops_fps(const void* x, const int64_t* ptr, const int64_t* out_ptr, const int64_t* start, int64_t batches, int D,
               float* dist_workspace, int64_t* out, int dtype, gnnops_stream_t stream);
int gnnops_random_walk(const int64_t* rowptr, const int64_t* col, const int64_t* start, int64_t walkers, int walk_length,
                       uint64_t seed, int64_t* out, gnnops_stream_t stream);
/* torch_cluster.random_walk with node2vec's bias (p: return, q: in-out; ops.txt:41): after a uniform first step, a candidate drawn
 * uniformly from the current node's neighbours is accepted with probability (1/p, 1, 1/q) / max(1/p, 1, 1/q) when it is the
 * previous node / a neighbour of the previous node / neither (rejection sampling, as the package). Every row of the CSR
 * adjacency must be sorted ascending (the neighbour test is a binary search). */
int gnnops_random_walk_node2vec(const int64_t* rowptr, const int64_t* col, const int64_t* start, int64_t walkers, int walk_length,
                                double p, double q, uint64_t seed, int64_t* out, gnnops_stream_t stream);
int gnnops_graclus_rounds(const int64_t* rowptr, const int64_t* col, const void* weight, int64_t N, uint64_t seed, int rounds,
                          int64_t* cluster, int64_t* proposal, int* d_active, int finish, int dtype, gnnops_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GNNOPS_H */
