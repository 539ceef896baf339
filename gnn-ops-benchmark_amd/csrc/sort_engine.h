// sort_engine.h — host-side interface of the radix-sort pass (kernels: sort_engine_impl.h,
// compiled once in sort_engine.hip). One call = one stable 8-bit LSD pass over key/value pairs.
#pragma once
#include "common.h"

namespace sortengine {

constexpr int THREADS = 512;
constexpr int WAVES = THREADS / 64;
constexpr int ROUNDS = 16;
constexpr int TILE = WAVES * ROUNDS * 64;  // 8192 keys per workgroup
constexpr int RADIX = 256;

// Source descriptor of a pass that builds its 8-byte keys on the fly from a destination index and a value array
// (scatter1d.hip): key = (destination << 32) | fp32 bits of the value; a destination outside [0, n_dst) becomes `sentinel`.
// The descriptor lives in DEVICE memory (the kernels take a pointer to it, like any other key array).
template <typename T>
struct DstValSrc {
    const int64_t* idx;
    const T* val;
    int64_t n_dst;
    uint32_t sentinel;
};

// tile_hist: u32[RADIX * num_tiles], digit_total: u32[RADIX]; num_tiles = ceil(n / TILE).
int pass_first_i64(const int64_t* index, uint32_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                   uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream);
int pass_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, int64_t n,
             int shift, uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream);
int pass_first_u32(const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                   uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream);
int pass_first_f32(const float* keys_in, uint32_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                   uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream);
int pass_first_u64(const uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                   uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream);
int pass_u64(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n,
             int shift, uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream);
int pass_first_dstval_f32(const DstValSrc<float>* desc, uint64_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                          uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream);
int pass_first_dstval_f16(const DstValSrc<__half>* desc, uint64_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                          uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream);
int pass_first_dstval_bf16(const DstValSrc<__hip_bfloat16>* desc, uint64_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                           uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream);


}  // namespace sortengine
