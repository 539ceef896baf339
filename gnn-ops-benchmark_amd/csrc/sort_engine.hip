// sort_engine.hip — instantiations of the radix pass used by plan.hip, sort.hip and sparse.hip.
#include "sort_engine_impl.h"

namespace sortengine {

int pass_first_i64(const int64_t* index, uint32_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                   uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream) {
    return run_pass<KeyI64Low32, true, true>(index, nullptr, keys_out, vals_out, n, shift, tile_hist, digit_total,
                                             num_tiles, stream);
}

int pass_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, int64_t n,
             int shift, uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream) {
    return run_pass<KeyU32, false, true>(keys_in, vals_in, keys_out, vals_out, n, shift, tile_hist, digit_total,
                                         num_tiles, stream);
}

int pass_first_u32(const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                   uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream) {
    return run_pass<KeyU32, true, true>(keys_in, nullptr, keys_out, vals_out, n, shift, tile_hist, digit_total, num_tiles,
                                        stream);
}

int pass_first_f32(const float* keys_in, uint32_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                   uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream) {
    return run_pass<KeyF32, true, true>(keys_in, nullptr, keys_out, vals_out, n, shift, tile_hist, digit_total,
                                        num_tiles, stream);
}

int pass_first_u64(const uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                   uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream) {
    return run_pass<KeyU64, true, true>(keys_in, nullptr, keys_out, vals_out, n, shift, tile_hist, digit_total,
                                        num_tiles, stream);
}

int pass_u64(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n,
             int shift, uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream) {
    return run_pass<KeyU64, false, true>(keys_in, vals_in, keys_out, vals_out, n, shift, tile_hist, digit_total,
                                         num_tiles, stream);
}

int pass_first_dstval_f32(const DstValSrc<float>* desc, uint64_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                          uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream) {
    return run_pass<KeyDstVal<float>, true, true>(desc, nullptr, keys_out, vals_out, n, shift, tile_hist, digit_total, num_tiles, stream);
}

int pass_first_dstval_f16(const DstValSrc<__half>* desc, uint64_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                          uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream) {
    return run_pass<KeyDstVal<__half>, true, true>(desc, nullptr, keys_out, vals_out, n, shift, tile_hist, digit_total, num_tiles, stream);
}

int pass_first_dstval_bf16(const DstValSrc<__hip_bfloat16>* desc, uint64_t* keys_out, uint32_t* vals_out, int64_t n, int shift,
                           uint32_t* tile_hist, uint32_t* digit_total, int num_tiles, hipStream_t stream) {
    return run_pass<KeyDstVal<__hip_bfloat16>, true, true>(desc, nullptr, keys_out, vals_out, n, shift, tile_hist, digit_total, num_tiles,
                                                          stream);
}

}  // namespace sortengine
