// Placeholder entries for the prefill kernels while they are being brought up: T == 1 / Tq == 1
// forward to the decode kernels, larger shapes report KX_EUNSUPPORTED (never a silent fallback).
#include "kx_common.hpp"

extern "C" int kx_gemm_w8_f32(const float* X, const int8_t* w, const float* scales,
                              int32_t group_size, float* Y, int32_t T, int32_t M, int32_t K,
                              void* stream) {
  if (T == 1) return kx_gemv_w8_f32(X, w, scales, group_size, Y, M, K, stream);
  return KX_EUNSUPPORTED;
}

extern "C" int kx_gemm_f32(const float* X, const float* w, float* Y, int32_t T, int32_t M, int32_t K,
                           void* stream) {
  if (T == 1) return kx_gemv_f32(X, w, Y, M, K, 1.f, stream);
  return KX_EUNSUPPORTED;
}

extern "C" int kx_flash_attention_f32(const float* query, const float* key_cache,
                                      const float* value_cache, float* out, int32_t Tq,
                                      int32_t head_num, int32_t kv_head_num, int32_t head_size,
                                      int32_t seq_len, int32_t pos, int32_t layer_idx,
                                      float softmax_scale, int32_t is_causal, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  (void)softmax_scale; (void)is_causal;
  if (Tq == 1 && kv_head_num > 0)
    return kx_mha_decode_f32(query, key_cache, value_cache, out, pos, head_num,
                             head_num / kv_head_num, head_size, kv_head_num * head_size, seq_len,
                             layer_idx, workspace, workspace_bytes, stream);
  return KX_EUNSUPPORTED;
}
