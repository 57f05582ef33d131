// Argument block and dispatcher shared by the W8 GEMV entry points.
#pragma once
#ifndef KX_FILE_ID
#define KX_FILE_ID 1
#endif
#include "kx_common.hpp"

namespace kx {

enum { GEMV_PLAIN = KX_GEMV_PLAIN, GEMV_GATE_UP = KX_GEMV_GATE_UP, GEMV_QKV_ROPE = KX_GEMV_QKV_ROPE };

struct GemvW8Args {
  const float* x;
  const int8_t* w;
  const float* scales;
  float* y0;
  float* y1;
  float* y2;
  const float* bias;
  const float* residual;
  const float* norm_w;  // non-null: x is RMS-normalised with this weight while it is staged
  float norm_eps;
  int32_t k0, k1;  // row routing: [0,k0)->y0, [k0,k1)->y1, [k1,K)->y2
  int32_t M, K;
  int32_t chunks;          // ceil(M / 1024)
  int32_t gs_shift;        // log2(group_size)
  int32_t groups_per_row;  // M / group_size
  int32_t rows_per_block;
  int32_t swiglu_half;  // GEMV_GATE_UP: rows p and p+swiglu_half are gate/up of output p
  // GEMV_QKV_ROPE
  const float* sin_cache;
  const float* cos_cache;
  const int32_t* pos_dev;  // position read on the device when non-null, else pos_val
  int32_t pos_val;
  int32_t rope_variant;
  int32_t head_size;
  int32_t kv_pos_stride;  // y1/y2 are advanced by pos * kv_pos_stride elements (KV-cache row)
  int32_t kv_f16;         // GEMV_QKV_ROPE: y1/y2 point at fp16 cache rows
  KX_BFIELD(12)  // operands of the bounds-checking build, in the order of GemvOperand
};
enum GemvOperand { GB_X, GB_W, GB_S, GB_Y0, GB_Y1, GB_Y2, GB_BIAS, GB_RES, GB_NORM, GB_SIN, GB_COS, GB_POS };

struct StepArgs;
// pick: the launch also ends the decode step (plain form without row routing only)
int gemv_w8_dispatch(GemvW8Args a, int32_t group_size, int mode, hipStream_t s, const StepArgs* pick = nullptr);

}  // namespace kx
