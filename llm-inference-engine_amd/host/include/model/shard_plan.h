// Tensor-parallel shard plan of one transformer layer: which rows / columns of each global weight
// matrix a rank holds, and which rank carries the residual into the row-parallel partial sums.
//
// Column-parallel (output rows split): Wq, Wk, Wv by whole heads, W1 and W3 by hidden units.
// Row-parallel (input columns split): Wo (columns follow the local query heads) and W2 (columns
// follow the local hidden units); their outputs are partial sums over ranks and are all-reduced
// (the hook of kuiper/source/op/matmul.cpp:81-127).  Pure host logic, no device calls.
#ifndef KX_HOST_MODEL_SHARD_PLAN_H_
#define KX_HOST_MODEL_SHARD_PLAN_H_
#include <cstdint>

namespace model {

struct ShardSlice {
  int32_t row0 = 0, rows = 0, col0 = 0, cols = 0;
};

struct ShardPlan {
  ShardSlice wq, wk, wv, wo, w1, w2, w3;
  int32_t head_local = 0, kv_head_local = 0, q_dim_local = 0, kv_dim_local = 0, hidden_local = 0;
  bool carries_residual = false;  // exactly one rank adds x into the partial sum before the reduce
};

// Returns false when the geometry does not divide over tp_size ranks.
inline bool make_shard_plan(int32_t dim, int32_t hidden_dim, int32_t head_num, int32_t kv_head_num,
                            int32_t tp_size, int32_t tp_rank, ShardPlan* out) {
  if (!out || tp_size < 1 || tp_rank < 0 || tp_rank >= tp_size) return false;
  if (dim <= 0 || hidden_dim <= 0 || head_num <= 0 || kv_head_num <= 0) return false;
  if (dim % head_num || head_num % kv_head_num) return false;
  if (head_num % tp_size || kv_head_num % tp_size || hidden_dim % tp_size) return false;
  const int32_t hs = dim / head_num;
  ShardPlan p;
  p.head_local = head_num / tp_size;
  p.kv_head_local = kv_head_num / tp_size;
  p.q_dim_local = p.head_local * hs;
  p.kv_dim_local = p.kv_head_local * hs;
  p.hidden_local = hidden_dim / tp_size;
  p.wq = {tp_rank * p.q_dim_local, p.q_dim_local, 0, dim};
  p.wk = {tp_rank * p.kv_dim_local, p.kv_dim_local, 0, dim};
  p.wv = p.wk;
  p.wo = {0, dim, tp_rank * p.q_dim_local, p.q_dim_local};
  p.w1 = {tp_rank * p.hidden_local, p.hidden_local, 0, dim};
  p.w3 = p.w1;
  p.w2 = {0, dim, tp_rank * p.hidden_local, p.hidden_local};
  p.carries_residual = tp_rank == 0;
  *out = p;
  return true;
}

}  // namespace model
#endif
