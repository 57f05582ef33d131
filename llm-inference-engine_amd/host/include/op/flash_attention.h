// Attention for Tq >= 1 query rows over the KV cache.  Same interface as op::FlashAttention
// (kuiper/include/op/flash_attention.h, source/op/flash_attention.cpp:18-88): inputs (Q, K cache,
// V cache), one output, softmax_scale defaulting to 1/sqrt(head_size), causal.
// Q is [Tq][head_num][head_size]; the last row sits at set_pos(pos).
#ifndef KX_HOST_OP_FLASH_ATTENTION_H_
#define KX_HOST_OP_FLASH_ATTENTION_H_
#include "op/layer.h"

namespace op {
class FlashAttention : public Layer {
 public:
  explicit FlashAttention(base::DeviceType device_type, int32_t layer_index, int32_t kv_mul,
                          int32_t kv_dim, int32_t seq_len, int32_t head_num, int32_t head_size,
                          float scale = 0.0f);
  using Layer::forward;  // keep the forward(in..., out) overloads visible
  base::Status forward() override;
  void set_pos(int32_t pos) { pos_ = pos; }
  void set_layer_idx(int32_t layer_idx) { layer_index_ = layer_idx; }
  void set_causal(bool causal) { is_causal_ = causal; }

 private:
  base::Status check() const override;
  int32_t layer_index_ = 0;
  int32_t kv_mul_ = 0;
  int32_t kv_dim_ = 0;
  int32_t seq_len_ = 0;
  int32_t head_num_ = 0;
  int32_t head_size_ = 0;
  int32_t pos_ = 0;
  float softmax_scale_ = 0.0f;
  bool is_causal_ = true;
};
}  // namespace op
#endif
