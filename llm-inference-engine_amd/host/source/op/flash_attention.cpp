#include "op/flash_attention.h"

#include <cmath>

#include "kernels/kernels_interface.h"

namespace op {
FlashAttention::FlashAttention(base::DeviceType device_type, int32_t layer_index, int32_t kv_mul,
                               int32_t kv_dim, int32_t seq_len, int32_t head_num,
                               int32_t head_size, float scale)
    : Layer(device_type, LayerType::kLayerMHA, "FlashAttention"),
      layer_index_(layer_index),
      kv_mul_(kv_mul),
      kv_dim_(kv_dim),
      seq_len_(seq_len),
      head_num_(head_num),
      head_size_(head_size),
      softmax_scale_(scale) {
  if (softmax_scale_ == 0.0f)  // flash_attention.cpp:30-33
    softmax_scale_ = 1.0f / std::sqrt(static_cast<float>(head_size_));
  reset_input_size(3);
  reset_output_size(1);
}

base::Status FlashAttention::check() const {
  for (int32_t i = 0; i < 3; ++i) {
    auto status = check_tensor(get_input(i), device_type_, data_type_);
    if (!status) {
      LOG(ERROR) << "The input tensor " << i << " error in the FlashAttention layer.";
      return status;
    }
  }
  const size_t row = static_cast<size_t>(head_num_) * head_size_;
  if (row == 0 || get_input(0).size() % row != 0)
    return base::error::InvalidArgument("The query size is not a multiple of head_num*head_size.");
  const int64_t tq = static_cast<int64_t>(get_input(0).size() / row);
  if (pos_ < 0 || pos_ >= seq_len_ || pos_ - (tq - 1) < 0)
    return base::error::InvalidArgument("The query rows do not fit before position pos.");
  if (get_output(0).size() != get_input(0).size())
    return base::error::InvalidArgument("The output size differs from the query size.");
  return check_tensor(get_output(0), device_type_, data_type_);
}

base::Status FlashAttention::forward() {
  auto status = check();
  if (!status) return status;
  if (device_type_ == base::DeviceType::kDeviceHIP) CHECK(cuda_config_ != nullptr);
  const int32_t kv_head_num = head_num_ / kv_mul_;
  kernel::get_flash_attention_kernel(device_type_)(
      get_input(0), get_input(1), get_input(2), get_output(0), head_num_, kv_head_num, head_size_,
      seq_len_, pos_, layer_index_, softmax_scale_, is_causal_, device_type_,
      cuda_config_ ? cuda_config_.get() : nullptr);
  return base::error::Success();
}
}  // namespace op
