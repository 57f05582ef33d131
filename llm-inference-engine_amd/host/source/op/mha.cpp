#include "op/mha.h"

#include "kernels/kernels_interface.h"

namespace op {
MultiHeadAttention::MultiHeadAttention(base::DeviceType device_type, int32_t layer_index,
                                       int32_t kv_mul, int32_t kv_dim, int32_t seq_len,
                                       int32_t head_num, int32_t head_size)
    : Layer(device_type, LayerType::kLayerMHA, "MultiHead"),
      layer_index_(layer_index),
      kv_mul_(kv_mul),
      kv_dim_(kv_dim),
      seq_len_(seq_len),
      head_num_(head_num),
      head_size_(head_size) {
  reset_input_size(5);
  reset_output_size(1);
}

base::Status MultiHeadAttention::check() const {
  // query, score scratch, key cache, value cache (mha.cpp:44-56); the fifth input is unused
  for (int32_t i = 0; i < 4; ++i) {
    auto status = check_tensor(get_input(i), device_type_, data_type_);
    if (!status) {
      LOG(ERROR) << "The input tensor " << i << " error in the mha layer.";
      return status;
    }
  }
  if (pos_ < 0 || pos_ >= seq_len_)
    return base::error::InvalidArgument("The position is outside the KV cache.");
  return check_tensor(get_output(0), device_type_, data_type_);
}

base::Status MultiHeadAttention::forward() {
  auto status = check();
  if (!status) return status;
  if (device_type_ == base::DeviceType::kDeviceHIP) CHECK(cuda_config_ != nullptr);
  kernel::get_mha_kernel(device_type_)(pos_, head_num_, layer_index_, seq_len_, kv_dim_, kv_mul_,
                                       head_size_, get_output(0), get_input(0), get_input(1),
                                       get_input(2), get_input(3), device_type_,
                                       cuda_config_ ? cuda_config_.get() : nullptr);
  return base::error::Success();
}
}  // namespace op
