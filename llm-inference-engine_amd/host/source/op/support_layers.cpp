#include "op/support_layers.h"

#include "kernels/kernels_interface.h"

namespace op {
// ---- RmsNorm (kuiper/source/op/rmsnorm.cpp) ----
RmsNormLayer::RmsNormLayer(base::DeviceType device_type, int32_t dim)
    : LayerParam(device_type, LayerType::kLayerRMSNorm, false, "RMSNorm"), dim_(dim) {
  reset_input_size(1);
  reset_output_size(1);
  reset_weight_size(1);
}
base::Status RmsNormLayer::check() const {
  const auto& in = get_input(0);
  auto status = check_tensor(in, device_type_, data_type_);
  if (!status) return status;
  if (in.size() % static_cast<size_t>(dim_) != 0)
    return base::error::InvalidArgument("The rmsnorm input is not a whole number of rows.");
  status = check_tensor_with_dim(get_weight(0), device_type_, data_type_, dim_);
  if (!status) return status;
  status = check_tensor(get_output(0), device_type_, data_type_);
  if (!status) return status;
  if (get_output(0).size() != in.size())
    return base::error::InvalidArgument("The rmsnorm output size differs from the input size.");
  return base::error::Success();
}
base::Status RmsNormLayer::forward() {
  auto status = check();
  if (!status) return status;
  kernel::ScopedKnobs knobs(-1, eps_);
  kernel::get_rmsnorm_kernel(device_type_)(get_input(0), get_weight(0), get_output(0), stream());
  return base::error::Success();
}

// ---- RoPE (kuiper/source/op/rope.cpp) ----
RoPELayer::RoPELayer(base::DeviceType device_type, int32_t dim, int32_t kv_dim, int32_t head_size)
    : Layer(device_type, LayerType::kLayerRoPe, "RoPe"),
      dim_(dim),
      kv_dim_(kv_dim),
      head_size_(head_size) {
  reset_input_size(5);
  reset_output_size(1);
}
base::Status RoPELayer::check() const {
  auto status = check_tensor_with_dim(get_input(2), base::DeviceType::kDeviceCPU,
                                      base::DataType::kDataTypeInt32, 1);
  if (!status) {
    LOG(ERROR) << "The position tensor error in the rope layer.";
    return status;
  }
  status = check_tensor_with_dim(get_input(0), device_type_, data_type_, dim_);
  if (!status) return status;
  status = check_tensor_with_dim(get_input(1), device_type_, data_type_, kv_dim_);
  if (!status) return status;
  status = check_tensor(get_input(3), device_type_, data_type_);
  if (!status) return status;
  return check_tensor(get_input(4), device_type_, data_type_);
}
base::Status RoPELayer::forward() {
  auto status = check();
  if (!status) return status;
  kernel::ScopedKnobs knobs(variant_, -1.f);
  kernel::get_rope_kernel(device_type_)(dim_, kv_dim_, head_size_, get_input(0), get_input(1),
                                        get_input(2), get_input(3), get_input(4), stream());
  return base::error::Success();
}

// ---- VecAdd (kuiper/source/op/add.cpp) ----
VecAddLayer::VecAddLayer(base::DeviceType device_type)
    : Layer(device_type, LayerType::kLayerAdd, "Add") {
  reset_input_size(2);
  reset_output_size(1);
}
base::Status VecAddLayer::check() const {
  const size_t n = get_input(0).size();
  for (const tensor::Tensor* t : {&get_input(0), &get_input(1), &get_output(0)}) {
    auto status = check_tensor(*t, device_type_, data_type_);
    if (!status) return status;
    if (t->size() != n) return base::error::InvalidArgument("The add operands differ in size.");
  }
  return base::error::Success();
}
base::Status VecAddLayer::forward() {
  auto status = check();
  if (!status) return status;
  kernel::get_add_kernel(device_type_)(get_input(0), get_input(1), get_output(0), stream());
  return base::error::Success();
}

// ---- SwiGLU (kuiper/source/op/swiglu.cpp) ----
SwiGLULayer::SwiGLULayer(base::DeviceType device_type, int32_t hidden_dim)
    : Layer(device_type, LayerType::kLayerSwiGLU, "SwiGLU"), hidden_dim_(hidden_dim) {
  reset_input_size(2);
  reset_output_size(1);
}
base::Status SwiGLULayer::check() const {
  for (const tensor::Tensor* t : {&get_input(0), &get_input(1), &get_output(0)}) {
    auto status = check_tensor(*t, device_type_, data_type_);
    if (!status) return status;
    if (t->size() % static_cast<size_t>(hidden_dim_) != 0 || t->size() != get_input(0).size())
      return base::error::InvalidArgument("The swiglu operands have a wrong size.");
  }
  return base::error::Success();
}
base::Status SwiGLULayer::forward() {
  auto status = check();
  if (!status) return status;
  kernel::get_swiglu_kernel(device_type_)(get_input(0), get_input(1), get_output(0), stream());
  return base::error::Success();
}

// ---- Embedding (kuiper/source/op/embedding.cpp:5-62) ----
EmbeddingLayer::EmbeddingLayer(base::DeviceType device_type, int32_t dim, int32_t seq_len,
                               int32_t vocab_size)
    : LayerParam(device_type, LayerType::kLayerEmbedding, false, "Embedding"),
      dim_(dim),
      seq_len_(seq_len),
      vocab_size_(vocab_size) {
  reset_weight_size(1);
  reset_input_size(2);
  reset_output_size(1);
}
base::Status EmbeddingLayer::check() const {
  const auto& ids = get_input(0);
  const size_t n = get_input(1).size();
  if (n > ids.size())
    return base::error::InvalidArgument("The number of input tensor is greater than seq len.");
  if (ids.is_empty() || ids.data_type() != base::DataType::kDataTypeInt32)
    return base::error::InvalidArgument("The token id tensor must be a non-empty int32 tensor.");
  auto status = check_tensor_with_dim(get_weight(0), device_type_, data_type_, vocab_size_, dim_);
  if (!status) {
    LOG(ERROR) << "The weight tensor error in the embedding layer.";
    return status;
  }
  status = check_tensor_with_dim(get_output(0), device_type_, data_type_, n, dim_);
  if (!status) {
    LOG(ERROR) << "The output tensor error in the embedding layer.";
    return status;
  }
  return base::error::Success();
}
base::Status EmbeddingLayer::forward() {
  auto status = check();
  if (!status) return status;
  kernel::get_emb_kernel(device_type_)(get_input(0), get_weight(0), get_output(0), vocab_size_,
                                       stream());
  return base::error::Success();
}
}  // namespace op
