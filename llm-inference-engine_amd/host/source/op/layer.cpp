#include "op/layer.h"

#include <numeric>

namespace op {
base::Status Layer::check_tensor(const tensor::Tensor& tensor, base::DeviceType device_type,
                                 base::DataType data_type) const {
  if (tensor.is_empty()) return base::error::InvalidArgument("The tensor parameter is empty.");
  if (tensor.device_type() != device_type)
    return base::error::InvalidArgument("The tensor has a wrong device type.");
  if (tensor.data_type() != data_type)
    return base::error::InvalidArgument("The tensor has a wrong data type.");
  return base::error::Success();
}

base::Status Layer::check_tensor_with_dim(const tensor::Tensor& tensor,
                                          base::DeviceType device_type, base::DataType data_type,
                                          std::initializer_list<int64_t> dims) const {
  auto status = check_tensor(tensor, device_type, data_type);
  if (!status) return status;
  // as in the reference, the tensor's own rank decides how many expected dims are consumed
  int32_t i = 0;
  for (int64_t want : dims) {
    if (i >= tensor.dims_size()) break;
    if (want != tensor.get_dim(i))
      return base::error::InvalidArgument("The tensor has a wrong dim in dim" + std::to_string(i));
    ++i;
  }
  if (i < tensor.dims_size())
    return base::error::InvalidArgument("The tensor has more dims than the layer expects");
  return base::error::Success();
}

base::Status Layer::check() const {
  return base::error::FunctionNotImplement("The check function is not implement yet");
}
base::Status Layer::forward() { return base::error::FunctionNotImplement(""); }

void Layer::set_input(int32_t idx, const tensor::Tensor& input) {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(inputs_.size()));
  inputs_[idx] = input;
}
void Layer::set_output(int32_t idx, const tensor::Tensor& output) {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(outputs_.size()));
  outputs_[idx] = output;
}
const tensor::Tensor& Layer::get_input(int32_t idx) const {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(inputs_.size()));
  return inputs_[idx];
}
tensor::Tensor& Layer::get_input(int32_t idx) {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(inputs_.size()));
  return inputs_[idx];
}
const tensor::Tensor& Layer::get_output(int32_t idx) const {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(outputs_.size()));
  return outputs_[idx];
}
tensor::Tensor& Layer::get_output(int32_t idx) {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(outputs_.size()));
  return outputs_[idx];
}

void Layer::to_cuda() {
  for (auto& t : inputs_)
    if (!t.is_empty()) t.to_hip(stream());
  for (auto& t : outputs_)
    if (!t.is_empty()) t.to_hip(stream());
}

void Layer::set_cuda_config(std::shared_ptr<kernel::CudaConfig> config) {
  if (config) cuda_config_ = std::move(config);
}

// forward(in..., out): bind then run (layer.cpp:237-285)
base::Status Layer::forward(const tensor::Tensor& in1, const tensor::Tensor& out1) {
  set_input(0, in1);
  set_output(0, out1);
  return forward();
}
base::Status Layer::forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                            const tensor::Tensor& out1) {
  set_input(0, in1);
  set_input(1, in2);
  set_output(0, out1);
  return forward();
}
base::Status Layer::forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                            const tensor::Tensor& in3, const tensor::Tensor& out1) {
  set_input(0, in1);
  set_input(1, in2);
  set_input(2, in3);
  set_output(0, out1);
  return forward();
}
base::Status Layer::forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                            const tensor::Tensor& in3, const tensor::Tensor& in4,
                            const tensor::Tensor& out1) {
  set_input(0, in1);
  set_input(1, in2);
  set_input(2, in3);
  set_input(3, in4);
  set_output(0, out1);
  return forward();
}
base::Status Layer::forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                            const tensor::Tensor& in3, const tensor::Tensor& in4,
                            const tensor::Tensor& in5, const tensor::Tensor& out1) {
  set_input(0, in1);
  set_input(1, in2);
  set_input(2, in3);
  set_input(3, in4);
  set_input(4, in5);
  set_output(0, out1);
  return forward();
}

tensor::Tensor& LayerParam::get_weight(int32_t idx) {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(weights_.size()));
  return weights_[idx];
}
const tensor::Tensor& LayerParam::get_weight(int32_t idx) const {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(weights_.size()));
  return weights_[idx];
}

void LayerParam::to_cuda() {
  Layer::to_cuda();
  for (auto& w : weights_)
    if (!w.is_empty()) w.to_hip(stream());
  if (!scales_.is_empty()) scales_.to_hip(stream());
}

base::Status LayerParam::set_weight(int32_t idx, const tensor::Tensor& weight) {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(weights_.size()));
  const base::DataType want =
      is_quant_layer_ ? base::DataType::kDataTypeInt8 : base::DataType::kDataTypeFp32;
  CHECK(weight.data_type() == want);
  if (!weight.is_empty()) CHECK(weight.device_type() == device_type_);
  weights_[idx] = weight;
  return base::error::Success();
}

base::Status LayerParam::set_weight(int32_t idx, const std::vector<int32_t>& dims,
                                    const void* weight_ptr, base::DeviceType device_type) {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(weights_.size()));
  CHECK_NE(weight_ptr, nullptr);
  const size_t count = std::accumulate(dims.begin(), dims.end(), size_t{1}, std::multiplies<>());
  if (!is_quant_layer_) {
    tensor::Tensor weight(base::DataType::kDataTypeFp32, dims, false, nullptr,
                          const_cast<void*>(weight_ptr));
    if (device_type != base::DeviceType::kDeviceUnknown) weight.set_device_type(device_type);
    weights_[idx] = weight;
    return base::error::Success();
  }
  // int8 block, then count/group_size fp32 scales right behind it (layer.cpp:209-214)
  CHECK_GT(group_size_, 0) << "set_group_size must precede set_weight on a quant layer";
  CHECK(count % static_cast<size_t>(group_size_) == 0);
  tensor::Tensor weight(base::DataType::kDataTypeInt8, dims, false, nullptr,
                        const_cast<void*>(weight_ptr));
  const int32_t scale_nums = static_cast<int32_t>(count / group_size_);
  const int8_t* bytes = static_cast<const int8_t*>(weight_ptr);
  tensor::Tensor scales(base::DataType::kDataTypeFp32, scale_nums, false, nullptr,
                        const_cast<int8_t*>(bytes + count));
  if (device_type != base::DeviceType::kDeviceUnknown) {
    weight.set_device_type(device_type);
    scales.set_device_type(device_type);
  }
  weights_[idx] = weight;
  scales_ = scales;
  return base::error::Success();
}

void LayerParam::set_scales(const tensor::Tensor& scales) {
  CHECK(!scales.is_empty());
  scales_ = scales;
}

int32_t LayerParam::get_scale_num() const {
  CHECK(!scales_.is_empty());
  return static_cast<int32_t>(scales_.size());
}
}  // namespace op
