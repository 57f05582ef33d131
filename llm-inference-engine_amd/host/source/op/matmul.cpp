#include "op/matmul.h"

#include "base/rccl_handler.h"
#include "kernels/kernels_interface.h"
#include "kx_hip.h"

namespace op {
MatmulLayer::MatmulLayer(base::DeviceType device_type, int32_t dim0, int32_t dim1,
                         bool is_quant_layer, bool has_bias)
    : LayerParam(device_type, LayerType::kLayerMatmul, is_quant_layer, "Matmul"),
      dim0_(dim0),
      dim1_(dim1),
      has_bias_(has_bias) {
  reset_input_size(1);
  reset_output_size(1);
  reset_weight_size(1);
  if (has_bias_) bias_.resize(1);
}

base::Status MatmulLayer::check() const {
  // input is [dim1] (one token) or [T, dim1] (prefill rows); output [dim0] or [T, dim0]
  const tensor::Tensor& in = get_input(0);
  const tensor::Tensor& out = get_output(0);
  base::Status status;
  const bool batched = in.dims_size() == 2;
  status = batched ? check_tensor_with_dim(in, device_type_, data_type_, in.get_dim(0), dim1_)
                   : check_tensor_with_dim(in, device_type_, data_type_, dim1_);
  if (!status) {
    LOG(ERROR) << "The input tensor error in the matmul layer.";
    return status;
  }
  status = check_tensor_with_dim(
      get_weight(0), device_type_,
      is_quant_layer_ ? base::DataType::kDataTypeInt8 : data_type_, dim0_, dim1_);
  if (!status) {
    LOG(ERROR) << "The weight tensor error in the matmul layer.";
    return status;
  }
  if (is_quant_layer_) {
    status = check_tensor_with_dim(scales_, device_type_, base::DataType::kDataTypeFp32,
                                   scales_.size());
    if (!status) {
      LOG(ERROR) << "The scale tensor error in the matmul layer.";
      return status;
    }
    if (group_size_ <= 0 ||
        scales_.size() * static_cast<size_t>(group_size_) != get_weight(0).size())
      return base::error::InvalidArgument("The scale count does not match weight/group_size.");
  }
  status = batched ? check_tensor_with_dim(out, device_type_, data_type_, in.get_dim(0), dim0_)
                   : check_tensor_with_dim(out, device_type_, data_type_, dim0_);
  if (!status) {
    LOG(ERROR) << "The output tensor error in the matmul layer.";
    return status;
  }
  return base::error::Success();
}

base::Status MatmulLayer::forward() {
  auto status = check();
  if (!status) return status;
  if (device_type_ == base::DeviceType::kDeviceHIP) CHECK(cuda_config_ != nullptr);
  const kernel::CudaConfig* cfg = cuda_config_ ? cuda_config_.get() : nullptr;
  if (is_quant_layer_) {
    kernel::get_matmul_kernel_quant8(device_type_)(get_input(0), get_weight(0), get_output(0),
                                                   group_size_, scales_, cfg);
  } else {
    kernel::get_matmul_kernel(device_type_)(get_input(0), get_weight(0), get_output(0), 1.f, cfg);
  }
  if (has_bias_) {
    // one bias row per output row; batched outputs add it row by row
    const tensor::Tensor& out = get_output(0);
    const int32_t rows = out.dims_size() == 2 ? out.get_dim(0) : 1;
    for (int32_t r = 0; r < rows; ++r) {
      tensor::Tensor row(base::DataType::kDataTypeFp32, dim0_, false, nullptr,
                         const_cast<float*>(out.ptr<float>(static_cast<int64_t>(r) * dim0_)));
      row.set_device_type(device_type_);
      kernel::get_add_kernel(device_type_)(row, get_bias(0), row, stream());
    }
  }

  if (need_all_reduce_ && tp_size_ > 1 && device_type_ == base::DeviceType::kDeviceHIP) {
    void* ptr = get_output(0).ptr<void>();
    const size_t count = get_output(0).size();
    if (reducer_) return reducer_(static_cast<float*>(ptr), count);
    auto comm = base::RcclHandler::get_instance();
    // A partial sum that is never reduced is silent corruption: refuse instead of returning it.
    if (!comm->raw_comm() || comm->world_size() != tp_size_)
      return base::error::InternalError(
          "row-parallel MatmulLayer with tp_size " + std::to_string(tp_size_) +
          " but no RCCL communicator of that size: call RcclHandler::Init with the unique id first");
    if (cuda_config_->comm_stream && cuda_config_->compute_done && cuda_config_->comm_done) {
      // compute stream -> (event) -> comm stream runs the all-reduce -> (event) -> compute
      // stream.  Same ordering as matmul.cpp:93-116, with the two events owned by the config
      // and reused instead of being created and destroyed on every call.
      int rc = kx_event_record(cuda_config_->compute_done, cuda_config_->stream);
      if (!rc) rc = kx_stream_wait_event(cuda_config_->comm_stream, cuda_config_->compute_done);
      if (rc) return base::error::InternalError(kx_error_string(rc));
      status = comm->AllReduce(ptr, ptr, count, data_type_, cuda_config_->comm_stream);
      if (!status) return status;
      rc = kx_event_record(cuda_config_->comm_done, cuda_config_->comm_stream);
      if (!rc) rc = kx_stream_wait_event(cuda_config_->stream, cuda_config_->comm_done);
      if (rc) return base::error::InternalError(kx_error_string(rc));
    } else {
      status = comm->AllReduce(ptr, ptr, count, data_type_, cuda_config_->stream);
      if (!status) return status;
    }
  }
  return base::error::Success();
}

void MatmulLayer::set_tp_config(int32_t tp_size, int32_t tp_rank, bool need_all_reduce) {
  tp_size_ = tp_size;
  tp_rank_ = tp_rank;
  need_all_reduce_ = need_all_reduce;
}

base::Status MatmulLayer::set_bias(int32_t idx, int32_t& dim, const void* bias_ptr,
                                   base::DeviceType device_type) {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(bias_.size()));
  CHECK_NE(bias_ptr, nullptr);
  tensor::Tensor bias(base::DataType::kDataTypeFp32, dim, false, nullptr,
                      const_cast<void*>(bias_ptr));
  if (device_type != base::DeviceType::kDeviceUnknown) bias.set_device_type(device_type);
  bias_[idx] = bias;
  return base::error::Success();
}

tensor::Tensor& MatmulLayer::get_bias(int32_t idx) {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(bias_.size()));
  return bias_[idx];
}
const tensor::Tensor& MatmulLayer::get_bias(int32_t idx) const {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(bias_.size()));
  return bias_[idx];
}

void MatmulLayer::to_cuda() {
  LayerParam::to_cuda();
  for (auto& b : bias_)
    if (!b.is_empty()) b.to_hip(stream());
}
}  // namespace op
