// Layer hierarchy of the host side: BaseLayer -> Layer -> LayerParam.
// Same public surface as kuiper/include/op/layer.h:25-188 (forward() plus the 1..5-input
// convenience overloads, set_input/set_output, set_weight incl. the int8 weights+scales blob
// contract, to_cuda, set_cuda_config) so reference call sites compile against it unchanged.
#ifndef KX_HOST_OP_LAYER_H_
#define KX_HOST_OP_LAYER_H_
#include <initializer_list>
#include <memory>
#include <string>
#include <vector>

#include "base/base.h"
#include "base/hip_config.h"
#include "tensor/tensor.h"

namespace op {
enum class LayerType : uint8_t {
  kLayerUnknown = 0,
  kLayerLinear = 1,
  kLayerEncode = 2,
  kLayerEmbedding = 3,
  kLayerRMSNorm = 4,
  kLayerMatmul = 5,
  kLayerRoPe = 6,
  kLayerMHA = 7,
  kLayerSoftmax = 8,
  kLayerAdd = 9,
  kLayerSwiGLU = 10,
};

class BaseLayer {
 public:
  explicit BaseLayer(base::DeviceType device_type, LayerType layer_type, base::DataType data_type,
                     std::string layer_name = "")
      : layer_name_(std::move(layer_name)),
        layer_type_(layer_type),
        data_type_(data_type),
        device_type_(device_type) {}
  virtual ~BaseLayer() = default;

  base::DataType data_type() const { return data_type_; }
  LayerType layer_type() const { return layer_type_; }

  virtual base::Status init() = 0;
  virtual base::Status forward() = 0;
  virtual base::Status forward(const tensor::Tensor& in1, const tensor::Tensor& out1) = 0;
  virtual base::Status forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                               const tensor::Tensor& out1) = 0;
  virtual base::Status forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                               const tensor::Tensor& in3, const tensor::Tensor& out1) = 0;
  virtual base::Status forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                               const tensor::Tensor& in3, const tensor::Tensor& in4,
                               const tensor::Tensor& out1) = 0;
  virtual base::Status forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                               const tensor::Tensor& in3, const tensor::Tensor& in4,
                               const tensor::Tensor& in5, const tensor::Tensor& out1) = 0;
  virtual void set_input(int32_t idx, const tensor::Tensor& input) = 0;
  virtual void set_output(int32_t idx, const tensor::Tensor& output) = 0;
  virtual size_t input_size() const = 0;
  virtual size_t output_size() const = 0;
  virtual base::Status check() const = 0;
  virtual tensor::Tensor& get_input(int32_t idx) = 0;
  virtual tensor::Tensor& get_output(int32_t idx) = 0;
  virtual const tensor::Tensor& get_input(int32_t idx) const = 0;
  virtual const tensor::Tensor& get_output(int32_t idx) const = 0;

  virtual base::Status set_weight(int32_t, const tensor::Tensor&) {
    return base::error::FunctionNotImplement();
  }
  virtual base::Status set_weight(int32_t, const std::vector<int32_t>&, const void*,
                                  base::DeviceType = base::DeviceType::kDeviceUnknown) {
    return base::error::FunctionNotImplement();
  }

  const std::string& get_layer_name() const { return layer_name_; }
  void set_layer_name(const std::string& n) { layer_name_ = n; }
  base::DeviceType device_type() const { return device_type_; }
  void set_device_type(base::DeviceType d) { device_type_ = d; }

 protected:
  std::string layer_name_;
  LayerType layer_type_ = LayerType::kLayerUnknown;
  base::DataType data_type_ = base::DataType::kDataTypeUnknown;
  base::DeviceType device_type_ = base::DeviceType::kDeviceUnknown;
};

class Layer : public BaseLayer {
 public:
  explicit Layer(base::DeviceType device_type, LayerType layer_type, std::string layer_name = "")
      : BaseLayer(device_type, layer_type, base::DataType::kDataTypeFp32, std::move(layer_name)) {}

  base::Status init() override { return base::error::Success(); }

  base::Status check_tensor(const tensor::Tensor& tensor, base::DeviceType device_type,
                            base::DataType data_type) const;
  // Dimension check.  The reference takes the expected dims as C varargs (layer.cpp:57-81);
  // an initializer list carries the same information with type checking.
  base::Status check_tensor_with_dim(const tensor::Tensor& tensor, base::DeviceType device_type,
                                     base::DataType data_type,
                                     std::initializer_list<int64_t> dims) const;
  template <typename... D>
  base::Status check_tensor_with_dim(const tensor::Tensor& tensor, base::DeviceType device_type,
                                     base::DataType data_type, D... dims) const {
    return check_tensor_with_dim(tensor, device_type, data_type,
                                 std::initializer_list<int64_t>{static_cast<int64_t>(dims)...});
  }

  base::Status check() const override;
  base::Status forward() override;
  base::Status forward(const tensor::Tensor& in1, const tensor::Tensor& out1) override;
  base::Status forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                       const tensor::Tensor& out1) override;
  base::Status forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                       const tensor::Tensor& in3, const tensor::Tensor& out1) override;
  base::Status forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                       const tensor::Tensor& in3, const tensor::Tensor& in4,
                       const tensor::Tensor& out1) override;
  base::Status forward(const tensor::Tensor& in1, const tensor::Tensor& in2,
                       const tensor::Tensor& in3, const tensor::Tensor& in4,
                       const tensor::Tensor& in5, const tensor::Tensor& out1) override;

  void set_input(int32_t idx, const tensor::Tensor& input) override;
  void set_output(int32_t idx, const tensor::Tensor& output) override;
  const tensor::Tensor& get_input(int32_t idx) const override;
  const tensor::Tensor& get_output(int32_t idx) const override;
  tensor::Tensor& get_input(int32_t idx) override;
  tensor::Tensor& get_output(int32_t idx) override;
  size_t input_size() const override { return inputs_.size(); }
  size_t output_size() const override { return outputs_.size(); }
  void reset_input_size(size_t size) { inputs_.resize(size); }
  void reset_output_size(size_t size) { outputs_.resize(size); }

  // Moves resident tensors to the device.  Name kept from the reference (layer.cpp:127-138).
  virtual void to_cuda();
  void to_hip() { to_cuda(); }

  void set_cuda_config(std::shared_ptr<kernel::CudaConfig> config);
  std::shared_ptr<kernel::CudaConfig> cuda_config() const { return cuda_config_; }

 protected:
  void* stream() const { return cuda_config_ ? cuda_config_->stream : nullptr; }
  std::vector<tensor::Tensor> inputs_;
  std::vector<tensor::Tensor> outputs_;
  std::shared_ptr<kernel::CudaConfig> cuda_config_;
};

class LayerParam : public Layer {
 public:
  explicit LayerParam(base::DeviceType device_type, LayerType layer_type,
                      bool is_quant_layer = false, std::string layer_name = "")
      : Layer(device_type, layer_type, std::move(layer_name)), is_quant_layer_(is_quant_layer) {}

  size_t weight_size() const { return weights_.size(); }
  void reset_weight_size(size_t size) { weights_.resize(size); }
  tensor::Tensor& get_weight(int32_t idx);
  const tensor::Tensor& get_weight(int32_t idx) const;
  void to_cuda() override;

  base::Status set_weight(int32_t idx, const tensor::Tensor& weight) override;
  // For a quant layer `weight_ptr` addresses the int8 block and the fp32 scales follow it
  // immediately (layer.cpp:203-216); group_size must have been set first.
  base::Status set_weight(int32_t idx, const std::vector<int32_t>& dims, const void* weight_ptr,
                          base::DeviceType device_type = base::DeviceType::kDeviceUnknown) override;
  void set_scales(const tensor::Tensor& scales);
  void set_group_size(int32_t group_size) { group_size_ = group_size; }
  int32_t get_scale_num() const;
  bool is_quant_layer() const { return is_quant_layer_; }

 protected:
  int32_t group_size_ = 0;
  bool is_quant_layer_ = false;
  tensor::Tensor scales_;
  std::vector<tensor::Tensor> weights_;
};
}  // namespace op
#endif
