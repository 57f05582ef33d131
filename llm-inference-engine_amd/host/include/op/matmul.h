// Linear layer y = W x (+ bias) with optional tensor-parallel all-reduce.
// Same interface as op::MatmulLayer (kuiper/include/op/matmul.h:10-53, source/op/matmul.cpp).
// Weight [dim0, dim1] = [out, in]; fp32, or int8 with per-group fp32 scales.
#ifndef KX_HOST_OP_MATMUL_H_
#define KX_HOST_OP_MATMUL_H_
#include <functional>

#include "op/layer.h"

namespace op {
class MatmulLayer : public LayerParam {
 public:
  explicit MatmulLayer(base::DeviceType device_type, int32_t dim0, int32_t dim1,
                       bool is_quant_layer = false, bool has_bias = false);

  base::Status check() const override;
  using Layer::forward;  // keep the forward(in..., out) overloads visible
  base::Status forward() override;

  // fp32 bias of `dim` elements (the reference's quant branch reinterprets the bias as int8 +
  // scales, matmul.cpp:157-171; its exporters only ever write fp32 biases, so fp32 it is).
  base::Status set_bias(int32_t idx, int32_t& dim, const void* bias_ptr,
                        base::DeviceType device_type);
  tensor::Tensor& get_bias(int32_t idx);
  const tensor::Tensor& get_bias(int32_t idx) const;
  void to_cuda() override;

  // Column-parallel layers (Wq/Wk/Wv/W1/W3) need no exchange; row-parallel layers (Wo/W2) sum
  // their partial outputs across ranks: need_all_reduce = true (matmul.h:28-42).
  void set_tp_config(int32_t tp_size, int32_t tp_rank, bool need_all_reduce = false);
  // Optional: the sum over ranks is delegated to the owner (a model routes it through its own
  // all-reduce, which may be RCCL on the side stream or a test rendezvous).  Without a reducer the
  // layer uses the RcclHandler singleton and fails when that has no communicator of tp_size ranks.
  using Reducer = std::function<base::Status(float* data, size_t count)>;
  void set_reducer(Reducer reducer) { reducer_ = std::move(reducer); }

  int32_t dim0() const { return dim0_; }
  int32_t dim1() const { return dim1_; }

 private:
  int32_t dim0_ = 0;
  int32_t dim1_ = 0;
  bool has_bias_ = false;
  std::vector<tensor::Tensor> bias_;
  int32_t tp_size_ = 1;
  int32_t tp_rank_ = 0;
  bool need_all_reduce_ = false;
  Reducer reducer_;
};
}  // namespace op
#endif
