// The small layers of the token path: RmsNormLayer, RoPELayer, VecAddLayer, SwiGLULayer,
// EmbeddingLayer.  Interfaces as kuiper/include/op/{rmsnorm,rope,add,swiglu,embedding}.h.
#ifndef KX_HOST_OP_SUPPORT_LAYERS_H_
#define KX_HOST_OP_SUPPORT_LAYERS_H_
#include <utility>

#include "op/layer.h"

namespace op {
class RmsNormLayer : public LayerParam {
 public:
  explicit RmsNormLayer(base::DeviceType device_type, int32_t dim);
  base::Status check() const override;
  using Layer::forward;  // keep the forward(in..., out) overloads visible
  base::Status forward() override;
  // epsilon of this layer's model (the reference compiles 1e-5 or 1e-6 in,
  // cpu/rmsnorm_kernel.cpp:24-28); unset = the thread's kernel::model_knobs() value
  void set_eps(float eps) { eps_ = eps; }

 private:
  int32_t dim_ = 0;
  float eps_ = -1.f;
};

// inputs: q, k, pos (host int32), sin table, cos table; rotates q and k in place
class RoPELayer : public Layer {
 public:
  explicit RoPELayer(base::DeviceType device_type, int32_t dim, int32_t kv_dim, int32_t head_size);
  base::Status check() const override;
  using Layer::forward;  // keep the forward(in..., out) overloads visible
  base::Status forward() override;
  // pairing of this layer's model: 0 interleaved, 1 half split (the reference's LLAMA3_SUPPORT /
  // QWEN2_SUPPORT switch, cpu/rope_kernel.cpp:3,43,83); unset = kernel::model_knobs()
  void set_rope_variant(int32_t variant) { variant_ = variant; }

 private:
  int32_t dim_ = 0;
  int32_t kv_dim_ = 0;
  int32_t head_size_ = 0;
  int32_t variant_ = -1;
};

class VecAddLayer : public Layer {
 public:
  explicit VecAddLayer(base::DeviceType device_type);
  base::Status check() const override;
  using Layer::forward;  // keep the forward(in..., out) overloads visible
  base::Status forward() override;
};

class SwiGLULayer : public Layer {
 public:
  explicit SwiGLULayer(base::DeviceType device_type, int32_t hidden_dim);
  base::Status check() const override;
  using Layer::forward;  // keep the forward(in..., out) overloads visible
  base::Status forward() override;

 private:
  int32_t hidden_dim_ = 0;
};

struct EmbeddingOutput {
  tensor::Tensor input_tokens;
  tensor::Tensor input_embeddings;
  tensor::Tensor input_token_num;
  explicit EmbeddingOutput(tensor::Tensor tokens, tensor::Tensor embeddings, tensor::Tensor num)
      : input_tokens(std::move(tokens)),
        input_embeddings(std::move(embeddings)),
        input_token_num(std::move(num)) {}
};

// inputs: token ids (host or device int32), token-count tensor; output [n, dim]
class EmbeddingLayer : public LayerParam {
 public:
  explicit EmbeddingLayer(base::DeviceType device_type, int32_t dim, int32_t seq_len,
                          int32_t vocab_size);
  base::Status check() const override;
  using Layer::forward;  // keep the forward(in..., out) overloads visible
  base::Status forward() override;

 private:
  int32_t dim_ = 0;
  int32_t seq_len_ = 0;
  int32_t vocab_size_ = 0;
};
}  // namespace op
#endif
