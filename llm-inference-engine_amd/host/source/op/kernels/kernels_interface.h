// Kernel registry: function-pointer types and get_*_kernel(DeviceType) getters.
// Same typedefs and getter names as kuiper/source/op/kernels/kernels_interface.h:6-82, so a
// layer written against the reference's registry calls these unchanged.  This library registers
// kernels for base::DeviceType::kDeviceHIP only; any other device type is a fatal error, as an
// unknown device is in the reference (kernels_interfaces.cpp:28-31).  There is no CPU branch:
// the reference's CPU kernels live in the reference.
#ifndef KX_HOST_KERNELS_INTERFACE_H_
#define KX_HOST_KERNELS_INTERFACE_H_
#include "base/hip_config.h"
#include "tensor/tensor.h"

namespace kernel {
typedef void (*AddKernel)(const tensor::Tensor& input1, const tensor::Tensor& input2,
                          const tensor::Tensor& output, void* stream);

typedef void (*MatmulKernel)(const tensor::Tensor& input, const tensor::Tensor& weight,
                             const tensor::Tensor& output, float scale, const CudaConfig* config);

typedef void (*MatmulKernelQuant)(const tensor::Tensor& input, const tensor::Tensor& weight,
                                  const tensor::Tensor& output, int32_t group_size,
                                  const tensor::Tensor& scale, const CudaConfig* config);

typedef void (*EmbeddingKernel)(const tensor::Tensor& input, const tensor::Tensor& weight,
                                const tensor::Tensor& output, int32_t vocab_size, void* stream);

typedef void (*SwigluKernel)(const tensor::Tensor& input1, const tensor::Tensor& input2,
                             const tensor::Tensor& output, void* stream);

typedef void (*MHAKernel)(int32_t pos, int32_t head_num, int32_t layer_index, int32_t seq_len,
                          int32_t kv_dim, int32_t kv_mul, int32_t head_size,
                          const tensor::Tensor& mha_out, const tensor::Tensor& query_tensor,
                          const tensor::Tensor& score_tensor,
                          const tensor::Tensor& key_cache_tensor,
                          const tensor::Tensor& value_cache_tensor, base::DeviceType device_type,
                          CudaConfig*);

typedef void (*RMSNormKernel)(const tensor::Tensor& input, const tensor::Tensor& weight,
                              const tensor::Tensor& output, void* stream);

typedef void (*RMSNormKernelDim)(const tensor::Tensor& input, const tensor::Tensor& weight,
                                 const tensor::Tensor& output, int32_t dim, void* stream);

typedef void (*RoPEKernel)(int32_t dim, int32_t kv_dim, int32_t head_size,
                           const tensor::Tensor& input_q, const tensor::Tensor& input_k,
                           const tensor::Tensor& input_pos, const tensor::Tensor& sin_cache,
                           const tensor::Tensor& cos_cache, void* stream);

typedef void (*ScaleKernel)(float scale, const tensor::Tensor& input, void* stream);

typedef void (*SoftmaxInplaceKernel)(const tensor::Tensor& input, void* stream);

typedef void (*ScaleSumKernel)(const tensor::Tensor& value, const tensor::Tensor& scale,
                               const tensor::Tensor& output, int t, int size, int stride,
                               void* stream);

typedef void (*FlashAttentionKernel)(const tensor::Tensor& query, const tensor::Tensor& key,
                                     const tensor::Tensor& value, const tensor::Tensor& output,
                                     int32_t head_num, int32_t kv_head_num, int32_t head_size,
                                     int32_t seq_len, int32_t pos, int32_t layer_idx,
                                     float softmax_scale, bool is_causal,
                                     base::DeviceType device_type, CudaConfig* config);

AddKernel get_add_kernel(base::DeviceType device_type);
EmbeddingKernel get_emb_kernel(base::DeviceType device_type);
MatmulKernel get_matmul_kernel(base::DeviceType device_type);
MatmulKernelQuant get_matmul_kernel_quant8(base::DeviceType device_type);
MHAKernel get_mha_kernel(base::DeviceType device_type);
RMSNormKernel get_rmsnorm_kernel(base::DeviceType device_type);
RMSNormKernelDim get_rmsnorm_dim_kernel(base::DeviceType device_type);
RoPEKernel get_rope_kernel(base::DeviceType device_type);
ScaleKernel get_scale_kernel(base::DeviceType device_type);
SoftmaxInplaceKernel get_softmax_kernel(base::DeviceType device_type);
ScaleSumKernel get_scale_sum_kernel(base::DeviceType device_type);
SwigluKernel get_swiglu_kernel(base::DeviceType device_type, void* stream = nullptr);
FlashAttentionKernel get_flash_attention_kernel(base::DeviceType device_type);

// Model-level knobs the reference fixes with compile-time macros (LLAMA3_SUPPORT /
// QWEN2_SUPPORT: cpu/rope_kernel.cpp:3,43,83; cpu/rmsnorm_kernel.cpp:24-28).  The registry
// signatures have no slot for them.  RmsNormLayer / RoPELayer carry their own values (set_eps,
// set_rope_variant) and install them for the duration of their kernel call with ScopedKnobs, so
// two models of different families can live in one process; model_knobs() is the per-thread
// value a registry function reads, and what a layer that was given no value of its own falls
// back to.  Defaults = the reference's default build (interleaved RoPE, eps 1e-5).
struct ModelKnobs {
  int32_t rope_variant = 0;  // KX_ROPE_INTERLEAVED
  float rms_eps = 1e-5f;
};
ModelKnobs& model_knobs();
class ScopedKnobs {
 public:
  // negative values keep what is installed already
  ScopedKnobs(int32_t rope_variant, float rms_eps) : saved_(model_knobs()) {
    if (rope_variant >= 0) model_knobs().rope_variant = rope_variant;
    if (rms_eps >= 0.f) model_knobs().rms_eps = rms_eps;
  }
  ~ScopedKnobs() { model_knobs() = saved_; }
  ScopedKnobs(const ScopedKnobs&) = delete;
  ScopedKnobs& operator=(const ScopedKnobs&) = delete;

 private:
  ModelKnobs saved_;
};

// Fills the RoPE tables on the device (sin_cos_cache_calc_cu, cuda/rope_kernel.cu:138-146).
void sin_cos_cache_calc_hip(int head_size, int max_seq_len, float theta,
                            const tensor::Tensor& sin_cache, const tensor::Tensor& cos_cache,
                            void* stream);
}  // namespace kernel
#endif
