// Registry bodies: one branch per device type, as kuiper/source/op/kernels/kernels_interfaces.cpp
// :23-159 -- with the HIP device as the only registered one.
#include "kernels_interface.h"

#include "hip/hip_kernels.h"

namespace kernel {
ModelKnobs& model_knobs() {
  static thread_local ModelKnobs knobs;
  return knobs;
}

#define KX_GETTER(RET, NAME, FN, WHAT)                                        \
  RET NAME(base::DeviceType device_type) {                                    \
    if (device_type == base::DeviceType::kDeviceHIP) return FN;               \
    LOG(FATAL) << "Unknown device type for get " WHAT " kernel (this library " \
                  "registers kDeviceHIP only).";                              \
    return nullptr;                                                           \
  }

KX_GETTER(AddKernel, get_add_kernel, add_kernel_hip, "an add")
KX_GETTER(EmbeddingKernel, get_emb_kernel, emb_kernel_hip, "an embedding")
KX_GETTER(MatmulKernel, get_matmul_kernel, matmul_kernel_hip, "a matmul")
KX_GETTER(MatmulKernelQuant, get_matmul_kernel_quant8, matmul_kernel_hip_qint8, "a quant matmul")
KX_GETTER(MHAKernel, get_mha_kernel, mha_kernel_hip, "an mha")
KX_GETTER(RMSNormKernel, get_rmsnorm_kernel, rmsnorm_kernel_hip, "a rmsnorm")
KX_GETTER(RMSNormKernelDim, get_rmsnorm_dim_kernel, rmsnorm_kernel_hip_dim, "a rmsnorm dim")
KX_GETTER(RoPEKernel, get_rope_kernel, rope_kernel_hip, "a rope")
KX_GETTER(ScaleKernel, get_scale_kernel, scale_inplace_hip, "a scale")
KX_GETTER(SoftmaxInplaceKernel, get_softmax_kernel, softmax_inplace_hip, "a softmax")
KX_GETTER(ScaleSumKernel, get_scale_sum_kernel, scale_sum_kernel_hip, "a scale and reduce")
KX_GETTER(FlashAttentionKernel, get_flash_attention_kernel, flash_attention_kernel_hip,
          "a flash attention")
#undef KX_GETTER

SwigluKernel get_swiglu_kernel(base::DeviceType device_type, void* stream) {
  UNUSED(stream);
  if (device_type == base::DeviceType::kDeviceHIP) return swiglu_kernel_hip;
  LOG(FATAL) << "Unknown device type for get a swiglu kernel (this library registers kDeviceHIP only).";
  return nullptr;
}
}  // namespace kernel
