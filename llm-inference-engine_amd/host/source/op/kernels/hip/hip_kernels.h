// Registry functions for base::DeviceType::kDeviceHIP.  Each unwraps tensors into raw device
// pointers, checks the preconditions the reference checks with glog CHECKs, and calls one entry
// of the C ABI (include/kx_hip.h).  A non-zero return code from the C ABI is fatal here, the
// same observable convention as the reference's CHECK aborts (cuda/matmul_kernel.cu:114-124).
#ifndef KX_HOST_HIP_KERNELS_H_
#define KX_HOST_HIP_KERNELS_H_
#include "../kernels_interface.h"

namespace kernel {
void add_kernel_hip(const tensor::Tensor& input1, const tensor::Tensor& input2,
                    const tensor::Tensor& output, void* stream);
void emb_kernel_hip(const tensor::Tensor& input, const tensor::Tensor& weight,
                    const tensor::Tensor& output, int32_t vocab_size, void* stream);
void matmul_kernel_hip(const tensor::Tensor& input, const tensor::Tensor& weight,
                       const tensor::Tensor& output, float scale, const CudaConfig* config);
void matmul_kernel_hip_qint8(const tensor::Tensor& input, const tensor::Tensor& weight,
                             const tensor::Tensor& output, int32_t group_size,
                             const tensor::Tensor& scale, const CudaConfig* config);
void mha_kernel_hip(int32_t pos, int32_t head_num, int32_t layer_index, int32_t seq_len,
                    int32_t kv_dim, int32_t kv_mul, int32_t head_size, const tensor::Tensor& mha_out,
                    const tensor::Tensor& query_tensor, const tensor::Tensor& score_tensor,
                    const tensor::Tensor& key_cache_tensor, const tensor::Tensor& value_cache_tensor,
                    base::DeviceType device_type, CudaConfig* config);
void rmsnorm_kernel_hip(const tensor::Tensor& input, const tensor::Tensor& weight,
                        const tensor::Tensor& output, void* stream);
void rmsnorm_kernel_hip_dim(const tensor::Tensor& input, const tensor::Tensor& weight,
                            const tensor::Tensor& output, int32_t dim, void* stream);
void rope_kernel_hip(int32_t dim, int32_t kv_dim, int32_t head_size, const tensor::Tensor& input_q,
                     const tensor::Tensor& input_k, const tensor::Tensor& input_pos,
                     const tensor::Tensor& sin_cache, const tensor::Tensor& cos_cache, void* stream);
void swiglu_kernel_hip(const tensor::Tensor& input1, const tensor::Tensor& input2,
                       const tensor::Tensor& output, void* stream);
void scale_inplace_hip(float scale, const tensor::Tensor& input, void* stream);
void softmax_inplace_hip(const tensor::Tensor& input, void* stream);
void scale_sum_kernel_hip(const tensor::Tensor& value, const tensor::Tensor& scale,
                          const tensor::Tensor& output, int pos, int size, int stride, void* stream);
void flash_attention_kernel_hip(const tensor::Tensor& query, const tensor::Tensor& key,
                                const tensor::Tensor& value, const tensor::Tensor& output,
                                int32_t head_num, int32_t kv_head_num, int32_t head_size,
                                int32_t seq_len, int32_t pos, int32_t layer_idx, float softmax_scale,
                                bool is_causal, base::DeviceType device_type, CudaConfig* config);
}  // namespace kernel
#endif
