#include "hip_kernels.h"

#include "kx_hip.h"

namespace kernel {
namespace {
constexpr base::DeviceType kHip = base::DeviceType::kDeviceHIP;

void ok(int rc, const char* what) {
  if (rc != 0)
    LOG(FATAL) << what << " failed: " << kx_error_string(rc) << " (rc " << rc << ")";
}
template <typename T>
T* out_ptr(const tensor::Tensor& t) {
  // kernels receive const Tensor& and write the output through it, as the reference does
  // (cuda/matmul_kernel.cu:104)
  return const_cast<T*>(t.ptr<T>());
}
}  // namespace

void add_kernel_hip(const tensor::Tensor& input1, const tensor::Tensor& input2,
                    const tensor::Tensor& output, void* stream) {
  CHECK(!input1.is_empty() && !input2.is_empty() && !output.is_empty());
  CHECK_EQ(input1.size(), input2.size());
  CHECK_EQ(input1.size(), output.size());
  ok(kx_add_f32(input1.ptr<float>(), input2.ptr<float>(), out_ptr<float>(output),
                static_cast<int32_t>(output.size()), stream),
     "kx_add_f32");
}

void emb_kernel_hip(const tensor::Tensor& input, const tensor::Tensor& weight,
                    const tensor::Tensor& output, int32_t vocab_size, void* stream) {
  CHECK(!input.is_empty() && !weight.is_empty() && !output.is_empty());
  CHECK(weight.device_type() == kHip && output.device_type() == kHip);
  const int32_t n = static_cast<int32_t>(input.size());
  const int32_t dim = weight.get_dim(1);
  // The reference keeps token ids on the host and copies them per call (cuda/emb_kernel.cu:27-31);
  // ids already on the device are used in place.
  if (input.device_type() == kHip) {
    ok(kx_embedding_f32(input.ptr<int32_t>(), n, weight.ptr<float>(), dim, vocab_size,
                        out_ptr<float>(output), stream),
       "kx_embedding_f32");
    return;
  }
  tensor::Tensor ids(base::DataType::kDataTypeInt32, n, true,
                     base::HIPDeviceAllocatorFactory::get_instance());
  ok(kx_memcpy_h2d(ids.ptr<int32_t>(), input.ptr<int32_t>(), (size_t)n * sizeof(int32_t), stream),
     "token upload");
  ok(kx_embedding_f32(ids.ptr<int32_t>(), n, weight.ptr<float>(), dim, vocab_size,
                      out_ptr<float>(output), stream),
     "kx_embedding_f32");
  ok(kx_stream_sync(stream), "sync");  // ids is freed on return
}

void matmul_kernel_hip(const tensor::Tensor& input, const tensor::Tensor& weight,
                       const tensor::Tensor& output, float scale, const CudaConfig* config) {
  CHECK(!input.is_empty() && input.dims_size() <= 2);
  CHECK(input.device_type() == kHip);
  CHECK(!weight.is_empty() && weight.dims_size() == 2);
  CHECK(weight.device_type() == kHip);
  const int32_t K = weight.get_dim(0), M = weight.get_dim(1);
  void* stream = config ? config->stream : nullptr;
  if (input.dims_size() == 2 && input.get_dim(0) > 1) {
    // [T, M] rows: the batched (prefill) form, new behind this registry entry
    CHECK_EQ(M, input.get_dim(1));
    const int32_t T = input.get_dim(0);
    CHECK_EQ(output.size(), static_cast<size_t>(T) * K);
    CHECK(scale == 1.f) << "batched matmul has no scale argument";
    ok(kx_gemm_f32(input.ptr<float>(), weight.ptr<float>(), out_ptr<float>(output), T, M, K, stream),
       "kx_gemm_f32");
    return;
  }
  CHECK_EQ(static_cast<size_t>(M), input.size());
  ok(kx_gemv_f32(input.ptr<float>(), weight.ptr<float>(), out_ptr<float>(output), M, K, scale,
                 stream),
     "kx_gemv_f32");
}

void matmul_kernel_hip_qint8(const tensor::Tensor& input, const tensor::Tensor& weight,
                             const tensor::Tensor& output, int32_t group_size,
                             const tensor::Tensor& scale, const CudaConfig* config) {
  CHECK(config != nullptr);  // cuda/matmul_kernel.cu:114
  CHECK(!input.is_empty() && input.dims_size() <= 2);
  CHECK(input.device_type() == kHip);
  CHECK(!weight.is_empty() && weight.dims_size() == 2);
  CHECK(weight.device_type() == kHip);
  const int32_t K = weight.get_dim(0), M = weight.get_dim(1);
  CHECK_EQ(M % 4, 0);  // :123
  if (input.dims_size() == 2 && input.get_dim(0) > 1) {
    CHECK_EQ(M, input.get_dim(1));
    const int32_t T = input.get_dim(0);
    CHECK_EQ(output.size(), static_cast<size_t>(T) * K);
    ok(kx_gemm_w8_f32(input.ptr<float>(), weight.ptr<int8_t>(), scale.ptr<float>(), group_size,
                      out_ptr<float>(output), T, M, K, config->stream),
       "kx_gemm_w8_f32");
    return;
  }
  CHECK_EQ(static_cast<size_t>(M), input.size());
  ok(kx_gemv_w8_f32(input.ptr<float>(), weight.ptr<int8_t>(), scale.ptr<float>(), group_size,
                    out_ptr<float>(output), M, K, config->stream),
     "kx_gemv_w8_f32");
}

void mha_kernel_hip(int32_t pos, int32_t head_num, int32_t layer_index, int32_t seq_len,
                    int32_t kv_dim, int32_t kv_mul, int32_t head_size, const tensor::Tensor& mha_out,
                    const tensor::Tensor& query_tensor, const tensor::Tensor& score_tensor,
                    const tensor::Tensor& key_cache_tensor, const tensor::Tensor& value_cache_tensor,
                    base::DeviceType device_type, CudaConfig* config) {
  UNUSED(device_type);
  CHECK(config != nullptr);
  // Scratch: the config's attention workspace (zeroed once by reserve_attn_workspace) when
  // present.  Else the score tensor the reference passes ([head_num][seq_len], llama3.cpp:523-525):
  // that one is allocated un-zeroed and still holds the last call's contents, so it goes through
  // the entry that assumes nothing about its bytes.
  if (!config->attn_workspace) {
    ok(kx_mha_decode_scores_f32(query_tensor.ptr<float>(), key_cache_tensor.ptr<float>(),
                                value_cache_tensor.ptr<float>(), out_ptr<float>(mha_out), pos,
                                head_num, kv_mul, head_size, kv_dim, seq_len, layer_index,
                                score_tensor.is_empty() ? nullptr : out_ptr<float>(score_tensor),
                                score_tensor.is_empty() ? 0 : score_tensor.byte_size(),
                                config->stream),
       "kx_mha_decode_scores_f32");
    return;
  }
  void* ws = config->attn_workspace;
  size_t ws_bytes = config->attn_workspace_bytes;
  ok(kx_mha_decode_f32(query_tensor.ptr<float>(), key_cache_tensor.ptr<float>(),
                       value_cache_tensor.ptr<float>(), out_ptr<float>(mha_out), pos, head_num,
                       kv_mul, head_size, kv_dim, seq_len, layer_index, ws, ws_bytes,
                       config->stream),
     "kx_mha_decode_f32");
}

void rmsnorm_kernel_hip(const tensor::Tensor& input, const tensor::Tensor& weight,
                        const tensor::Tensor& output, void* stream) {
  CHECK(!input.is_empty() && !weight.is_empty() && !output.is_empty());
  CHECK(input.device_type() == kHip && weight.device_type() == kHip &&
        output.device_type() == kHip);
  const int32_t dim = static_cast<int32_t>(weight.size());
  CHECK_EQ(input.size() % dim, 0u);
  const int32_t rows = static_cast<int32_t>(input.size() / dim);
  ok(kx_rmsnorm_f32(input.ptr<float>(), weight.ptr<float>(), out_ptr<float>(output), dim, rows,
                    model_knobs().rms_eps, stream),
     "kx_rmsnorm_f32");
}

void rmsnorm_kernel_hip_dim(const tensor::Tensor& input, const tensor::Tensor& weight,
                            const tensor::Tensor& output, int32_t dim, void* stream) {
  // rmsnorm_kernel_cu_dim (cuda/rmsnorm_kernel.cu:8-60): rows of `dim` sharing one weight
  CHECK(!input.is_empty() && !weight.is_empty() && !output.is_empty());
  CHECK_EQ(input.size() % dim, 0u);
  ok(kx_rmsnorm_f32(input.ptr<float>(), weight.ptr<float>(), out_ptr<float>(output), dim,
                    static_cast<int32_t>(input.size() / dim), model_knobs().rms_eps, stream),
     "kx_rmsnorm_f32");
}

void rope_kernel_hip(int32_t dim, int32_t kv_dim, int32_t head_size, const tensor::Tensor& input_q,
                     const tensor::Tensor& input_k, const tensor::Tensor& input_pos,
                     const tensor::Tensor& sin_cache, const tensor::Tensor& cos_cache, void* stream) {
  // the position tensor lives on the host, as in the reference (cuda/rope_kernel.cu:151)
  CHECK(input_pos.device_type() == base::DeviceType::kDeviceCPU);
  const int32_t pos = *input_pos.ptr<int32_t>(0);
  ok(kx_rope_f32(model_knobs().rope_variant, dim, kv_dim, head_size, out_ptr<float>(input_q),
                 out_ptr<float>(input_k), kv_dim, pos, 1, sin_cache.ptr<float>(),
                 cos_cache.ptr<float>(), stream),
     "kx_rope_f32");
}

void swiglu_kernel_hip(const tensor::Tensor& input1, const tensor::Tensor& input2,
                       const tensor::Tensor& output, void* stream) {
  CHECK(!input1.is_empty() && !input2.is_empty() && !output.is_empty());
  CHECK_EQ(input1.size(), input2.size());
  ok(kx_swiglu_f32(input1.ptr<float>(), input2.ptr<float>(), out_ptr<float>(output),
                   static_cast<int32_t>(input1.size()), stream),
     "kx_swiglu_f32");
}

// The pieces of the reference's CPU attention (cpu/mha_kernel.cpp:38-59), each one launch here.
void scale_inplace_hip(float scale, const tensor::Tensor& input, void* stream) {
  CHECK(!input.is_empty());  // cpu/scale_kernel.cpp:5
  ok(kx_scale_f32(scale, out_ptr<float>(input), static_cast<int32_t>(input.size()), stream),
     "kx_scale_f32");
}

void softmax_inplace_hip(const tensor::Tensor& input, void* stream) {
  CHECK(!input.is_empty());
  ok(kx_softmax_f32(out_ptr<float>(input), static_cast<int32_t>(input.size()), stream),
     "kx_softmax_f32");
}

void scale_sum_kernel_hip(const tensor::Tensor& value, const tensor::Tensor& scale,
                          const tensor::Tensor& output, int pos, int size, int stride, void* stream) {
  CHECK_EQ(value.is_empty(), false);  // cpu/scale_sum_kernel.cpp:9-13
  CHECK_EQ(scale.is_empty(), false);
  CHECK_EQ(output.is_empty(), false);
  CHECK_EQ(static_cast<size_t>(size), value.size());
  CHECK_EQ(static_cast<size_t>(size), output.size());
  ok(kx_scale_sum_f32(value.ptr<float>(), scale.ptr<float>(), out_ptr<float>(output), pos, size, stride,
                      stream),
     "kx_scale_sum_f32");
}

void flash_attention_kernel_hip(const tensor::Tensor& query, const tensor::Tensor& key,
                                const tensor::Tensor& value, const tensor::Tensor& output,
                                int32_t head_num, int32_t kv_head_num, int32_t head_size,
                                int32_t seq_len, int32_t pos, int32_t layer_idx, float softmax_scale,
                                bool is_causal, base::DeviceType device_type, CudaConfig* config) {
  UNUSED(device_type);
  CHECK(config != nullptr);
  CHECK(head_num > 0 && head_size > 0);
  CHECK_EQ(query.size() % (static_cast<size_t>(head_num) * head_size), 0u);
  const int32_t tq = static_cast<int32_t>(query.size() / (static_cast<size_t>(head_num) * head_size));
  // The real layer index and GQA ratio are passed on; the reference's dispatch wrapper drops
  // both (cpu/flash_attention_kernel.cpp:47,64,78 pass layer 0 and kv_mul 1).
  ok(kx_flash_attention_f32(query.ptr<float>(), key.ptr<float>(), value.ptr<float>(),
                            out_ptr<float>(output), tq, head_num, kv_head_num, head_size, seq_len,
                            pos, layer_idx, softmax_scale, is_causal ? 1 : 0,
                            config->attn_workspace, config->attn_workspace_bytes, config->stream),
     "kx_flash_attention_f32");
}

void sin_cos_cache_calc_hip(int head_size, int max_seq_len, float theta,
                            const tensor::Tensor& sin_cache, const tensor::Tensor& cos_cache,
                            void* stream) {
  CHECK(!sin_cache.is_empty() && !cos_cache.is_empty());
  ok(kx_sin_cos_cache_f32(head_size, max_seq_len, theta, out_ptr<float>(sin_cache),
                          out_ptr<float>(cos_cache), stream),
     "kx_sin_cos_cache_f32");
}
}  // namespace kernel
