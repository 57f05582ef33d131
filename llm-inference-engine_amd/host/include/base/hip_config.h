// Stream pair (+ attention workspace) every layer carries.
// Mirrors kernel::CudaConfig (kuiper/include/base/cuda_config.h:6-17): {stream, comm_stream},
// destroyed with the config.  Streams are opaque (void*) because host code never includes HIP
// headers; they are created/destroyed through the C ABI (include/kx_hip.h).
#ifndef KX_HOST_BASE_HIP_CONFIG_H_
#define KX_HOST_BASE_HIP_CONFIG_H_
#include <cstddef>

namespace kernel {
struct HipConfig {
  void* stream = nullptr;
  void* comm_stream = nullptr;
  // events reused by every tensor-parallel all-reduce (the reference creates and destroys a
  // pair per call, kuiper/source/op/matmul.cpp:95-120)
  void* compute_done = nullptr;
  void* comm_done = nullptr;
  // scratch for split-KV attention partials; sized by kx_attn_workspace_bytes
  void* attn_workspace = nullptr;
  size_t attn_workspace_bytes = 0;

  HipConfig() = default;
  HipConfig(const HipConfig&) = delete;
  HipConfig& operator=(const HipConfig&) = delete;
  ~HipConfig();

  // Creates both streams and the event pair on the current device.  Returns 0 or a C-ABI code.
  int create_streams();
  int reserve_attn_workspace(int head_num, int kv_head_num, int head_size, int seq_len);
};
// The registry typedefs keep the reference's spelling (kernels_interface.h:9-54).
using CudaConfig = HipConfig;
}  // namespace kernel
#endif
