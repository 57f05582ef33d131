// Pre-allocated KV cache arena.  Mirrors KVCacheManager / KVCacheConfig
// (kuiper/include/base/kv_cache_manager.h:12-65): one allocation = K block then V block,
// zero-filled once, with per-layer pointers handed out.
//
// Layout (what every attention kernel indexes -- cpu/mha_kernel.cpp:10, cuda/mha_kernel.cu:115,
// qwen2.cpp:473-476): each block is [batch][layer][max_seq_len][kv_dim] fp32 with
// kv_dim = kv_head_num * head_size, so layer stride = max_seq_len * kv_dim.  The reference's
// manager multiplies the layer stride by kv_head_num once more (kv_cache_manager.cpp:26-33,
// 63-72), which over-allocates kv_head_num times and disagrees with its own kernels; that is
// not reproduced.
#ifndef KX_HOST_BASE_KV_CACHE_MANAGER_H_
#define KX_HOST_BASE_KV_CACHE_MANAGER_H_
#include <memory>
#include <utility>

#include "base/alloc.h"
#include "base/base.h"

namespace base {
struct KVCacheConfig {
  int32_t max_batch_size = 1;
  int32_t max_seq_len = 0;
  int32_t layer_num = 0;
  int32_t kv_head_num = 0;
  int32_t kv_dim = 0;  // kv_head_num * head_size (the row length of the cache)
  DataType dtype = DataType::kDataTypeFp32;
};

class KVCacheManager {
 public:
  explicit KVCacheManager(DeviceType device_type, const KVCacheConfig& config);
  ~KVCacheManager();
  KVCacheManager(const KVCacheManager&) = delete;
  KVCacheManager& operator=(const KVCacheManager&) = delete;

  Status allocate();
  // {k_ptr, v_ptr} of one layer (row 0 of that layer's [max_seq_len][kv_dim] block)
  std::pair<void*, void*> get_cache_ptr(int32_t layer_idx, int32_t batch_idx = 0);
  void* get_raw_buffer_k() { return raw_buffer_; }
  void* get_raw_buffer_v();
  size_t total_bytes() const { return total_bytes_; }
  size_t layer_stride_elements() const {
    return static_cast<size_t>(config_.max_seq_len) * config_.kv_dim;
  }
  const KVCacheConfig& config() const { return config_; }

 private:
  DeviceType device_type_;
  KVCacheConfig config_;
  void* raw_buffer_ = nullptr;
  size_t total_bytes_ = 0;
  std::shared_ptr<DeviceAllocator> allocator_;
};
}  // namespace base
#endif
