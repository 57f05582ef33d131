#include "base/kv_cache_manager.h"

namespace base {
KVCacheManager::KVCacheManager(DeviceType device_type, const KVCacheConfig& config)
    : device_type_(device_type), config_(config) {
  if (device_type_ == DeviceType::kDeviceCPU) {
    allocator_ = CPUDeviceAllocatorFactory::get_instance();
  } else if (device_type_ == DeviceType::kDeviceHIP) {
    allocator_ = HIPDeviceAllocatorFactory::get_instance();
  }
}

KVCacheManager::~KVCacheManager() {
  if (raw_buffer_ && allocator_) allocator_->release(raw_buffer_);
}

Status KVCacheManager::allocate() {
  if (!allocator_) return error::InternalError("Allocator not initialized in KVCacheManager");
  if (config_.max_seq_len <= 0 || config_.layer_num <= 0 || config_.kv_dim <= 0 ||
      config_.max_batch_size <= 0)
    return error::InvalidArgument("KVCacheConfig has a non-positive dimension");
  const size_t per_block = static_cast<size_t>(config_.max_batch_size) * config_.layer_num *
                           layer_stride_elements();
  total_bytes_ = 2 * per_block * DataTypeSize(config_.dtype);
  raw_buffer_ = allocator_->allocate(total_bytes_);
  if (!raw_buffer_) return error::InternalError("Failed to allocate KV cache memory");
  allocator_->memset_zero(raw_buffer_, total_bytes_, nullptr, true);
  return error::Success();
}

void* KVCacheManager::get_raw_buffer_v() {
  if (!raw_buffer_) return nullptr;
  return static_cast<int8_t*>(raw_buffer_) + total_bytes_ / 2;
}

std::pair<void*, void*> KVCacheManager::get_cache_ptr(int32_t layer_idx, int32_t batch_idx) {
  if (!raw_buffer_) return {nullptr, nullptr};
  CHECK(layer_idx >= 0 && layer_idx < config_.layer_num);
  CHECK(batch_idx >= 0 && batch_idx < config_.max_batch_size);
  const size_t esz = DataTypeSize(config_.dtype);
  const size_t off =
      (static_cast<size_t>(batch_idx) * config_.layer_num + layer_idx) * layer_stride_elements() * esz;
  int8_t* k = static_cast<int8_t*>(raw_buffer_) + off;
  int8_t* v = static_cast<int8_t*>(get_raw_buffer_v()) + off;
  return {k, v};
}
}  // namespace base
