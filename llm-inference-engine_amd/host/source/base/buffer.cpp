// kuiper/source/base/buffer.cpp:5-141 restated for the CPU/HIP device pair.
#include "base/buffer.h"

namespace base {
Buffer::Buffer(size_t byte_size, std::shared_ptr<DeviceAllocator> allocator, void* ptr,
               bool use_external)
    : byte_size_(byte_size), ptr_(ptr), use_external_(use_external), allocator_(std::move(allocator)) {
  if (!ptr_ && allocator_ && byte_size_ > 0) {
    device_type_ = allocator_->device_type();
    use_external_ = false;
    ptr_ = allocator_->allocate(byte_size_);
  }
}

Buffer::~Buffer() {
  if (!use_external_ && ptr_ && allocator_) {
    allocator_->release(ptr_);
    ptr_ = nullptr;
  }
}

bool Buffer::allocate() {
  if (byte_size_ == 0) {
    ptr_ = nullptr;
    return true;
  }
  if (!allocator_) return false;
  use_external_ = false;
  ptr_ = allocator_->allocate(byte_size_);
  return ptr_ != nullptr;
}

static MemcpyKind kind_for(DeviceType src, DeviceType dst) {
  const bool s = src == DeviceType::kDeviceCPU, d = dst == DeviceType::kDeviceCPU;
  if (s && d) return MemcpyKind::kMemcpyCPU2CPU;
  if (!s && d) return MemcpyKind::kMemcpyHIP2CPU;
  if (s && !d) return MemcpyKind::kMemcpyCPU2HIP;
  return MemcpyKind::kMemcpyHIP2HIP;
}

void Buffer::copy_from(const Buffer* buffer) const {
  CHECK(allocator_ != nullptr);
  CHECK(buffer != nullptr && buffer->ptr_ != nullptr);
  const size_t n = byte_size_ < buffer->byte_size_ ? byte_size_ : buffer->byte_size_;
  if (!n) return;
  CHECK(buffer->device_type() != DeviceType::kDeviceUnknown &&
        device_type() != DeviceType::kDeviceUnknown);
  allocator_->memcpy(buffer->ptr_, ptr_, n, kind_for(buffer->device_type(), device_type()));
}

void Buffer::copy_from(const Buffer& buffer) const { copy_from(&buffer); }
}  // namespace base
