// Device allocators.  Mirrors kuiper/include/base/alloc.h:7-92: DeviceAllocator with
// allocate/release/memcpy/memset_zero, a CPU allocator and a device allocator, each reached
// through a factory singleton.  The device allocator is HIP through the C ABI and does not pool
// (the reference's CUDA allocator caches freed blocks, alloc_cu.cpp; the token path here
// allocates once up front, so there is nothing to recycle).
#ifndef KX_HOST_BASE_ALLOC_H_
#define KX_HOST_BASE_ALLOC_H_
#include <memory>

#include "base/base.h"

namespace base {
enum class MemcpyKind {
  kMemcpyCPU2CPU = 0,
  kMemcpyCPU2HIP = 1,
  kMemcpyHIP2CPU = 2,
  kMemcpyHIP2HIP = 3,
  // reference spellings (alloc.h:8-12) for drop-in call sites
  kMemcpyCPU2CUDA = kMemcpyCPU2HIP,
  kMemcpyCUDA2CPU = kMemcpyHIP2CPU,
  kMemcpyCUDA2CUDA = kMemcpyHIP2HIP,
};

class DeviceAllocator {
 public:
  explicit DeviceAllocator(DeviceType device_type) : device_type_(device_type) {}
  virtual ~DeviceAllocator() = default;
  virtual DeviceType device_type() const { return device_type_; }
  virtual void release(void* ptr) const = 0;
  virtual void* allocate(size_t byte_size) const = 0;
  virtual void memcpy(const void* src_ptr, void* dest_ptr, size_t byte_size,
                      MemcpyKind memcpy_kind = MemcpyKind::kMemcpyCPU2CPU, void* stream = nullptr,
                      bool need_sync = false) const;
  virtual void memset_zero(void* ptr, size_t byte_size, void* stream, bool need_sync = false);

 private:
  DeviceType device_type_ = DeviceType::kDeviceUnknown;
};

class CPUDeviceAllocator : public DeviceAllocator {
 public:
  CPUDeviceAllocator() : DeviceAllocator(DeviceType::kDeviceCPU) {}
  void* allocate(size_t byte_size) const override;
  void release(void* ptr) const override;
};

class HIPDeviceAllocator : public DeviceAllocator {
 public:
  HIPDeviceAllocator() : DeviceAllocator(DeviceType::kDeviceHIP) {}
  void* allocate(size_t byte_size) const override;
  void release(void* ptr) const override;
};

class CPUDeviceAllocatorFactory {
 public:
  static std::shared_ptr<CPUDeviceAllocator> get_instance();
};
class HIPDeviceAllocatorFactory {
 public:
  static std::shared_ptr<HIPDeviceAllocator> get_instance();
};
}  // namespace base
#endif
