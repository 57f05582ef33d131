// Allocators over malloc and over the HIP C ABI.
// Behaviour follows kuiper/source/base/alloc.cpp (memcpy kinds, optional stream, optional sync)
// and alloc_cpu.cpp / alloc_cu.cpp (allocate / release), minus the CUDA allocator's block cache.
#include "base/alloc.h"

#include <cstdlib>
#include <cstring>

#include "kx_hip.h"

namespace base {

static void hip_check(int rc, const char* what) {
  if (rc != 0) LOG(FATAL) << what << " failed: " << kx_error_string(rc) << " (" << rc << ")";
}

void DeviceAllocator::memcpy(const void* src_ptr, void* dest_ptr, size_t byte_size,
                             MemcpyKind memcpy_kind, void* stream, bool need_sync) const {
  CHECK_NE(src_ptr, nullptr);
  CHECK_NE(dest_ptr, nullptr);
  if (!byte_size) return;
  switch (memcpy_kind) {
    case MemcpyKind::kMemcpyCPU2CPU:
      std::memcpy(dest_ptr, src_ptr, byte_size);
      break;
    case MemcpyKind::kMemcpyCPU2HIP:
      hip_check(kx_memcpy_h2d(dest_ptr, src_ptr, byte_size, stream), "memcpy host->device");
      break;
    case MemcpyKind::kMemcpyHIP2CPU:
      hip_check(kx_memcpy_d2h(dest_ptr, src_ptr, byte_size, stream), "memcpy device->host");
      break;
    case MemcpyKind::kMemcpyHIP2HIP:
      hip_check(kx_memcpy_d2d(dest_ptr, src_ptr, byte_size, stream), "memcpy device->device");
      break;
    default:
      LOG(FATAL) << "Unknown memcpy kind: " << int(memcpy_kind);
  }
  if (need_sync && memcpy_kind != MemcpyKind::kMemcpyCPU2CPU) hip_check(kx_device_sync(), "sync");
}

void DeviceAllocator::memset_zero(void* ptr, size_t byte_size, void* stream, bool need_sync) {
  CHECK(device_type_ != DeviceType::kDeviceUnknown);
  if (!byte_size) return;
  if (device_type_ == DeviceType::kDeviceCPU) {
    std::memset(ptr, 0, byte_size);
  } else {
    hip_check(kx_memset(ptr, 0, byte_size, stream), "memset");
    if (need_sync) hip_check(kx_device_sync(), "sync");
  }
}

void* CPUDeviceAllocator::allocate(size_t byte_size) const {
  if (!byte_size) return nullptr;
  void* p = nullptr;
  // 64-byte alignment so host staging buffers can be read with vector loads after upload
  if (posix_memalign(&p, 64, byte_size) != 0) return nullptr;
  return p;
}
void CPUDeviceAllocator::release(void* ptr) const {
  if (ptr) std::free(ptr);
}

void* HIPDeviceAllocator::allocate(size_t byte_size) const {
  if (!byte_size) return nullptr;
  void* p = nullptr;
  const int rc = kx_malloc(&p, byte_size);
  if (rc != 0) {
    LOG(ERROR) << "device allocation of " << byte_size << " bytes failed: " << kx_error_string(rc);
    return nullptr;
  }
  return p;
}
void HIPDeviceAllocator::release(void* ptr) const {
  if (ptr) hip_check(kx_free(ptr), "free");
}

std::shared_ptr<CPUDeviceAllocator> CPUDeviceAllocatorFactory::get_instance() {
  static std::shared_ptr<CPUDeviceAllocator> inst = std::make_shared<CPUDeviceAllocator>();
  return inst;
}
std::shared_ptr<HIPDeviceAllocator> HIPDeviceAllocatorFactory::get_instance() {
  static std::shared_ptr<HIPDeviceAllocator> inst = std::make_shared<HIPDeviceAllocator>();
  return inst;
}
}  // namespace base
