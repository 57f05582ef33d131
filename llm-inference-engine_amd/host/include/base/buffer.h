// Owning or external byte range.  Mirrors kuiper/include/base/buffer.h:6-43: a Buffer built
// around an external pointer never frees it (buffer.cpp:18-25), which is how weights alias the
// model blob until they are moved to the device.
#ifndef KX_HOST_BASE_BUFFER_H_
#define KX_HOST_BASE_BUFFER_H_
#include <memory>

#include "base/alloc.h"

namespace base {
class Buffer : public NoCopyable, public std::enable_shared_from_this<Buffer> {
 public:
  Buffer() = default;
  explicit Buffer(size_t byte_size, std::shared_ptr<DeviceAllocator> allocator = nullptr,
                  void* ptr = nullptr, bool use_external = false);
  virtual ~Buffer();

  bool allocate();
  void copy_from(const Buffer& buffer) const;
  void copy_from(const Buffer* buffer) const;
  void* ptr() { return ptr_; }
  const void* ptr() const { return ptr_; }
  size_t byte_size() const { return byte_size_; }
  std::shared_ptr<DeviceAllocator> allocator() const { return allocator_; }
  DeviceType device_type() const { return device_type_; }
  void set_device_type(DeviceType device_type) { device_type_ = device_type; }
  std::shared_ptr<Buffer> get_shared_from_this() { return shared_from_this(); }
  bool is_external() const { return use_external_; }

 private:
  size_t byte_size_ = 0;
  void* ptr_ = nullptr;
  bool use_external_ = false;
  DeviceType device_type_ = DeviceType::kDeviceUnknown;
  std::shared_ptr<DeviceAllocator> allocator_;
};
}  // namespace base
#endif
