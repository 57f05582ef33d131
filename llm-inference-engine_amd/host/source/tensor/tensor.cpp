// kuiper/source/tensor/tensor.cpp restated.  One behaviour is deliberately different: when a
// constructor is given BOTH need_alloc=true and an external pointer, the reference's 2..4-dim
// constructors silently allocate and drop the pointer (tensor.cpp:57-61,71-75), which is how its
// KV-cache tensors end up disconnected from the KVCacheManager arena (llama3.cpp:498-504).
// Here an external pointer always wins: the tensor views that memory.
#include "tensor/tensor.h"

#include <numeric>

namespace tensor {
static size_t product(const std::vector<int32_t>& dims) {
  if (dims.empty()) return 0;
  size_t n = 1;
  for (int32_t d : dims) n *= static_cast<size_t>(d);
  return n;
}

void Tensor::construct(bool need_alloc, std::shared_ptr<base::DeviceAllocator> alloc, void* ptr) {
  size_ = product(dims_);
  if (ptr != nullptr) {
    // view of external memory; an allocator, if given, only tags the device type
    buffer_ = std::make_shared<base::Buffer>(byte_size(), nullptr, ptr, true);
    if (alloc) buffer_->set_device_type(alloc->device_type());
  } else if (need_alloc && alloc) {
    allocate(alloc);
  }
}

Tensor::Tensor(base::DataType data_type, int32_t dim0, bool need_alloc,
               std::shared_ptr<base::DeviceAllocator> alloc, void* ptr)
    : dims_{dim0}, data_type_(data_type) {
  construct(need_alloc, std::move(alloc), ptr);
}
Tensor::Tensor(base::DataType data_type, int32_t dim0, int32_t dim1, bool need_alloc,
               std::shared_ptr<base::DeviceAllocator> alloc, void* ptr)
    : dims_{dim0, dim1}, data_type_(data_type) {
  construct(need_alloc, std::move(alloc), ptr);
}
Tensor::Tensor(base::DataType data_type, int32_t dim0, int32_t dim1, int32_t dim2, bool need_alloc,
               std::shared_ptr<base::DeviceAllocator> alloc, void* ptr)
    : dims_{dim0, dim1, dim2}, data_type_(data_type) {
  construct(need_alloc, std::move(alloc), ptr);
}
Tensor::Tensor(base::DataType data_type, int32_t dim0, int32_t dim1, int32_t dim2, int32_t dim3,
               bool need_alloc, std::shared_ptr<base::DeviceAllocator> alloc, void* ptr)
    : dims_{dim0, dim1, dim2, dim3}, data_type_(data_type) {
  construct(need_alloc, std::move(alloc), ptr);
}
Tensor::Tensor(base::DataType data_type, std::vector<int32_t> dims, bool need_alloc,
               std::shared_ptr<base::DeviceAllocator> alloc, void* ptr)
    : dims_(std::move(dims)), data_type_(data_type) {
  construct(need_alloc, std::move(alloc), ptr);
}

void Tensor::init_buffer(std::shared_ptr<base::DeviceAllocator> alloc, base::DataType data_type,
                         bool need_alloc, void* ptr) {
  data_type_ = data_type;
  if (!alloc && !need_alloc) {
    buffer_ = std::make_shared<base::Buffer>(byte_size(), nullptr, ptr, true);
  } else {
    allocate(alloc, true);
  }
}

void Tensor::to_hip(void* stream) {
  CHECK(buffer_ != nullptr);
  const base::DeviceType dev = device_type();
  if (dev == base::DeviceType::kDeviceUnknown) {
    LOG(ERROR) << "The device type of the tensor is unknown.";
  } else if (dev == base::DeviceType::kDeviceCPU) {
    const size_t bytes = byte_size();
    auto alloc = base::HIPDeviceAllocatorFactory::get_instance();
    auto dev_buffer = std::make_shared<base::Buffer>(bytes, alloc);
    alloc->memcpy(buffer_->ptr(), dev_buffer->ptr(), bytes, base::MemcpyKind::kMemcpyCPU2HIP,
                  stream);
    buffer_ = dev_buffer;
  }
}

void Tensor::to_cpu() {
  CHECK(buffer_ != nullptr);
  const base::DeviceType dev = device_type();
  if (dev == base::DeviceType::kDeviceUnknown) {
    LOG(ERROR) << "The device type of the tensor is unknown.";
  } else if (dev == base::DeviceType::kDeviceHIP) {
    const size_t bytes = byte_size();
    auto alloc = base::CPUDeviceAllocatorFactory::get_instance();
    auto host_buffer = std::make_shared<base::Buffer>(bytes, alloc);
    alloc->memcpy(buffer_->ptr(), host_buffer->ptr(), bytes, base::MemcpyKind::kMemcpyHIP2CPU,
                  nullptr, true);
    buffer_ = host_buffer;
  }
}

int32_t Tensor::get_dim(int32_t idx) const {
  CHECK_GE(idx, 0);
  CHECK_LT(idx, static_cast<int32_t>(dims_.size()));
  return dims_[idx];
}

base::DeviceType Tensor::device_type() const {
  return buffer_ ? buffer_->device_type() : base::DeviceType::kDeviceUnknown;
}

void Tensor::set_device_type(base::DeviceType device_type) const {
  if (buffer_) buffer_->set_device_type(device_type);
}

bool Tensor::assign(std::shared_ptr<base::Buffer> buffer) {
  if (!buffer) {
    LOG(ERROR) << "The buffer parameter in the assign function is null pointer!";
    return false;
  }
  if (buffer_ && buffer_->device_type() != buffer->device_type()) {
    LOG(ERROR) << "The device type of the new buffer is different from the original one.";
  }
  if (byte_size() > buffer->byte_size()) {
    LOG(ERROR) << "The size of buffer is too small for the tensor!";
    return false;
  }
  buffer_ = std::move(buffer);
  return true;
}

bool Tensor::allocate(std::shared_ptr<base::DeviceAllocator> allocator, bool need_realloc) {
  if (!allocator) {
    LOG(ERROR) << "The allocator parameter in the allocate function is null pointer!";
    return false;
  }
  const size_t bytes = byte_size();
  if (!bytes) {
    buffer_ = std::make_shared<base::Buffer>(0, allocator, nullptr, false);
    buffer_->set_device_type(allocator->device_type());
    return true;
  }
  if (buffer_ && bytes <= buffer_->byte_size() && !need_realloc) return true;
  buffer_ = std::make_shared<base::Buffer>(bytes, allocator, nullptr);
  if (!buffer_->ptr()) {
    LOG(ERROR) << "The memory allocated is a null pointer!";
    return false;
  }
  return true;
}

void Tensor::reset(base::DataType data_type, const std::vector<int32_t>& dims) {
  data_type_ = data_type;
  dims_ = dims;
  size_ = product(dims);
  buffer_ = nullptr;
}

void Tensor::reshape(const std::vector<int32_t>& dims) {
  const size_t n = product(dims);
  if (buffer_ && n > size_) {
    auto grown = std::make_shared<base::Buffer>(n * base::DataTypeSize(data_type_),
                                                buffer_->allocator());
    CHECK(grown->ptr() != nullptr || n == 0);
    grown->copy_from(buffer_.get());
    buffer_ = grown;
  }
  dims_ = dims;
  size_ = n;
}

Tensor Tensor::clone() const {
  Tensor t = *this;
  auto alloc = buffer_->allocator();
  CHECK(alloc != nullptr) << "clone() needs an owning tensor";
  t.buffer_ = std::make_shared<base::Buffer>(byte_size(), alloc);
  t.buffer_->copy_from(buffer_.get());
  return t;
}

std::vector<size_t> Tensor::strides() const {
  std::vector<size_t> s(dims_.size(), 1);
  for (int i = static_cast<int>(dims_.size()) - 2; i >= 0; --i)
    s[i] = s[i + 1] * static_cast<size_t>(dims_[i + 1]);
  return s;
}

bool Tensor::is_empty() const {
  return size_ == 0 || buffer_ == nullptr || buffer_->ptr() == nullptr;
}
}  // namespace tensor
