// Dense tensor: dims + dtype + shared Buffer.  Mirrors kuiper/include/tensor/tensor.h:12-141
// (same constructors and accessors); to_cuda() keeps its name for drop-in call sites and moves
// the data to the HIP device.
#ifndef KX_HOST_TENSOR_TENSOR_H_
#define KX_HOST_TENSOR_TENSOR_H_
#include <memory>
#include <vector>

#include "base/base.h"
#include "base/buffer.h"

namespace tensor {
class Tensor {
 public:
  Tensor() = default;
  explicit Tensor(base::DataType data_type, int32_t dim0, bool need_alloc = false,
                  std::shared_ptr<base::DeviceAllocator> alloc = nullptr, void* ptr = nullptr);
  explicit Tensor(base::DataType data_type, int32_t dim0, int32_t dim1, bool need_alloc = false,
                  std::shared_ptr<base::DeviceAllocator> alloc = nullptr, void* ptr = nullptr);
  explicit Tensor(base::DataType data_type, int32_t dim0, int32_t dim1, int32_t dim2,
                  bool need_alloc = false, std::shared_ptr<base::DeviceAllocator> alloc = nullptr,
                  void* ptr = nullptr);
  explicit Tensor(base::DataType data_type, int32_t dim0, int32_t dim1, int32_t dim2, int32_t dim3,
                  bool need_alloc = false, std::shared_ptr<base::DeviceAllocator> alloc = nullptr,
                  void* ptr = nullptr);
  explicit Tensor(base::DataType data_type, std::vector<int32_t> dims, bool need_alloc = false,
                  std::shared_ptr<base::DeviceAllocator> alloc = nullptr, void* ptr = nullptr);

  void to_cpu();
  void to_hip(void* stream = nullptr);
  void to_cuda(void* stream = nullptr) { to_hip(stream); }

  bool is_empty() const;
  void init_buffer(std::shared_ptr<base::DeviceAllocator> alloc, base::DataType data_type,
                   bool need_alloc, void* ptr);
  void reshape(const std::vector<int32_t>& dims);
  std::shared_ptr<base::Buffer> get_buffer() const { return buffer_; }
  size_t size() const { return size_; }
  size_t byte_size() const { return size_ * base::DataTypeSize(data_type_); }
  int32_t dims_size() const { return static_cast<int32_t>(dims_.size()); }
  base::DataType data_type() const { return data_type_; }
  int32_t get_dim(int32_t idx) const;
  const std::vector<int32_t>& dims() const { return dims_; }
  std::vector<size_t> strides() const;
  bool assign(std::shared_ptr<base::Buffer> buffer);
  void reset(base::DataType data_type, const std::vector<int32_t>& dims);
  void set_device_type(base::DeviceType device_type) const;
  base::DeviceType device_type() const;
  bool allocate(std::shared_ptr<base::DeviceAllocator> allocator, bool need_realloc = false);
  Tensor clone() const;

  template <typename T>
  T* ptr() {
    return buffer_ ? reinterpret_cast<T*>(buffer_->ptr()) : nullptr;
  }
  template <typename T>
  const T* ptr() const {
    return buffer_ ? reinterpret_cast<const T*>(buffer_->ptr()) : nullptr;
  }
  template <typename T>
  T* ptr(int64_t index) {
    CHECK(buffer_ != nullptr && buffer_->ptr() != nullptr)
        << "The data area buffer of this tensor is empty or it points to a null pointer.";
    return reinterpret_cast<T*>(buffer_->ptr()) + index;
  }
  template <typename T>
  const T* ptr(int64_t index) const {
    CHECK(buffer_ != nullptr && buffer_->ptr() != nullptr)
        << "The data area buffer of this tensor is empty or it points to a null pointer.";
    return reinterpret_cast<const T*>(buffer_->ptr()) + index;
  }
  // Host-side element access (valid for CPU tensors only, as in the reference).
  template <typename T>
  T& index(int64_t offset) {
    CHECK_GE(offset, 0);
    CHECK_LT(offset, static_cast<int64_t>(size_));
    return *(reinterpret_cast<T*>(buffer_->ptr()) + offset);
  }
  template <typename T>
  const T& index(int64_t offset) const {
    CHECK_GE(offset, 0);
    CHECK_LT(offset, static_cast<int64_t>(size_));
    return *(reinterpret_cast<const T*>(buffer_->ptr()) + offset);
  }

 private:
  void construct(bool need_alloc, std::shared_ptr<base::DeviceAllocator> alloc, void* ptr);
  size_t size_ = 0;
  std::vector<int32_t> dims_;
  std::shared_ptr<base::Buffer> buffer_;
  base::DataType data_type_ = base::DataType::kDataTypeUnknown;
};
}  // namespace tensor
#endif
