// Tensor-parallel communication handle.  Mirrors NcclHandler
// (kuiper/include/base/nccl_handler.h:29-58): a process-wide singleton with Init / AllReduce /
// Broadcast / AllGather returning Status, world_size <= 1 turning every collective into a no-op
// (nccl_handler.cpp:86-93).  Differences: the communicator is really built (RCCL over xGMI,
// through the C ABI), and Init takes the 128-byte unique id that rank 0 produced -- the
// out-of-band exchange the reference leaves as a comment (nccl_handler.cpp:60-71).
#ifndef KX_HOST_BASE_RCCL_HANDLER_H_
#define KX_HOST_BASE_RCCL_HANDLER_H_
#include <array>
#include <memory>
#include <string>

#include "base/base.h"

namespace base {
using CommUniqueId = std::array<unsigned char, 128>;

class RcclHandler {
 public:
  RcclHandler() = default;
  ~RcclHandler();
  RcclHandler(const RcclHandler&) = delete;
  RcclHandler& operator=(const RcclHandler&) = delete;

  static std::shared_ptr<RcclHandler> get_instance();

  // Rank 0 calls this and ships the bytes to every other rank by any side channel.
  static Status CreateUniqueId(CommUniqueId* id);

  Status Init(int32_t rank, int32_t world_size, const CommUniqueId* id = nullptr);
  Status AllReduce(const void* sendbuff, void* recvbuff, size_t count, DataType datatype,
                   void* stream = nullptr);
  Status Broadcast(const void* sendbuff, void* recvbuff, size_t count, DataType datatype,
                   int32_t root, void* stream = nullptr);
  Status AllGather(const void* sendbuff, void* recvbuff, size_t send_count, DataType datatype,
                   void* stream = nullptr);

  int32_t rank() const { return rank_; }
  int32_t world_size() const { return world_size_; }
  void* raw_comm() const { return comm_; }

 private:
  int32_t rank_ = 0;
  int32_t world_size_ = 1;
  void* comm_ = nullptr;
};
// reference spelling for drop-in call sites
using NcclHandler = RcclHandler;
}  // namespace base
#endif
