// In-process stand-in for the tensor-parallel communicator: N models living in ONE process (one
// host thread each, any devices -- in practice all on one GPU) sum their buffers through a host
// rendezvous.  It exists so the whole tensor-parallel token loop -- shard plan, residual owner,
// placement of the two all-reduces per layer, batched prompt pass -- can be checked numerically
// on a single GPU.  It synchronises the stream on the host, so it cannot be captured into a
// graph; production runs use RcclHandler (kuiper/include/base/nccl_handler.h:29-58).
#ifndef KX_HOST_BASE_LOOPBACK_COMM_H_
#define KX_HOST_BASE_LOOPBACK_COMM_H_
#include <condition_variable>
#include <cstddef>
#include <mutex>
#include <vector>

#include "base/base.h"

namespace base {

class LoopbackGroup {
 public:
  explicit LoopbackGroup(int ranks) : n_(ranks), bufs_(ranks, nullptr) {}
  int size() const { return n_; }
  // In-place sum over the ranks' buffers of `count` floats, in rank order; every rank calls it.
  Status all_reduce(int rank, float* buf, size_t count, void* stream);
  // In-place all-gather: rank r's `count` floats sit at buf + r*count; afterwards every rank's
  // buffer holds all the slices.
  Status all_gather(int rank, float* buf, size_t count, void* stream);

 private:
  void arrive();
  int n_;
  std::vector<float*> bufs_;
  std::mutex mu_;
  std::condition_variable cv_;
  int waiting_ = 0;
  long generation_ = 0;
};

}  // namespace base
#endif
