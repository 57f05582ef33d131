#include "base/loopback_comm.h"

#include "kx_hip.h"

namespace base {

void LoopbackGroup::arrive() {
  std::unique_lock<std::mutex> lk(mu_);
  const long gen = generation_;
  if (++waiting_ == n_) {
    waiting_ = 0;
    ++generation_;
    cv_.notify_all();
  } else {
    cv_.wait(lk, [&] { return generation_ != gen; });
  }
}

Status LoopbackGroup::all_reduce(int rank, float* buf, size_t count, void* stream) {
  if (rank < 0 || rank >= n_ || !buf) return error::InvalidArgument("bad loopback rank/buffer");
  if (count > 0x7fffffffu) return error::InvalidArgument("loopback all-reduce too large");
  int rc = kx_stream_sync(stream);
  bufs_[rank] = buf;
  arrive();  // every rank's partial is complete and published
  if (rc == 0 && rank == 0) {
    for (int r = 1; r < n_ && rc == 0; ++r)
      rc = kx_add_f32(bufs_[0], bufs_[r], bufs_[0], (int32_t)count, stream);
    if (rc == 0) rc = kx_stream_sync(stream);
  }
  arrive();  // the sum sits in rank 0's buffer
  if (rc == 0 && rank != 0) {
    rc = kx_memcpy_d2d(buf, bufs_[0], count * sizeof(float), stream);
    if (rc == 0) rc = kx_stream_sync(stream);
  }
  arrive();  // nobody overwrites a buffer another rank still reads
  return rc == 0 ? error::Success() : error::InternalError("loopback all-reduce failed");
}

Status LoopbackGroup::all_gather(int rank, float* buf, size_t count, void* stream) {
  if (rank < 0 || rank >= n_ || !buf) return error::InvalidArgument("bad loopback rank/buffer");
  int rc = kx_stream_sync(stream);
  bufs_[rank] = buf;
  arrive();  // every rank's own slice is complete and published
  for (int r = 0; r < n_ && rc == 0; ++r)
    if (r != rank)
      rc = kx_memcpy_d2d(buf + (size_t)r * count, bufs_[r] + (size_t)r * count, count * sizeof(float), stream);
  if (rc == 0) rc = kx_stream_sync(stream);
  arrive();  // nobody overwrites a slice another rank still reads
  return rc == 0 ? error::Success() : error::InternalError("loopback all-gather failed");
}

}  // namespace base
