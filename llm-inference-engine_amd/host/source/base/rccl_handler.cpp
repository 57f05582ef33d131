#include "base/rccl_handler.h"

#include "kx_hip.h"

namespace base {
static Status from_rc(int rc, const char* what) {
  if (rc == 0) return error::Success();
  return error::InternalError(std::string(what) + ": " + kx_error_string(rc) + " (" +
                              std::to_string(rc) + ")");
}

std::shared_ptr<RcclHandler> RcclHandler::get_instance() {
  static std::shared_ptr<RcclHandler> inst = std::make_shared<RcclHandler>();
  return inst;
}

RcclHandler::~RcclHandler() {
  if (comm_) kx_comm_destroy(comm_);
}

Status RcclHandler::CreateUniqueId(CommUniqueId* id) {
  if (!id) return error::InvalidArgument("null unique id");
  return from_rc(kx_comm_unique_id(id->data()), "RCCL unique id");
}

Status RcclHandler::Init(int32_t rank, int32_t world_size, const CommUniqueId* id) {
  rank_ = rank;
  world_size_ = world_size;
  // One rank and no id: nothing to set up (nccl_handler.cpp:27-29).  One rank WITH an id builds a
  // real single-rank communicator, so the collectives can be exercised on one GPU.
  if (world_size_ <= 1 && !id) return error::Success();
  if (!id) return error::InvalidArgument("world_size > 1 needs the unique id made on rank 0");
  if (comm_) {
    kx_comm_destroy(comm_);
    comm_ = nullptr;
  }
  return from_rc(kx_comm_init_rank(&comm_, id->data(), world_size, rank), "RCCL comm init");
}

Status RcclHandler::AllReduce(const void* sendbuff, void* recvbuff, size_t count, DataType datatype,
                              void* stream) {
  if (world_size_ <= 1 && !comm_) return error::Success();
  if (datatype != DataType::kDataTypeFp32)
    return error::InvalidArgument("Unsupported data type for AllReduce");
  return from_rc(kx_comm_allreduce_sum_f32(comm_, static_cast<const float*>(sendbuff),
                                           static_cast<float*>(recvbuff), count, stream),
                 "RCCL AllReduce");
}

Status RcclHandler::Broadcast(const void* sendbuff, void* recvbuff, size_t count, DataType datatype,
                              int32_t root, void* stream) {
  if (world_size_ <= 1 && !comm_) return error::Success();
  if (datatype != DataType::kDataTypeFp32)
    return error::InvalidArgument("Unsupported data type for Broadcast");
  return from_rc(kx_comm_broadcast_f32(comm_, static_cast<const float*>(sendbuff),
                                       static_cast<float*>(recvbuff), count, root, stream),
                 "RCCL Broadcast");
}

Status RcclHandler::AllGather(const void* sendbuff, void* recvbuff, size_t send_count,
                              DataType datatype, void* stream) {
  if (world_size_ <= 1 && !comm_) return error::Success();
  if (datatype != DataType::kDataTypeFp32)
    return error::InvalidArgument("Unsupported data type for AllGather");
  return from_rc(kx_comm_allgather_f32(comm_, static_cast<const float*>(sendbuff),
                                       static_cast<float*>(recvbuff), send_count, stream),
                 "RCCL AllGather");
}
}  // namespace base
