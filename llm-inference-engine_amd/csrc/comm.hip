// Tensor-parallel communicator over RCCL (xGMI inside a node).
// Stands in for NcclHandler (kuiper/source/base/nccl_handler.cpp:24-181).  The reference never
// builds a communicator (ncclCommInitRank is commented out, :66-71); here rank 0 makes the
// unique id, the caller carries its 128 bytes to the other ranks, and every rank joins.
//
// librccl is resolved at first use with dlopen so a single-GPU process never loads it, and so a
// process that already carries an RCCL (e.g. through PyTorch) shares that one copy.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "kx_common.hpp"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  if (r.handle) return r;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) return r;
#define KX_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym))
  KX_SYM(GetUniqueId, "ncclGetUniqueId");
  KX_SYM(CommInitRank, "ncclCommInitRank");
  KX_SYM(CommDestroy, "ncclCommDestroy");
  KX_SYM(AllReduce, "ncclAllReduce");
  KX_SYM(Broadcast, "ncclBroadcast");
  KX_SYM(AllGather, "ncclAllGather");
#undef KX_SYM
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Broadcast &&
         r.AllGather;
  return r;
}

// ncclResult_t values are reported offset so they cannot be confused with hipError_t codes.
constexpr int kNcclBase = 10000;
inline int nccl_rc(ncclResult_t r) { return r == ncclSuccess ? KX_OK : kNcclBase + (int)r; }

}  // namespace

extern "C" int kx_comm_unique_id(void* id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "unique id is carried as 128 bytes");
  if (!id128) return KX_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return KX_EUNSUPPORTED;
  ncclUniqueId id;
  ncclResult_t rc = r.GetUniqueId(&id);
  if (rc != ncclSuccess) return nccl_rc(rc);
  memcpy(id128, &id, sizeof(id));
  return KX_OK;
}

extern "C" int kx_comm_init_rank(void** comm, const void* id128, int32_t world_size, int32_t rank) {
  if (!comm || !id128 || world_size < 1 || rank < 0 || rank >= world_size) return KX_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return KX_EUNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t rc = r.CommInitRank(&c, world_size, id, rank);
  if (rc != ncclSuccess) return nccl_rc(rc);
  *comm = c;
  return KX_OK;
}

extern "C" int kx_comm_destroy(void* comm) {
  if (!comm) return KX_OK;
  return nccl_rc(rccl().CommDestroy((ncclComm_t)comm));
}

extern "C" int kx_comm_allreduce_sum_f32(void* comm, const float* sendbuf, float* recvbuf,
                                         size_t count, void* stream) {
  if (!comm || !sendbuf || !recvbuf) return KX_EINVAL;
  if (count == 0) return KX_OK;
  return nccl_rc(rccl().AllReduce(sendbuf, recvbuf, count, ncclFloat, ncclSum, (ncclComm_t)comm,
                                  (hipStream_t)stream));
}

extern "C" int kx_comm_broadcast_f32(void* comm, const float* sendbuf, float* recvbuf, size_t count,
                                     int32_t root, void* stream) {
  if (!comm || !sendbuf || !recvbuf) return KX_EINVAL;
  if (count == 0) return KX_OK;
  return nccl_rc(rccl().Broadcast(sendbuf, recvbuf, count, ncclFloat, root, (ncclComm_t)comm,
                                  (hipStream_t)stream));
}

extern "C" int kx_comm_allgather_f32(void* comm, const float* sendbuf, float* recvbuf,
                                     size_t send_count, void* stream) {
  if (!comm || !sendbuf || !recvbuf) return KX_EINVAL;
  if (send_count == 0) return KX_OK;
  return nccl_rc(rccl().AllGather(sendbuf, recvbuf, send_count, ncclFloat, (ncclComm_t)comm,
                                  (hipStream_t)stream));
}
