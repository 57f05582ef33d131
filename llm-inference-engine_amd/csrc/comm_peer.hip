// One-launch all-reduce / all-gather for the token loop's small vectors (a token's `dim` floats,
// the ranks' argmax candidates) over peer-mapped mailboxes, gfx950 inside one xGMI node.
//
// The reference has one collective, NcclHandler::AllReduce (kuiper/source/base/nccl_handler.cpp:
// 84-117), called behind every row-parallel matmul (kuiper/source/op/matmul.cpp:93-116).  For the
// 16-32 KB a decode step reduces, a library all-reduce is all latency (protocol hops, two stream
// events around the side stream, 2 x layers times per token).  Here every rank owns a mailbox in
// fine-grained device memory that all the other ranks map through IPC handles, and one kernel on
// the compute stream
//   1. writes the rank's vector into its slot of EVERY rank's mailbox (posted writes over xGMI),
//   2. publishes a sequence number behind them (system-scope release) in every mailbox,
//   3. waits until its own mailbox carries that number from every rank (system-scope acquire),
//   4. adds the slots in rank order -- the same order on every rank, so all ranks hold bitwise
//      the same sum -- into the caller's buffer.
// Slots and flags are doubled by the parity of the sequence number: a rank can be at most one call
// ahead of another (it cannot finish call n+1 without the other's contribution, which is sent only
// after the other finished reading call n), so call n+1 never lands on data still being read.
// The sequence number lives in device memory and is advanced by the kernel, so a launch captured
// into a HIP graph replays correctly.  Waits are bounded by wall-clock time; a wait that gives up
// raises a flag the host reads with kx_peer_timed_out() and the launch finishes (with a wrong sum)
// instead of hanging the queue; once the flag is up no later call waits at all.  A rank whose flag
// is up stops posting data and sequence numbers and writes a "dead" word into every mailbox
// instead; a rank that finds a dead word while it waits raises its own flag at once (and passes the
// word on the same way), so one rank's failure shows in EVERY rank's kx_peer_timed_out() within
// one call -- no rank is left trusting sums that another rank no longer takes part in.
#include <string.h>

#include "kx_common.hpp"

namespace kx {

constexpr int kPeerMax = 8;            // ranks in a group (one xGMI node)
constexpr int kPeerFlagStride = 64;    // bytes between two flags
constexpr unsigned long long kPeerWaitTicks = 400000000ull;  // 4 s of the 100 MHz wall counter

struct PeerDev {  // in the rank's own device memory; read by its kernels only
  char* box[kPeerMax];  // every rank's mailbox as mapped here (own included)
  unsigned seq;         // calls completed
  unsigned timed_out;   // a wait gave up
};

struct PeerGroup {
  int world = 0, rank = 0, device = 0;
  size_t cap = 0;        // floats per slot
  size_t flags_off = 0;  // byte offset of the flags inside a mailbox
  size_t box_bytes = 0;
  char* box_local = nullptr;
  void* opened[kPeerMax] = {};
  PeerDev* dev = nullptr;
  bool connected = false;
};

__device__ __forceinline__ float* peer_slot(char* box, unsigned par, int world, int src, size_t cap) {
  return reinterpret_cast<float*>(box) + ((size_t)par * world + src) * cap;
}
__device__ __forceinline__ unsigned* peer_flag(char* box, size_t flags_off, unsigned par, int world,
                                               int src) {
  return reinterpret_cast<unsigned*>(box + flags_off + ((size_t)par * world + src) * kPeerFlagStride);
}
// behind the 2 x world sequence flags: one "dead" word per source rank
__device__ __forceinline__ unsigned* peer_dead(char* box, size_t flags_off, int world, int src) {
  return reinterpret_cast<unsigned*>(box + flags_off + ((size_t)2 * world + src) * kPeerFlagStride);
}

// GATHER: buf holds world x count floats, this rank's part at buf + rank * count; every part is
// filled in.  Otherwise buf holds count floats and receives the sum over the ranks.
template <bool GATHER>
__global__ __launch_bounds__(1024) void peer_exchange_kernel(PeerDev* ctx, float* buf, int count,
                                                             int world, int rank, size_t cap,
                                                             size_t flags_off) {
  __shared__ unsigned s_seq, s_dead;
  if (threadIdx.x == 0) {
    s_seq = ctx->seq + 1;
    s_dead = ctx->timed_out;  // a wait gave up earlier: later calls do not wait again
  }
  __syncthreads();
  const unsigned seq = s_seq, par = seq & 1u;
  const bool dead = s_dead != 0;
  const float* mine = GATHER ? buf + (size_t)rank * count : buf;
  if (dead) {
    // out of the protocol: tell every rank so (again: cheap, and a rank that connected late sees it)
    if ((int)threadIdx.x < world)
      __hip_atomic_store(peer_dead(ctx->box[threadIdx.x], flags_off, world, rank), 1u, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) ctx->seq = seq;
    return;
  }
  // 1. this rank's vector into its slot of every mailbox (16-byte pieces where its rows allow)
  const int n4 = ((uintptr_t)mine % 16 == 0) ? count >> 2 : 0;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(mine + 4 * i);
    for (int p = 0; p < world; ++p)
      *reinterpret_cast<f32x4*>(peer_slot(ctx->box[p], par, world, rank, cap) + 4 * i) = v;
  }
  for (int i = 4 * n4 + threadIdx.x; i < count; i += blockDim.x) {
    const float v = mine[i];
    for (int p = 0; p < world; ++p) peer_slot(ctx->box[p], par, world, rank, cap)[i] = v;
  }
  // 2. every thread's writes are out before the sequence number goes out behind them
  __atomic_thread_fence(__ATOMIC_RELEASE);  // system scope
  __syncthreads();
  if ((int)threadIdx.x < world)
    __hip_atomic_store(peer_flag(ctx->box[threadIdx.x], flags_off, par, world, rank), seq,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // 3. wait for every rank's number in the own mailbox -- or for word that the rank has left
  __shared__ unsigned s_gave_up;
  if (threadIdx.x == 0) s_gave_up = 0;
  __syncthreads();
  if ((int)threadIdx.x < world) {
    unsigned* f = peer_flag(ctx->box[rank], flags_off, par, world, (int)threadIdx.x);
    unsigned* d = peer_dead(ctx->box[rank], flags_off, world, (int)threadIdx.x);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      if (__hip_atomic_load(d, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0 ||
          __builtin_amdgcn_s_memrealtime() - t0 > kPeerWaitTicks) {
        ctx->timed_out = 1;
        s_gave_up = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    // a source that has left counts even when its number for THIS call did arrive before it went
    if (__hip_atomic_load(d, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
      ctx->timed_out = 1;
      s_gave_up = 1;
    }
  }
  __syncthreads();
  if (s_gave_up && (int)threadIdx.x < world)  // this rank leaves too: pass the word on right away
    __hip_atomic_store(peer_dead(ctx->box[threadIdx.x], flags_off, world, rank), 1u, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  __atomic_thread_fence(__ATOMIC_ACQUIRE);  // system scope: the slots as the other ranks wrote them
  // 4. slots in rank order
  char* own = ctx->box[rank];
  if (GATHER) {
    for (int q = 0; q < world; ++q) {
      if (q == rank) continue;
      const float* s = peer_slot(own, par, world, q, cap);
      for (int i = threadIdx.x; i < count; i += blockDim.x) buf[(size_t)q * count + i] = s[i];
    }
  } else {
    for (int i = threadIdx.x; i < (count >> 2); i += blockDim.x) {
      f32x4 acc = *reinterpret_cast<const f32x4*>(peer_slot(own, par, world, 0, cap) + 4 * i);
      for (int q = 1; q < world; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(peer_slot(own, par, world, q, cap) + 4 * i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<f32x4*>(buf + 4 * i) = acc;
    }
    for (int i = 4 * (count >> 2) + threadIdx.x; i < count; i += blockDim.x) {
      float acc = peer_slot(own, par, world, 0, cap)[i];
      for (int q = 1; q < world; ++q) acc += peer_slot(own, par, world, q, cap)[i];
      buf[i] = acc;
    }
  }
  if (threadIdx.x == 0) ctx->seq = seq;
}

}  // namespace kx

using kx::PeerGroup;

extern "C" int kx_peer_create(void** group, int32_t world_size, int32_t rank, size_t max_floats) {
  if (!group || world_size < 1 || world_size > kx::kPeerMax || rank < 0 || rank >= world_size ||
      max_floats == 0)
    return KX_EINVAL;
  int device = 0;
  KX_HIP_TRY(hipGetDevice(&device));
  auto* g = new PeerGroup();
  g->world = world_size;
  g->rank = rank;
  g->device = device;
  g->cap = (max_floats + 63) / 64 * 64;
  g->flags_off = 2 * (size_t)world_size * g->cap * sizeof(float);
  g->box_bytes = g->flags_off + 3 * (size_t)world_size * kx::kPeerFlagStride;  // 2 x world flags + world dead words
  void* box = nullptr;
  // memory the other ranks write while this rank's kernel runs: fine-grained (coherent within a
  // kernel's lifetime), not the default coarse-grained kind (coherent at kernel boundaries only)
  if (hipExtMallocWithFlags(&box, g->box_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    delete g;
    return KX_EUNSUPPORTED;
  }
  g->box_local = static_cast<char*>(box);
  if (hipMemset(box, 0, g->box_bytes) != hipSuccess || hipMalloc(&g->dev, sizeof(kx::PeerDev)) != hipSuccess ||
      hipMemset(g->dev, 0, sizeof(kx::PeerDev)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(box);
    if (g->dev) (void)hipFree(g->dev);
    delete g;
    return (int)hipErrorOutOfMemory;
  }
  *group = g;
  return KX_OK;
}

extern "C" int kx_peer_handle(void* group, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle is carried as 64 bytes");
  auto* g = static_cast<PeerGroup*>(group);
  if (!g || !handle64) return KX_EINVAL;
  hipIpcMemHandle_t h;
  KX_HIP_TRY(hipIpcGetMemHandle(&h, g->box_local));
  memcpy(handle64, &h, sizeof(h));
  return KX_OK;
}

extern "C" int kx_peer_connect(void* group, const void* handles) {
  auto* g = static_cast<PeerGroup*>(group);
  if (!g || !handles || g->connected) return KX_EINVAL;
  kx::PeerDev d{};
  for (int p = 0; p < g->world; ++p) {
    if (p == g->rank) {
      d.box[p] = g->box_local;
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const char*>(handles) + (size_t)p * sizeof(h), sizeof(h));
    void* ptr = nullptr;
    KX_HIP_TRY(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
    g->opened[p] = ptr;
    d.box[p] = static_cast<char*>(ptr);
  }
  KX_HIP_TRY(hipMemcpy(g->dev, &d, sizeof(d), hipMemcpyHostToDevice));
  g->connected = true;
  return KX_OK;
}

static int peer_launch(PeerGroup* g, float* buf, size_t count, bool gather, void* stream) {
  if (!g || !buf) return KX_EINVAL;
  if (!g->connected) return KX_EINVAL;
  if (count == 0) return KX_OK;
  if (count > g->cap) return KX_ELIMIT;
  if (!gather && ((uintptr_t)buf % 16)) return KX_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int threads = count >= 4096 ? 1024 : 256;
  if (gather)
    hipLaunchKernelGGL(kx::peer_exchange_kernel<true>, dim3(1), dim3(threads), 0, s, g->dev, buf, (int)count,
                       g->world, g->rank, g->cap, g->flags_off);
  else
    hipLaunchKernelGGL(kx::peer_exchange_kernel<false>, dim3(1), dim3(threads), 0, s, g->dev, buf, (int)count,
                       g->world, g->rank, g->cap, g->flags_off);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_peer_allreduce_sum_f32(void* group, float* buf, size_t count, void* stream) {
  return peer_launch(static_cast<PeerGroup*>(group), buf, count, false, stream);
}

extern "C" int kx_peer_allgather_f32(void* group, float* buf, size_t count_per_rank, void* stream) {
  return peer_launch(static_cast<PeerGroup*>(group), buf, count_per_rank, true, stream);
}

extern "C" int kx_peer_timed_out(void* group, int32_t* flag) {
  auto* g = static_cast<PeerGroup*>(group);
  if (!g || !flag) return KX_EINVAL;
  kx::PeerDev d{};
  KX_HIP_TRY(hipMemcpy(&d, g->dev, sizeof(d), hipMemcpyDeviceToHost));
  *flag = d.timed_out ? 1 : 0;
  return KX_OK;
}

extern "C" int kx_peer_destroy(void* group) {
  auto* g = static_cast<PeerGroup*>(group);
  if (!g) return KX_OK;
  for (int p = 0; p < g->world; ++p)
    if (g->opened[p]) (void)hipIpcCloseMemHandle(g->opened[p]);
  if (g->box_local) (void)hipFree(g->box_local);
  if (g->dev) (void)hipFree(g->dev);
  delete g;
  return KX_OK;
}
