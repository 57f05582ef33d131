// Thin HIP runtime wrapper behind the C ABI: device memory, copies, streams, events, graphs.
// Stands in for the reference's CUDADeviceAllocator (kuiper/source/base/alloc_cu.cpp), the memcpy
// and memset paths of DeviceAllocator (alloc.cpp) and the stream pair of CudaConfig
// (kuiper/include/base/cuda_config.h:6-17).  No pooling: the token path allocates once.
#include <string.h>

#include <map>
#include <mutex>

#include "kx_common.hpp"

// ---- bounds-checking build: the table of live kx_malloc blocks and the violation record ----
#ifdef KX_BOUNDS
namespace kx {
namespace {
std::mutex g_bmu;
std::map<uintptr_t, size_t> g_blocks;  // base -> bytes
BoundsLog* g_blog = nullptr;
}  // namespace
void bounds_track(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_bmu);
  g_blocks[(uintptr_t)p] = bytes;
}
void bounds_untrack(const void* p) {
  std::lock_guard<std::mutex> lk(g_bmu);
  g_blocks.erase((uintptr_t)p);
}
BRange bounds_lookup(const void* p) {
  std::lock_guard<std::mutex> lk(g_bmu);
  auto it = g_blocks.upper_bound((uintptr_t)p);
  if (it == g_blocks.begin()) return BRange{nullptr, nullptr};
  --it;
  if ((uintptr_t)p >= it->first + it->second) return BRange{nullptr, nullptr};  // not one of ours
  return BRange{reinterpret_cast<const char*>(it->first), reinterpret_cast<const char*>(it->first + it->second)};
}
BoundsLog* bounds_log() {
  std::lock_guard<std::mutex> lk(g_bmu);
  if (!g_blog) {
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(BoundsLog)) != hipSuccess) return nullptr;
    (void)hipMemset(p, 0, sizeof(BoundsLog));
    g_blog = static_cast<BoundsLog*>(p);
  }
  return g_blog;
}
}  // namespace kx
#endif

extern "C" int kx_bounds_enabled(void) {
#ifdef KX_BOUNDS
  return 1;
#else
  return 0;
#endif
}

extern "C" int kx_bounds_report(kx_bounds_record* out, int32_t max_records, uint64_t* total) {
  if (total) *total = 0;
#ifdef KX_BOUNDS
  kx::BoundsLog* log = kx::bounds_log();
  if (!log) return KX_EINVAL;
  KX_HIP_TRY(hipDeviceSynchronize());
  static kx::BoundsLog host;
  KX_HIP_TRY(hipMemcpy(&host, log, sizeof(host), hipMemcpyDeviceToHost));
  if (total) *total = host.total;
  const uint64_t n = host.total < 1024 ? host.total : 1024;
  for (uint64_t i = 0; i < n && out && (int64_t)i < max_records; ++i) {
    out[i].file_id = (int32_t)(host.e[i].site >> 16);
    out[i].line = (int32_t)(host.e[i].site & 0xffff);
    out[i].operand = (int32_t)host.e[i].operand;
    out[i].offset = host.e[i].offset;
    out[i].extent = host.e[i].extent;
  }
  return KX_OK;
#else
  (void)out; (void)max_records;
  return KX_EUNSUPPORTED;
#endif
}

extern "C" int kx_bounds_reset(void) {
#ifdef KX_BOUNDS
  kx::BoundsLog* log = kx::bounds_log();
  if (!log) return KX_EINVAL;
  KX_HIP_TRY(hipDeviceSynchronize());
  KX_HIP_TRY(hipMemset(log, 0, sizeof(kx::BoundsLog)));
  return KX_OK;
#else
  return KX_EUNSUPPORTED;
#endif
}

extern "C" const char* kx_error_string(int code) {
  switch (code) {
    case KX_OK: return "ok";
    case KX_EINVAL: return "kx: invalid argument";
    case KX_EUNSUPPORTED: return "kx: unsupported shape";
    case KX_EWORKSPACE: return "kx: workspace too small";
    case KX_ELIMIT: return "kx: input exceeds the fixed scratch capacity of this entry point";
    default: break;
  }
  if (code >= 10000) return "kx: RCCL call failed (code - 10000 is the ncclResult_t)";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "kx: unknown error";
}

extern "C" int kx_device_count(int* count) {
  if (!count) return KX_EINVAL;
  KX_HIP_TRY(hipGetDeviceCount(count));
  return KX_OK;
}
extern "C" int kx_set_device(int device) {
  KX_HIP_TRY(hipSetDevice(device));
  return KX_OK;
}
extern "C" int kx_get_device(int* device) {
  if (!device) return KX_EINVAL;
  KX_HIP_TRY(hipGetDevice(device));
  return KX_OK;
}
extern "C" int kx_device_name(char* buf, size_t buf_bytes) {
  if (!buf || buf_bytes == 0) return KX_EINVAL;
  int dev = 0;
  KX_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  KX_HIP_TRY(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, buf_bytes, "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
           prop.multiProcessorCount);
  return KX_OK;
}

extern "C" int kx_malloc(void** ptr, size_t bytes) {
  if (!ptr) return KX_EINVAL;
  *ptr = nullptr;
  if (bytes == 0) return KX_OK;
  KX_HIP_TRY(hipMalloc(ptr, bytes));
#ifdef KX_BOUNDS
  kx::bounds_track(*ptr, bytes);
#endif
  return KX_OK;
}
extern "C" int kx_free(void* ptr) {
  if (!ptr) return KX_OK;
#ifdef KX_BOUNDS
  kx::bounds_untrack(ptr);
#endif
  KX_HIP_TRY(hipFree(ptr));
  return KX_OK;
}
extern "C" int kx_memset(void* ptr, int value, size_t bytes, void* stream) {
  if (!ptr && bytes) return KX_EINVAL;
  if (bytes == 0) return KX_OK;
  KX_HIP_TRY(hipMemsetAsync(ptr, value, bytes, (hipStream_t)stream));
  // Without a stream the fill has happened when the call returns.  The streams of this library
  // are created non-blocking, so nothing orders work on them behind a fill that is still queued on
  // the null stream: a kernel that writes the buffer next (rotary tables, synthetic weights, the
  // first K/V rows) could otherwise run ahead of the fill and have its output zeroed afterwards.
  if (!stream) KX_HIP_TRY(hipStreamSynchronize(nullptr));
  return KX_OK;
}

static int copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, void* stream) {
  if (bytes == 0) return KX_OK;
  if (!dst || !src) return KX_EINVAL;
  if (stream) {
    KX_HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, (hipStream_t)stream));
  } else {
    KX_HIP_TRY(hipMemcpy(dst, src, bytes, kind));
    // a device-to-device hipMemcpy may return before the copy has run; without a stream every
    // transfer of this library is complete on return (see kx_memset)
    if (kind == hipMemcpyDeviceToDevice) KX_HIP_TRY(hipStreamSynchronize(nullptr));
  }
  return KX_OK;
}
extern "C" int kx_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  return copy(dst, src, bytes, hipMemcpyHostToDevice, stream);
}
extern "C" int kx_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  return copy(dst, src, bytes, hipMemcpyDeviceToHost, stream);
}
extern "C" int kx_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  return copy(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
}
extern "C" int kx_host_alloc_pinned(void** ptr, size_t bytes) {
  if (!ptr) return KX_EINVAL;
  KX_HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
  return KX_OK;
}
extern "C" int kx_host_free_pinned(void* ptr) {
  if (!ptr) return KX_OK;
  KX_HIP_TRY(hipHostFree(ptr));
  return KX_OK;
}

extern "C" int kx_stream_create(void** stream) {
  if (!stream) return KX_EINVAL;
  hipStream_t s;
  KX_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return KX_OK;
}
extern "C" int kx_stream_destroy(void* stream) {
  if (!stream) return KX_OK;
  KX_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return KX_OK;
}
extern "C" int kx_stream_sync(void* stream) {
  KX_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return KX_OK;
}
extern "C" int kx_mem_info(size_t* free_bytes, size_t* total_bytes) {
  size_t f = 0, t = 0;
  KX_HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return KX_OK;
}

extern "C" int kx_device_sync(void) {
  KX_HIP_TRY(hipDeviceSynchronize());
  return KX_OK;
}

extern "C" int kx_event_create(void** event) {
  if (!event) return KX_EINVAL;
  hipEvent_t e;
  KX_HIP_TRY(hipEventCreate(&e));
  *event = e;
  return KX_OK;
}
extern "C" int kx_event_destroy(void* event) {
  if (!event) return KX_OK;
  KX_HIP_TRY(hipEventDestroy((hipEvent_t)event));
  return KX_OK;
}
extern "C" int kx_event_record(void* event, void* stream) {
  if (!event) return KX_EINVAL;
  KX_HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return KX_OK;
}
extern "C" int kx_event_sync(void* event) {
  if (!event) return KX_EINVAL;
  KX_HIP_TRY(hipEventSynchronize((hipEvent_t)event));
  return KX_OK;
}
extern "C" int kx_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!start || !stop || !ms) return KX_EINVAL;
  KX_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return KX_OK;
}
extern "C" int kx_stream_wait_event(void* stream, void* event) {
  if (!event) return KX_EINVAL;
  KX_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return KX_OK;
}

extern "C" int kx_graph_begin_capture(void* stream) {
  if (!stream) return KX_EINVAL;  // the legacy default stream cannot be captured
  KX_HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return KX_OK;
}
extern "C" int kx_graph_end_capture(void* stream, void** graph_exec) {
  if (!stream || !graph_exec) return KX_EINVAL;
  hipGraph_t g = nullptr;
  KX_HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return (int)e;
  *graph_exec = ge;
  return KX_OK;
}
extern "C" int kx_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) return KX_EINVAL;
  KX_HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return KX_OK;
}
extern "C" int kx_graph_destroy(void* graph_exec) {
  if (!graph_exec) return KX_OK;
  KX_HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return KX_OK;
}
