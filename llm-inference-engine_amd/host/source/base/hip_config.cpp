#include "base/hip_config.h"

#include "base/base.h"
#include "kx_hip.h"

namespace kernel {
HipConfig::~HipConfig() {
  if (compute_done) kx_event_destroy(compute_done);
  if (comm_done) kx_event_destroy(comm_done);
  if (stream) kx_stream_destroy(stream);
  if (comm_stream) kx_stream_destroy(comm_stream);
  if (attn_workspace) kx_free(attn_workspace);
}

int HipConfig::create_streams() {
  int rc;
  if (!stream && (rc = kx_stream_create(&stream))) return rc;
  if (!comm_stream && (rc = kx_stream_create(&comm_stream))) return rc;
  if (!compute_done && (rc = kx_event_create(&compute_done))) return rc;
  if (!comm_done && (rc = kx_event_create(&comm_done))) return rc;
  return 0;
}

int HipConfig::reserve_attn_workspace(int head_num, int kv_head_num, int head_size, int seq_len) {
  size_t need = kx_attn_workspace_bytes(head_num, head_size);
  // the generic attention kernel wants a [head_num][seq_len] score scratch
  // behind the ticket words that head the workspace
  const size_t score = 16384 + (size_t)head_num * (size_t)seq_len * sizeof(float);
  if (score > need) need = score;
  // and the flash-attention prefill wants fp16 copies of up to seq_len K/V rows of a layer
  const size_t fa = kx_flash_attention_workspace_bytes(seq_len - 1, kv_head_num, head_size);
  if (fa > need) need = fa;
  if (need <= attn_workspace_bytes) return 0;
  if (attn_workspace) kx_free(attn_workspace);
  attn_workspace = nullptr;
  attn_workspace_bytes = 0;
  int rc = kx_malloc(&attn_workspace, need);
  // the split kernels count arrivals in the head of the workspace and expect it zeroed once
  if (rc == 0) rc = kx_memset(attn_workspace, 0, need, nullptr);
  if (rc == 0) rc = kx_device_sync();
  if (rc == 0) attn_workspace_bytes = need;
  return rc;
}
}  // namespace kernel
