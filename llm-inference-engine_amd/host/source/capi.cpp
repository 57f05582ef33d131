// extern "C" shim over model::LlamaModel and a few layer-level probes (host/include/kx_host.h).
#include "kx_host.h"

#include <cmath>
#include <cstring>
#include <string>

#include "kx_hip.h"
#include "model/llama.h"
#include "op/kernels/kernels_interface.h"

namespace {
thread_local std::string g_err;
int fail(const base::Status& s) {
  g_err = s.get_err_msg();
  if (g_err.empty()) g_err = "status code " + std::to_string(int(s));
  return int(s) ? int(s) : -1;
}
int done(const base::Status& s) { return s ? 0 : fail(s); }
model::LlamaModel* M(void* p) { return static_cast<model::LlamaModel*>(p); }
constexpr base::DeviceType kHip = base::DeviceType::kDeviceHIP;

tensor::Tensor dev_from_host(base::DataType dt, std::vector<int32_t> dims, const void* host) {
  tensor::Tensor t(dt, std::move(dims), true, base::HIPDeviceAllocatorFactory::get_instance());
  if (host && t.byte_size()) kx_memcpy_h2d(t.ptr<void>(), host, t.byte_size(), nullptr);
  return t;
}
}  // namespace

extern "C" {
const char* kxm_last_error(void) { return g_err.c_str(); }

void* kxm_create(const kxm_config* c) {
  if (!c) return nullptr;
  model::TransformerConfig t;
  t.dim = c->dim; t.hidden_dim = c->hidden_dim; t.layer_num = c->layer_num;
  t.head_num = c->head_num; t.kv_head_num = c->kv_head_num; t.vocab_size = c->vocab_size;
  t.seq_len = c->seq_len; t.group_size = c->group_size; t.rope_variant = c->rope_variant;
  t.rope_theta = c->rope_theta; t.rms_eps = c->rms_eps;
  t.tp_size = c->tp_size < 1 ? 1 : c->tp_size; t.tp_rank = c->tp_rank;
  t.kv_f16 = c->kv_f16 != 0;
  t.qkv_bias = c->qkv_bias != 0;
  t.prefill_weight_cache = c->prefill_weight_cache;
  auto* m = new model::LlamaModel();
  base::Status s = m->init(t);
  if (!s) {
    fail(s);
    delete m;
    return nullptr;
  }
  return m;
}
void kxm_destroy(void* m) { delete M(m); }
int kxm_init_synthetic(void* m, uint64_t seed) { return done(M(m)->init_synthetic(seed)); }
int kxm_load_blob(void* m, const void* blob, size_t bytes) { return done(M(m)->load_blob(blob, bytes)); }
int kxm_set_qkv_bias(void* m, int32_t layer, const float* bq, const float* bk, const float* bv) {
  return done(M(m)->set_qkv_bias(layer, bq, bk, bv));
}
int kxm_read_weight(void* m, const char* name, int32_t layer, void* out, size_t bytes) {
  return done(M(m)->read_weight(name, layer, out, bytes));
}
int kxm_shard_plan(int32_t dim, int32_t hidden_dim, int32_t head_num, int32_t kv_head_num,
                   int32_t tp_size, int32_t tp_rank, int32_t* slices28, int32_t* carries_residual) {
  model::ShardPlan p;
  if (!slices28 || !model::make_shard_plan(dim, hidden_dim, head_num, kv_head_num, tp_size, tp_rank, &p))
    return -1;
  const model::ShardSlice* all[7] = {&p.wq, &p.wk, &p.wv, &p.wo, &p.w1, &p.w2, &p.w3};
  for (int i = 0; i < 7; ++i) {
    slices28[4 * i + 0] = all[i]->row0;
    slices28[4 * i + 1] = all[i]->rows;
    slices28[4 * i + 2] = all[i]->col0;
    slices28[4 * i + 3] = all[i]->cols;
  }
  if (carries_residual) *carries_residual = p.carries_residual ? 1 : 0;
  return 0;
}

int kxm_comm_unique_id(void* id128) {
  base::CommUniqueId id;
  base::Status s = base::RcclHandler::CreateUniqueId(&id);
  if (!s) return fail(s);
  std::memcpy(id128, id.data(), id.size());
  return 0;
}
int kxm_init_comm(void* m, const void* id128) {
  base::CommUniqueId id;
  if (id128) std::memcpy(id.data(), id128, id.size());
  return done(M(m)->init_comm(id128 ? &id : nullptr));
}
int kxm_peer_create(void* m, uint64_t max_floats, void* handle64) {
  return done(M(m)->peer_create((size_t)max_floats, handle64));
}
int kxm_peer_connect(void* m, const void* handles) { return done(M(m)->peer_connect(handles)); }
int kxm_peer_disable(void* m) {
  if (!m) return -1;
  M(m)->peer_disable();
  return 0;
}
int kxm_peer_timed_out(void* m, int32_t* flag) {
  bool f = false;
  const int rc = done(M(m)->peer_timed_out(&f));
  if (flag) *flag = f ? 1 : 0;
  return rc;
}
int kxm_set_prefill_weight_cache(void* m, int32_t policy) {
  if (!m || policy < -1 || policy > 1) return -1;
  M(m)->set_prefill_weight_cache(policy);
  return 0;
}

int kxm_weight_image_bytes(void* m, uint64_t* bytes) {
  if (!m || !bytes) return -1;
  *bytes = M(m)->weight_image_bytes();
  return 0;
}

int kxm_set_prefill_subblocks(void* m, int32_t n) {
  if (!m) return -1;
  M(m)->set_prefill_subblocks(n);
  return 0;
}
void* kxm_loopback_create(int32_t ranks) {
  if (ranks < 1) return nullptr;
  return new std::shared_ptr<base::LoopbackGroup>(std::make_shared<base::LoopbackGroup>(ranks));
}
void kxm_loopback_destroy(void* group) {
  delete static_cast<std::shared_ptr<base::LoopbackGroup>*>(group);
}
int kxm_join_loopback(void* m, void* group) {
  if (!m || !group) return -1;
  M(m)->join_loopback(*static_cast<std::shared_ptr<base::LoopbackGroup>*>(group));
  return 0;
}
int kxm_forward_layers(void* m, int32_t token, int32_t pos, float* logits) {
  return done(M(m)->forward_layers(token, pos, logits));
}
int kxm_set_state(void* m, int32_t token, int32_t pos) { return done(M(m)->set_state(token, pos)); }
int kxm_step_logits(void* m, int32_t token, int32_t pos, float* logits) {
  return done(M(m)->step_logits(token, pos, logits));
}
int kxm_decode(void* m, int32_t steps, int32_t use_graph) { return done(M(m)->decode(steps, use_graph != 0)); }
int kxm_prefill(void* m, const int32_t* tokens, int32_t n, int32_t pos0, float* logits,
                int32_t exact) {
  return done(M(m)->prefill(tokens, n, pos0, logits, exact != 0));
}
int kxm_read_history(void* m, int32_t* out, int32_t n) { return done(M(m)->read_history(out, n)); }
int kxm_read_state(void* m, int32_t* token, int32_t* pos) { return done(M(m)->read_state(token, pos)); }
int kxm_read_kv(void* m, int32_t layer, int32_t rows, float* k, float* v) {
  return done(M(m)->read_kv(layer, rows, k, v));
}
int kxm_sync(void* m) { return done(M(m)->sync()); }
int kxm_profile_gemv(void* m, int32_t steps, int64_t* launches, double* bytes, double* ms) {
  return done(M(m)->profile_gemv(steps, launches, bytes, ms));
}
int kxm_profile_allreduce(void* m, uint64_t count, int32_t iters, double* ms) {
  return done(M(m)->profile_allreduce((size_t)count, iters, ms));
}
double kxm_linear_bytes_per_token(void* m) { return M(m)->linear_weight_bytes_per_token(); }
void* kxm_stream(void* m) { return M(m)->hip_config()->stream; }

// ---------------- layer-level probes ----------------
int kxm_test_matmul_layer(const float* x, const void* w, const float* scales, int32_t group_size,
                          const float* bias, int32_t dim0, int32_t dim1, int32_t rows, float* y) {
  auto cfg = std::make_shared<kernel::CudaConfig>();
  if (int rc = cfg->create_streams()) return rc;
  const bool quant = group_size > 0;
  op::MatmulLayer layer(kHip, dim0, dim1, quant, bias != nullptr);
  layer.set_cuda_config(cfg);
  // Bind host memory exactly as the reference's loaders do (weights then scales in one blob for
  // quant layers, layer.cpp:203-216), then move the layer to the device.
  std::vector<uint8_t> blob;
  if (quant) {
    const size_t n = (size_t)dim0 * dim1;
    blob.resize(n + (n / group_size) * 4);
    std::memcpy(blob.data(), w, n);
    std::memcpy(blob.data() + n, scales, (n / group_size) * 4);
    layer.set_group_size(group_size);
    layer.set_weight(0, {dim0, dim1}, blob.data(), base::DeviceType::kDeviceCPU);
  } else {
    layer.set_weight(0, {dim0, dim1}, w, base::DeviceType::kDeviceCPU);
  }
  int32_t bdim = dim0;
  if (bias) layer.set_bias(0, bdim, bias, base::DeviceType::kDeviceCPU);
  layer.to_cuda();
  tensor::Tensor in = rows > 1 ? dev_from_host(base::DataType::kDataTypeFp32, {rows, dim1}, x)
                               : dev_from_host(base::DataType::kDataTypeFp32, {dim1}, x);
  tensor::Tensor out = rows > 1 ? dev_from_host(base::DataType::kDataTypeFp32, {rows, dim0}, nullptr)
                                : dev_from_host(base::DataType::kDataTypeFp32, {dim0}, nullptr);
  base::Status s = layer.forward(in, out);
  if (!s) return fail(s);
  kx_stream_sync(cfg->stream);
  out.to_cpu();
  std::memcpy(y, out.ptr<float>(), out.byte_size());
  return 0;
}

static int attn_common(bool flash, const float* q, const float* kc, const float* vc, int32_t layers,
                       int32_t seq_len, int32_t head_num, int32_t kv_mul, int32_t head_size,
                       int32_t layer_idx, int32_t pos, int32_t tq, float* out_host,
                       bool reserve_workspace = true) {
  auto cfg = std::make_shared<kernel::CudaConfig>();
  if (int rc = cfg->create_streams()) return rc;
  if (reserve_workspace)
    if (int rc = cfg->reserve_attn_workspace(head_num, head_num / kv_mul, head_size, seq_len)) return rc;
  const int32_t kv_dim = head_num / kv_mul * head_size;
  const base::DataType f32 = base::DataType::kDataTypeFp32;
  tensor::Tensor key = dev_from_host(f32, {layers, seq_len, kv_dim}, kc);
  tensor::Tensor val = dev_from_host(f32, {layers, seq_len, kv_dim}, vc);
  tensor::Tensor query = dev_from_host(f32, {tq, head_num * head_size}, q);
  tensor::Tensor out = dev_from_host(f32, {tq, head_num * head_size}, nullptr);
  base::Status s;
  if (flash) {
    op::FlashAttention layer(kHip, 0, kv_mul, kv_dim, seq_len, head_num, head_size);
    layer.set_cuda_config(cfg);
    layer.set_pos(pos);
    layer.set_layer_idx(layer_idx);
    s = layer.forward(query, key, val, out);
  } else {
    op::MultiHeadAttention layer(kHip, 0, kv_mul, kv_dim, seq_len, head_num, head_size);
    layer.set_cuda_config(cfg);
    layer.set_pos(pos);
    layer.set_layer_idx(layer_idx);
    tensor::Tensor score = dev_from_host(f32, {head_num, seq_len}, nullptr);
    // the reference allocates this tensor un-zeroed and leaves probabilities in it
    // (llama3.cpp:523-525): without a reserved workspace the layer must cope with any bytes
    if (!reserve_workspace) kx_memset(score.ptr<void>(), 0xFF, score.byte_size(), nullptr);
    s = layer.forward(query, score, key, val, out);
    if (s && !reserve_workspace) s = layer.forward(query, score, key, val, out);  // and with its own leftovers
  }
  if (!s) return fail(s);
  kx_stream_sync(cfg->stream);
  out.to_cpu();
  std::memcpy(out_host, out.ptr<float>(), out.byte_size());
  return 0;
}

int kxm_test_mha_layer(const float* q, const float* kc, const float* vc, int32_t layers,
                       int32_t seq_len, int32_t head_num, int32_t kv_mul, int32_t head_size,
                       int32_t layer_idx, int32_t pos, float* out) {
  return attn_common(false, q, kc, vc, layers, seq_len, head_num, kv_mul, head_size, layer_idx, pos,
                     1, out);
}
int kxm_test_mha_layer_no_workspace(const float* q, const float* kc, const float* vc, int32_t layers,
                                    int32_t seq_len, int32_t head_num, int32_t kv_mul,
                                    int32_t head_size, int32_t layer_idx, int32_t pos, float* out) {
  return attn_common(false, q, kc, vc, layers, seq_len, head_num, kv_mul, head_size, layer_idx, pos,
                     1, out, false);
}
int kxm_test_matmul_tp_without_comm(void) {
  // a row-parallel layer of a 2-rank group with no communicator behind it must fail, not return
  // its unreduced partial sum
  auto cfg = std::make_shared<kernel::CudaConfig>();
  if (int rc = cfg->create_streams()) return rc;
  op::MatmulLayer layer(kHip, 4, 8, false, false);
  layer.set_cuda_config(cfg);
  std::vector<float> w(32, 1.f), x(8, 1.f);
  layer.set_weight(0, {4, 8}, w.data(), base::DeviceType::kDeviceCPU);
  layer.to_cuda();
  layer.set_tp_config(2, 0, true);
  tensor::Tensor in = dev_from_host(base::DataType::kDataTypeFp32, {8}, x.data());
  tensor::Tensor out = dev_from_host(base::DataType::kDataTypeFp32, {4}, nullptr);
  base::Status s = layer.forward(in, out);
  kx_stream_sync(cfg->stream);
  if (s) return 0;
  fail(s);
  return 1;
}
int kxm_test_flash_layer(const float* q, const float* kc, const float* vc, int32_t layers,
                         int32_t seq_len, int32_t head_num, int32_t kv_mul, int32_t head_size,
                         int32_t layer_idx, int32_t pos, int32_t tq, float* out) {
  return attn_common(true, q, kc, vc, layers, seq_len, head_num, kv_mul, head_size, layer_idx, pos,
                     tq, out);
}

int kxm_test_kv_layout(int32_t layers, int32_t seq_len, int32_t kv_head_num, int32_t head_size,
                       int32_t layer_idx, int64_t* k_off, int64_t* v_off, int64_t* total_bytes) {
  base::KVCacheConfig c;
  c.max_seq_len = seq_len;
  c.layer_num = layers;
  c.kv_head_num = kv_head_num;
  c.kv_dim = kv_head_num * head_size;
  base::KVCacheManager mgr(base::DeviceType::kDeviceCPU, c);  // host arena: layout only
  base::Status s = mgr.allocate();
  if (!s) return fail(s);
  auto [k, v] = mgr.get_cache_ptr(layer_idx);
  const char* base_ptr = static_cast<const char*>(mgr.get_raw_buffer_k());
  *k_off = static_cast<const char*>(k) - base_ptr;
  *v_off = static_cast<const char*>(v) - base_ptr;
  *total_bytes = static_cast<int64_t>(mgr.total_bytes());
  return 0;
}

int kxm_test_registry(void) {
  using namespace kernel;
  return get_add_kernel(kHip) && get_emb_kernel(kHip) && get_matmul_kernel(kHip) &&
         get_matmul_kernel_quant8(kHip) && get_mha_kernel(kHip) && get_rmsnorm_kernel(kHip) &&
         get_rmsnorm_dim_kernel(kHip) && get_rope_kernel(kHip) && get_swiglu_kernel(kHip) &&
         get_flash_attention_kernel(kHip) && get_scale_kernel(kHip) && get_softmax_kernel(kHip) &&
         get_scale_sum_kernel(kHip);
}

int kxm_test_mha_from_registry_pieces(const float* q, const float* kc, const float* vc, int32_t layers,
                                      int32_t seq_len, int32_t head_num, int32_t kv_mul,
                                      int32_t head_size, int32_t layer_idx, int32_t pos, float* out_host) {
  // The body of the reference's CPU attention (cpu/mha_kernel.cpp:10-60) with kDeviceHIP: one
  // matmul-kernel call per key, the softmax kernel per head, the scale-sum kernel per head.
  using namespace kernel;
  auto cfg = std::make_shared<CudaConfig>();
  if (int rc = cfg->create_streams()) return rc;
  const base::DataType f32 = base::DataType::kDataTypeFp32;
  const int32_t kv_dim = head_num / kv_mul * head_size;
  tensor::Tensor key = dev_from_host(f32, {layers, seq_len, kv_dim}, kc);
  tensor::Tensor val = dev_from_host(f32, {layers, seq_len, kv_dim}, vc);
  tensor::Tensor query = dev_from_host(f32, {head_num * head_size}, q);
  tensor::Tensor score = dev_from_host(f32, {head_num, seq_len}, nullptr);
  tensor::Tensor out = dev_from_host(f32, {head_num * head_size}, nullptr);
  auto view = [&](std::vector<int32_t> dims, const float* ptr) {
    tensor::Tensor t(f32, std::move(dims), false, nullptr, const_cast<float*>(ptr));
    t.set_device_type(kHip);
    return t;
  };
  const size_t layer_offset = (size_t)layer_idx * seq_len * kv_dim;
  const float scale = 1.f / std::sqrt(static_cast<float>(head_size));
  for (int32_t h = 0; h < head_num; ++h) {
    float* score_head = score.ptr<float>() + (size_t)h * seq_len;
    tensor::Tensor query_mat = view({head_size}, query.ptr<float>() + (size_t)h * head_size);
    for (int32_t t = 0; t <= pos; ++t) {
      const size_t cache_offset = (size_t)t * kv_dim + (size_t)(h / kv_mul) * head_size;
      tensor::Tensor key_mat = view({1, head_size}, key.ptr<float>() + layer_offset + cache_offset);
      tensor::Tensor score_mat = view({1}, score_head + t);
      get_matmul_kernel(kHip)(query_mat, key_mat, score_mat, scale, cfg.get());
    }
    tensor::Tensor score_head_tensor = view({pos + 1}, score_head);
    get_softmax_kernel(kHip)(score_head_tensor, cfg->stream);
    float* out_head = out.ptr<float>() + (size_t)h * head_size;
    if (int rc = kx_memset(out_head, 0, sizeof(float) * head_size, cfg->stream)) return rc;
    tensor::Tensor output_tensor = view({head_size}, out_head);
    tensor::Tensor value_tensor =
        view({head_size}, val.ptr<float>() + layer_offset + (size_t)(h / kv_mul) * head_size);
    get_scale_sum_kernel(kHip)(value_tensor, score_head_tensor, output_tensor, pos, head_size, kv_dim,
                               cfg->stream);
  }
  // the scale kernel, on a copy of the finished rows: out2 = 2 * out behind the first head_num*hs floats
  kx_stream_sync(cfg->stream);
  out.to_cpu();
  std::memcpy(out_host, out.ptr<float>(), out.byte_size());
  tensor::Tensor twice = dev_from_host(f32, {head_num * head_size}, out_host);
  get_scale_kernel(kHip)(2.0f, twice, cfg->stream);
  kx_stream_sync(cfg->stream);
  twice.to_cpu();
  std::memcpy(out_host + (size_t)head_num * head_size, twice.ptr<float>(), twice.byte_size());
  return 0;
}
}  // extern "C"
