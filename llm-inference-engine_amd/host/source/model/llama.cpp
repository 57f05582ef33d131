#include "model/llama.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kx_hip.h"
#include "op/kernels/kernels_interface.h"

namespace model {
namespace {
constexpr base::DeviceType kHip = base::DeviceType::kDeviceHIP;

base::Status rc_status(int rc, const char* what) {
  if (rc == 0) return base::error::Success();
  return base::error::InternalError(std::string(what) + ": " + kx_error_string(rc) + " (rc " +
                                    std::to_string(rc) + ")");
}
#define KX_RET(expr)                                  \
  do {                                                \
    base::Status st_ = rc_status((expr), #expr);      \
    if (!st_) return st_;                             \
  } while (0)
#define ST_RET(expr)            \
  do {                          \
    base::Status st_ = (expr);  \
    if (!st_) return st_;       \
  } while (0)

template <typename T>
base::Status dev_alloc(T*& p, size_t count) {
  void* v = nullptr;
  int rc = kx_malloc(&v, count * sizeof(T));
  if (rc) return rc_status(rc, "kx_malloc");
  rc = kx_memset(v, 0, count * sizeof(T), nullptr);
  p = static_cast<T*>(v);
  return rc_status(rc, "kx_memset");
}
template <typename T>
void dev_free(T*& p) {
  if (p) kx_free(p);
  p = nullptr;
}

tensor::Tensor view(base::DataType dt, std::vector<int32_t> dims, void* ptr) {
  tensor::Tensor t(dt, std::move(dims), false, nullptr, ptr);
  t.set_device_type(kHip);
  return t;
}

double gemv_bytes(int64_t K, int64_t M, int gs) {
  return (double)K * M + 4.0 * (double)K * M / gs + 4.0 * (double)(M + K);
}
}  // namespace

base::Status TransformerConfig::finalize() {
  if (dim <= 0 || hidden_dim <= 0 || layer_num <= 0 || head_num <= 0 || kv_head_num <= 0 ||
      vocab_size <= 0 || seq_len <= 0)
    return base::error::InvalidArgument("model config has a non-positive dimension");
  if (dim % head_num || head_num % kv_head_num)
    return base::error::InvalidArgument("dim/head_num/kv_head_num do not divide");
  head_size = dim / head_num;                 // model.cpp:134
  kv_dim = (dim * kv_head_num) / head_num;    // model.cpp:132
  kv_mul = head_num / kv_head_num;            // model.cpp:133
  if (tp_size < 1 || tp_rank < 0 || tp_rank >= tp_size)
    return base::error::InvalidArgument("bad tensor-parallel rank/size");
  if (!make_shard_plan(dim, hidden_dim, head_num, kv_head_num, tp_size, tp_rank, &plan))
    return base::error::InvalidArgument("heads / kv heads / hidden_dim must divide by tp_size");
  head_local = plan.head_local;
  kv_head_local = plan.kv_head_local;
  q_dim_local = plan.q_dim_local;
  kv_dim_local = plan.kv_dim_local;
  hidden_local = plan.hidden_local;
  cls_split = tp_size > 1 && vocab_size % tp_size == 0 && vocab_size < (1 << 24);
  vocab_local = cls_split ? vocab_size / tp_size : vocab_size;
  vocab_row0 = cls_split ? tp_rank * vocab_local : 0;
  if (group_size > 0) {
    // quant groups run along the input dimension; row-parallel shards must cut on group borders
    if (dim % group_size || hidden_dim % group_size || q_dim_local % group_size ||
        hidden_local % group_size)
      return base::error::InvalidArgument("dims (and tensor-parallel shards) must be multiples of group_size");
  }
  return base::error::Success();
}

size_t LlamaModel::Linear::w_bytes() const { return (size_t)rows * cols; }

// ------------------------------------------------------------------------------------------
struct LlamaModel::RefLayers {
  std::vector<std::shared_ptr<op::MatmulLayer>> wq, wk, wv, wo, w1, w2, w3;
  std::shared_ptr<op::MatmulLayer> cls;
  std::vector<std::shared_ptr<op::RmsNormLayer>> rmsnorm;  // attn[L], ffn[L], final
  std::shared_ptr<op::RoPELayer> rope;
  std::shared_ptr<op::MultiHeadAttention> mha;
  std::shared_ptr<op::VecAddLayer> add;
  std::shared_ptr<op::SwiGLULayer> swiglu;
  std::shared_ptr<op::EmbeddingLayer> embedding;
  std::map<ModelBufferType, tensor::Tensor> buffers;
  tensor::Tensor pos_tensor, token_tensor, token_num;
};

LlamaModel::LlamaModel() = default;

LlamaModel::~LlamaModel() {
  if (graph_) kx_graph_destroy(graph_);
  if (peer_) kx_peer_destroy(peer_);
  for (auto& l : layers_) {
    free_linear(l.qkv);
    free_linear(l.wo);
    free_linear(l.w13);
    free_linear(l.w2);
    dev_free(l.attn_norm);
    dev_free(l.ffn_norm);
  }
  free_linear(cls_);
  dev_free(emb_); dev_free(final_norm_);
  dev_free(x_); dev_free(q_); dev_free(att_); dev_free(h_); dev_free(logits_);
  dev_free(rms_); dev_free(h3_); dev_free(tmp_dim_);
  dev_free(sin_); dev_free(cos_);
  dev_free(token_dev_); dev_free(pos_dev_); dev_free(history_dev_); dev_free(step_ws_);
  dev_free(pf_);
  dev_free(cand_);
}

base::Status LlamaModel::alloc_linear(Linear& l, int32_t rows, int32_t cols) {
  l.rows = rows;
  l.cols = cols;
  const size_t n = (size_t)rows * cols;
  void* w = nullptr;
  KX_RET(kx_malloc(&w, quant_ ? n : n * sizeof(float)));
  l.w = w;
  if (quant_) ST_RET(dev_alloc(l.s, n / cfg_.group_size));
  return base::error::Success();
}

void LlamaModel::free_linear(Linear& l) {
  if (l.w) kx_free(l.w);
  l.w = nullptr;
  if (l.w16) kx_free(l.w16);
  l.w16 = nullptr;
  l.w16_bytes = 0;
  l.w16_ready = false;
  dev_free(l.s);
  dev_free(l.bias);
}

base::Status LlamaModel::init(const TransformerConfig& cfg_in) {
  cfg_ = cfg_in;
  ST_RET(cfg_.finalize());
  quant_ = cfg_.group_size > 0;
  hip_ = std::make_shared<kernel::CudaConfig>();
  KX_RET(hip_->create_streams());  // llama3.cpp:124-128
  KX_RET(hip_->reserve_attn_workspace(cfg_.head_local, cfg_.kv_head_local, cfg_.head_size, cfg_.seq_len));

  const int32_t dim = cfg_.dim;
  layers_.resize(cfg_.layer_num);
  for (auto& l : layers_) {
    ST_RET(alloc_linear(l.qkv, cfg_.q_dim_local + 2 * cfg_.kv_dim_local, dim));
    if (cfg_.qkv_bias) ST_RET(dev_alloc(l.qkv.bias, cfg_.q_dim_local + 2 * cfg_.kv_dim_local));
    ST_RET(alloc_linear(l.wo, dim, cfg_.q_dim_local));
    ST_RET(alloc_linear(l.w13, 2 * cfg_.hidden_local, dim));
    ST_RET(alloc_linear(l.w2, dim, cfg_.hidden_local));
    ST_RET(dev_alloc(l.attn_norm, dim));
    ST_RET(dev_alloc(l.ffn_norm, dim));
  }
  ST_RET(alloc_linear(cls_, cfg_.vocab_local, dim));
  ST_RET(dev_alloc(cand_, 2 * (size_t)(1 + cfg_.tp_size)));
  ST_RET(dev_alloc(emb_, (size_t)cfg_.vocab_size * dim));
  ST_RET(dev_alloc(final_norm_, dim));

  ST_RET(dev_alloc(x_, dim));
  ST_RET(dev_alloc(q_, cfg_.q_dim_local));
  ST_RET(dev_alloc(att_, cfg_.q_dim_local));
  ST_RET(dev_alloc(h_, cfg_.hidden_local));
  ST_RET(dev_alloc(h3_, cfg_.hidden_local));
  ST_RET(dev_alloc(rms_, dim));
  ST_RET(dev_alloc(tmp_dim_, dim));
  ST_RET(dev_alloc(logits_, cfg_.vocab_size));
  ST_RET(dev_alloc(sin_, (size_t)cfg_.seq_len * cfg_.head_size));
  ST_RET(dev_alloc(cos_, (size_t)cfg_.seq_len * cfg_.head_size));
  ST_RET(dev_alloc(token_dev_, 1));
  ST_RET(dev_alloc(pos_dev_, 1));
  ST_RET(dev_alloc(history_dev_, cfg_.seq_len + 1));
  ST_RET(dev_alloc(step_ws_, kx_step_workspace_bytes() / sizeof(float)));
  KX_RET(kx_memset(step_ws_, 0, kx_step_workspace_bytes(), hip_->stream));  // the arrival counter starts at zero
  KX_RET(kx_sin_cos_cache_f32(cfg_.head_size, cfg_.seq_len, cfg_.rope_theta, sin_, cos_,
                              hip_->stream));

  // KV cache: K block then V block, each [layer][seq_len][kv_dim_local] (llama3.cpp:487-496)
  base::KVCacheConfig kvc;
  kvc.max_batch_size = 1;
  kvc.max_seq_len = cfg_.seq_len;
  kvc.layer_num = cfg_.layer_num;
  kvc.kv_head_num = cfg_.kv_head_local;
  kvc.kv_dim = cfg_.kv_dim_local;
  kvc.dtype = cfg_.kv_f16 ? base::DataType::kDataTypeFp16 : base::DataType::kDataTypeFp32;
  kv_ = std::make_unique<base::KVCacheManager>(kHip, kvc);
  ST_RET(kv_->allocate());
  KX_RET(kx_stream_sync(hip_->stream));
  build_layers();
  return base::error::Success();
}

// ---------------- weights ----------------
base::Status LlamaModel::fill_linear_rows(Linear& l, int32_t row_off, int32_t rows, int64_t g_ld,
                                          int64_t g_row0, int64_t g_col0, uint64_t seed) {
  void* st = hip_->stream;
  if (quant_) {
    int8_t* w = static_cast<int8_t*>(l.w) + (size_t)row_off * l.cols;
    KX_RET(kx_fill_hash_i8(w, rows, l.cols, g_ld, g_row0, g_col0, seed, st));
    const int gs = cfg_.group_size;
    float* s = l.s + (size_t)row_off * (l.cols / gs);
    KX_RET(kx_fill_hash_f32(s, rows, l.cols / gs, g_ld / gs, g_row0, g_col0 / gs, seed ^ 0x5ca1e5ull,
                            1e-4f, 3e-4f, st));
  } else {
    float* w = static_cast<float*>(l.w) + (size_t)row_off * l.cols;
    KX_RET(kx_fill_hash_f32(w, rows, l.cols, g_ld, g_row0, g_col0, seed, -0.035f, 0.035f, st));
  }
  return base::error::Success();
}

// fp16 images of every layer matrix for the long-prompt GEMMs, when the policy and the free
// memory allow; otherwise the GEMMs keep dequantising into scratch.
base::Status LlamaModel::reserve_weight_images() {
  if (weight_images_ != 0) return base::error::Success();
  weight_images_ = -1;
  if (cfg_.prefill_weight_cache == 0 || !quant_) return base::error::Success();
  size_t total = 0;
  for (auto& l : layers_)
    for (Linear* m : {&l.qkv, &l.wo, &l.w13, &l.w2})
      total += kx_gemm_w8_f16act_workspace_bytes(m->cols, m->rows) + 256;
  if (cfg_.prefill_weight_cache < 0) {
    size_t free_b = 0, total_b = 0;
    KX_RET(kx_mem_info(&free_b, &total_b));
    if (total > free_b / 3) return base::error::Success();
  }
  for (auto& l : layers_)
    for (Linear* m : {&l.qkv, &l.wo, &l.w13, &l.w2}) {
      const size_t need = kx_gemm_w8_f16act_workspace_bytes(m->cols, m->rows);
      void* p = nullptr;
      if (kx_malloc(&p, need) != 0) {  // out of memory after all: give the images back
        drop_weight_images(true);
        return base::error::Success();
      }
      m->w16 = p;
      m->w16_bytes = need;
      m->w16_ready = false;
    }
  weight_images_ = 1;
  return base::error::Success();
}

void LlamaModel::drop_weight_images(bool free_memory) {
  for (auto& l : layers_)
    for (Linear* m : {&l.qkv, &l.wo, &l.w13, &l.w2}) {
      m->w16_ready = false;
      if (free_memory && m->w16) {
        kx_free(m->w16);
        m->w16 = nullptr;
        m->w16_bytes = 0;
      }
    }
  if (free_memory) weight_images_ = -1;
}

void LlamaModel::set_prefill_weight_cache(int32_t policy) {
  cfg_.prefill_weight_cache = policy;
  if (policy == 0) drop_weight_images(true);
  weight_images_ = 0;  // decided again by the next long prompt
  if (policy != 0)
    for (auto& l : layers_)
      if (l.qkv.w16) weight_images_ = 1;
}

size_t LlamaModel::weight_image_bytes() const {
  size_t total = 0;
  for (const auto& l : layers_)
    for (const Linear* m : {&l.qkv, &l.wo, &l.w13, &l.w2}) total += m->w16_bytes;
  return total;
}

base::Status LlamaModel::init_synthetic(uint64_t seed) {
  drop_weight_images(false);
  const TransformerConfig& c = cfg_;
  const ShardPlan& sp = c.plan;
  void* st = hip_->stream;
  std::vector<float> ones(c.dim, 1.0f);
  for (int l = 0; l < c.layer_num; ++l) {
    LayerWeights& lw = layers_[l];
    auto sd = [&](int m) { return seed * 1000003ull + (uint64_t)l * 64 + m; };
    // every shard is a slice of one global tensor: element values depend on global indices only
    auto fill = [&](Linear& lin, int32_t dst_row, const ShardSlice& sl, int64_t g_ld, int m) {
      return fill_linear_rows(lin, dst_row, sl.rows, g_ld, sl.row0, sl.col0, sd(m));
    };
    ST_RET(fill(lw.qkv, 0, sp.wq, c.dim, 0));
    ST_RET(fill(lw.qkv, c.q_dim_local, sp.wk, c.dim, 1));
    ST_RET(fill(lw.qkv, c.q_dim_local + c.kv_dim_local, sp.wv, c.dim, 2));
    ST_RET(fill(lw.wo, 0, sp.wo, c.dim, 3));
    ST_RET(fill(lw.w13, 0, sp.w1, c.dim, 4));
    ST_RET(fill(lw.w13, c.hidden_local, sp.w3, c.dim, 6));
    ST_RET(fill(lw.w2, 0, sp.w2, c.hidden_dim, 5));
    if (c.qkv_bias) {
      // element r of a rank's bias slice is element (row0 + r) of the unsharded bias vector
      auto fill_bias = [&](int32_t dst, const ShardSlice& sl, int m) {
        return kx_fill_hash_f32(lw.qkv.bias + dst, 1, sl.rows, (int64_t)1 << 40, 0, sl.row0, sd(m),
                                -0.5f, 0.5f, st);
      };
      KX_RET(fill_bias(0, sp.wq, 8));
      KX_RET(fill_bias(c.q_dim_local, sp.wk, 9));
      KX_RET(fill_bias(c.q_dim_local + c.kv_dim_local, sp.wv, 10));
    }
    KX_RET(kx_memcpy_h2d(lw.attn_norm, ones.data(), c.dim * sizeof(float), nullptr));
    KX_RET(kx_memcpy_h2d(lw.ffn_norm, ones.data(), c.dim * sizeof(float), nullptr));
  }
  ST_RET(fill_linear_rows(cls_, 0, c.vocab_local, c.dim, c.vocab_row0, 0, seed * 1000003ull + 7777));
  KX_RET(kx_fill_hash_f32(emb_, c.vocab_size, c.dim, c.dim, 0, 0, seed * 1000003ull + 8888, -1.0f,
                          1.0f, st));
  KX_RET(kx_memcpy_h2d(final_norm_, ones.data(), c.dim * sizeof(float), nullptr));
  KX_RET(kx_stream_sync(st));
  return base::error::Success();
}

base::Status LlamaModel::upload_matrix(Linear& l, int32_t dst_row_off, const void* src_w,
                                       const float* src_s, int32_t g_rows, int32_t g_cols,
                                       int32_t row0, int32_t rows, int32_t col0, int32_t cols) {
  UNUSED(g_rows);
  CHECK_EQ(cols, l.cols);
  const size_t esz = quant_ ? 1 : sizeof(float);
  std::vector<uint8_t> stage((size_t)rows * cols * esz);
  const uint8_t* src = static_cast<const uint8_t*>(src_w);
  for (int32_t i = 0; i < rows; ++i)
    std::memcpy(stage.data() + (size_t)i * cols * esz,
                src + ((size_t)(row0 + i) * g_cols + col0) * esz, (size_t)cols * esz);
  KX_RET(kx_memcpy_h2d(static_cast<uint8_t*>(l.w) + (size_t)dst_row_off * cols * esz, stage.data(),
                       stage.size(), nullptr));
  if (quant_) {
    const int gs = cfg_.group_size;
    const int32_t gpr = g_cols / gs, lg = cols / gs;
    std::vector<float> ss((size_t)rows * lg);
    for (int32_t i = 0; i < rows; ++i)
      // scales may be unaligned inside the blob: byte copy
      std::memcpy(ss.data() + (size_t)i * lg,
                  reinterpret_cast<const uint8_t*>(src_s) +
                      ((size_t)(row0 + i) * gpr + col0 / gs) * sizeof(float),
                  (size_t)lg * sizeof(float));
    KX_RET(kx_memcpy_h2d(l.s + (size_t)dst_row_off * lg, ss.data(), ss.size() * sizeof(float),
                         nullptr));
  }
  return base::error::Success();
}

base::Status LlamaModel::load_blob(const void* blob, size_t bytes) {
  drop_weight_images(false);
  const TransformerConfig& c = cfg_;
  const size_t hdr = quant_ ? 32 : 28;  // model.cpp:110-114
  if (bytes < hdr) return base::error::ModelParseError("blob shorter than its header");
  ModelConfig mc;
  std::memcpy(&mc, blob, sizeof(mc));
  const bool shared = mc.vocab_size > 0;  // model.cpp:138-142
  if (mc.dim != c.dim || mc.hidden_dim != c.hidden_dim || mc.layer_num != c.layer_num ||
      mc.head_num != c.head_num || mc.kv_head_num != c.kv_head_num ||
      std::abs(mc.vocab_size) != c.vocab_size || mc.seq_len > c.seq_len)
    return base::error::ModelParseError("blob header does not match the model config");
  if (quant_) {
    int32_t gs = 0;
    std::memcpy(&gs, static_cast<const uint8_t*>(blob) + 28, 4);
    if (gs != c.group_size) return base::error::ModelParseError("group size mismatch");
    // A tied classifier has no int8 copy in the file (tools/export_llama.py:183-192); the
    // reference then reads the fp32 embedding bytes as int8 (llama3.cpp:273-286).  Here the int8
    // classifier is made at load, below, by the exporter's own rule from the fp32 table.
  }
  const size_t dim = c.dim, hid = c.hidden_dim, L = c.layer_num, V = c.vocab_size;
  const size_t kvd = c.kv_dim;
  const ShardPlan& sp = c.plan;
  const uint8_t* p = static_cast<const uint8_t*>(blob) + hdr;
  const uint8_t* end = static_cast<const uint8_t*>(blob) + bytes;
  const size_t esz = quant_ ? 1 : 4;
  auto msize = [&](size_t rows, size_t cols) {
    const size_t n = rows * cols;
    return n * esz + (quant_ ? (n / c.group_size) * 4 : 0);
  };
  // one matrix kind for all layers, then the next kind (llama3.cpp:210-271, 323-380)
  auto take = [&](size_t rows, size_t cols, size_t l, const uint8_t*& w, const float*& s) {
    const uint8_t* base = p + l * msize(rows, cols);
    w = base;
    s = reinterpret_cast<const float*>(base + rows * cols * esz);
  };
  auto upload_kind = [&](size_t rows, size_t cols, auto&& per_layer) -> base::Status {
    if (p + L * msize(rows, cols) > end) return base::error::ModelParseError("blob truncated");
    for (size_t l = 0; l < L; ++l) {
      const uint8_t* w;
      const float* s;
      take(rows, cols, l, w, s);
      ST_RET(per_layer(l, w, s));
    }
    p += L * msize(rows, cols);
    return base::error::Success();
  };
  // fp32 matrices each followed by their bias vector (tools/export_qwen2.py:102-110; reader
  // qwen2.cpp:306-333): the rank's rows of the bias go to qkv.bias at dst_off
  auto upload_kind_bias = [&](size_t rows, size_t cols, auto&& per_layer, int32_t dst_off,
                              const ShardSlice& sl) -> base::Status {
    const size_t block = (rows * cols + rows) * 4;
    if (p + L * block > end) return base::error::ModelParseError("blob truncated");
    for (size_t l = 0; l < L; ++l) {
      const uint8_t* w = p + l * block;
      ST_RET(per_layer(l, w, nullptr));
      KX_RET(kx_memcpy_h2d(layers_[l].qkv.bias + dst_off, w + (rows * cols + sl.row0) * 4,
                           (size_t)sl.rows * 4, nullptr));
    }
    p += L * block;
    return base::error::Success();
  };
  const float *emb = nullptr, *attn_norm = nullptr, *ffn_norm = nullptr, *final_norm = nullptr;
  const uint8_t* cls_w = nullptr;
  const float* cls_s = nullptr;
  const int32_t ql = c.q_dim_local, kl = c.kv_dim_local, hl = c.hidden_local;
  auto up = [&](Linear& lin, int32_t dst_row, const uint8_t* w, const float* s, size_t g_rows,
                size_t g_cols, const ShardSlice& sl) {
    return upload_matrix(lin, dst_row, w, s, (int32_t)g_rows, (int32_t)g_cols, sl.row0, sl.rows,
                         sl.col0, sl.cols);
  };
  auto up_q = [&](size_t l, const uint8_t* w, const float* s) {
    return up(layers_[l].qkv, 0, w, s, dim, dim, sp.wq);
  };
  auto up_k = [&](size_t l, const uint8_t* w, const float* s) {
    return up(layers_[l].qkv, ql, w, s, kvd, dim, sp.wk);
  };
  auto up_v = [&](size_t l, const uint8_t* w, const float* s) {
    return up(layers_[l].qkv, ql + kl, w, s, kvd, dim, sp.wv);
  };
  auto up_o = [&](size_t l, const uint8_t* w, const float* s) {
    return up(layers_[l].wo, 0, w, s, dim, dim, sp.wo);
  };
  auto up_1 = [&](size_t l, const uint8_t* w, const float* s) {
    return up(layers_[l].w13, 0, w, s, hid, dim, sp.w1);
  };
  auto up_2 = [&](size_t l, const uint8_t* w, const float* s) {
    return up(layers_[l].w2, 0, w, s, dim, hid, sp.w2);
  };
  auto up_3 = [&](size_t l, const uint8_t* w, const float* s) {
    return up(layers_[l].w13, hl, w, s, hid, dim, sp.w3);
  };
  if (!quant_) {
    // tools/export_llama.py:96-127
    emb = reinterpret_cast<const float*>(p);        p += V * dim * 4;
    attn_norm = reinterpret_cast<const float*>(p);  p += L * dim * 4;
    if (c.qkv_bias) {
      ST_RET(upload_kind_bias(dim, dim, up_q, 0, sp.wq));
      ST_RET(upload_kind_bias(kvd, dim, up_k, ql, sp.wk));
      ST_RET(upload_kind_bias(kvd, dim, up_v, ql + kl, sp.wv));
    } else {
      ST_RET(upload_kind(dim, dim, up_q));
      ST_RET(upload_kind(kvd, dim, up_k));
      ST_RET(upload_kind(kvd, dim, up_v));
    }
    ST_RET(upload_kind(dim, dim, up_o));
    ffn_norm = reinterpret_cast<const float*>(p);   p += L * dim * 4;
    ST_RET(upload_kind(hid, dim, up_1));
    ST_RET(upload_kind(dim, hid, up_2));
    ST_RET(upload_kind(hid, dim, up_3));
    final_norm = reinterpret_cast<const float*>(p); p += dim * 4;
    p += (size_t)mc.seq_len * c.head_size * 4;  // freqs_cos + freqs_sin, skipped (llama3.cpp:384-385)
    cls_w = shared ? reinterpret_cast<const uint8_t*>(emb) : p;
    if (!shared) p += V * dim * 4;
  } else {
    // tools/export_llama.py:153-203
    ST_RET(upload_kind(dim, dim, up_q));
    ST_RET(upload_kind(kvd, dim, up_k));
    ST_RET(upload_kind(kvd, dim, up_v));
    ST_RET(upload_kind(dim, dim, up_o));
    ST_RET(upload_kind(hid, dim, up_1));
    ST_RET(upload_kind(dim, hid, up_2));
    ST_RET(upload_kind(hid, dim, up_3));
    if (!shared) {
      cls_w = p;
      cls_s = reinterpret_cast<const float*>(p + V * dim);
      p += msize(V, dim);
    }
    emb = reinterpret_cast<const float*>(p);        p += V * dim * 4;
    attn_norm = reinterpret_cast<const float*>(p);  p += L * dim * 4;
    ffn_norm = reinterpret_cast<const float*>(p);   p += L * dim * 4;
    final_norm = reinterpret_cast<const float*>(p); p += dim * 4;
  }
  if (p > end) return base::error::ModelParseError("blob truncated");
  KX_RET(kx_memcpy_h2d(emb_, emb, V * dim * 4, nullptr));
  if (quant_ && shared) {
    // quantize_q80 (tools/export_llama.py:49-73) of the rank's rows of the embedding table, on the
    // device: what the exporter writes for a classifier of its own.  Groups never straddle rows
    // (dim is a multiple of the group size).
    KX_RET(kx_quantize_q80_f32(emb_ + (size_t)c.vocab_row0 * dim, (size_t)c.vocab_local * dim, c.group_size,
                               static_cast<int8_t*>(cls_.w), cls_.s, hip_->stream));
    KX_RET(kx_stream_sync(hip_->stream));
  } else {
    ST_RET(upload_matrix(cls_, 0, cls_w, cls_s, V, dim, c.vocab_row0, c.vocab_local, 0, dim));
  }
  for (size_t l = 0; l < L; ++l) {
    KX_RET(kx_memcpy_h2d(layers_[l].attn_norm, reinterpret_cast<const uint8_t*>(attn_norm) + l * dim * 4,
                         dim * 4, nullptr));
    KX_RET(kx_memcpy_h2d(layers_[l].ffn_norm, reinterpret_cast<const uint8_t*>(ffn_norm) + l * dim * 4,
                         dim * 4, nullptr));
  }
  KX_RET(kx_memcpy_h2d(final_norm_, final_norm, dim * 4, nullptr));
  return base::error::Success();
}

base::Status LlamaModel::set_qkv_bias(int32_t layer, const float* bq, const float* bk,
                                      const float* bv) {
  const TransformerConfig& c = cfg_;
  if (!c.qkv_bias) return base::error::InvalidArgument("the model was created without qkv_bias");
  if (layer < 0 || layer >= c.layer_num || !bq || !bk || !bv)
    return base::error::InvalidArgument("bad layer or null bias");
  float* dst = layers_[layer].qkv.bias;
  KX_RET(kx_stream_sync(hip_->stream));
  KX_RET(kx_memcpy_h2d(dst, bq + c.plan.wq.row0, (size_t)c.q_dim_local * 4, nullptr));
  KX_RET(kx_memcpy_h2d(dst + c.q_dim_local, bk + c.plan.wk.row0, (size_t)c.kv_dim_local * 4, nullptr));
  KX_RET(kx_memcpy_h2d(dst + c.q_dim_local + c.kv_dim_local, bv + c.plan.wv.row0,
                       (size_t)c.kv_dim_local * 4, nullptr));
  return base::error::Success();
}

base::Status LlamaModel::read_weight(const std::string& name, int32_t layer, void* host_out,
                                     size_t bytes) const {
  const TransformerConfig& c = cfg_;
  const size_t esz = quant_ ? 1 : 4;
  const void* src = nullptr;
  size_t n = 0;
  auto sub = [&](const Linear& l, int32_t row_off, int32_t rows, bool scales) {
    if (scales) {
      if (!quant_) return;
      src = l.s + (size_t)row_off * (l.cols / c.group_size);
      n = (size_t)rows * (l.cols / c.group_size) * 4;
    } else {
      src = static_cast<const uint8_t*>(l.w) + (size_t)row_off * l.cols * esz;
      n = (size_t)rows * l.cols * esz;
    }
  };
  std::string base_name = name;
  bool scales = false;
  if (name.size() > 2 && name.substr(name.size() - 2) == "_s") {
    scales = true;
    base_name = name.substr(0, name.size() - 2);
  }
  if (base_name == "emb") { src = emb_; n = (size_t)c.vocab_size * c.dim * 4; }
  else if (base_name == "final_norm") { src = final_norm_; n = (size_t)c.dim * 4; }
  else if (base_name == "cls") sub(cls_, 0, cls_.rows, scales);
  else {
    if (layer < 0 || layer >= c.layer_num) return base::error::InvalidArgument("layer out of range");
    const LayerWeights& lw = layers_[layer];
    if (base_name == "attn_norm") { src = lw.attn_norm; n = (size_t)c.dim * 4; }
    else if (base_name == "ffn_norm") { src = lw.ffn_norm; n = (size_t)c.dim * 4; }
    else if (base_name == "bq" && lw.qkv.bias) { src = lw.qkv.bias; n = (size_t)c.q_dim_local * 4; }
    else if (base_name == "bk" && lw.qkv.bias) { src = lw.qkv.bias + c.q_dim_local; n = (size_t)c.kv_dim_local * 4; }
    else if (base_name == "bv" && lw.qkv.bias) {
      src = lw.qkv.bias + c.q_dim_local + c.kv_dim_local;
      n = (size_t)c.kv_dim_local * 4;
    }
    else if (base_name == "wq") sub(lw.qkv, 0, c.q_dim_local, scales);
    else if (base_name == "wk") sub(lw.qkv, c.q_dim_local, c.kv_dim_local, scales);
    else if (base_name == "wv") sub(lw.qkv, c.q_dim_local + c.kv_dim_local, c.kv_dim_local, scales);
    else if (base_name == "wo") sub(lw.wo, 0, lw.wo.rows, scales);
    else if (base_name == "w1") sub(lw.w13, 0, c.hidden_local, scales);
    else if (base_name == "w3") sub(lw.w13, c.hidden_local, c.hidden_local, scales);
    else if (base_name == "w2") sub(lw.w2, 0, lw.w2.rows, scales);
  }
  if (!src) return base::error::InvalidArgument("unknown weight name " + name);
  if (bytes != n)
    return base::error::InvalidArgument("weight " + name + " is " + std::to_string(n) + " bytes");
  KX_RET(kx_device_sync());
  KX_RET(kx_memcpy_d2h(host_out, src, n, nullptr));
  return base::error::Success();
}

base::Status LlamaModel::init_comm(const base::CommUniqueId* id) {
  ST_RET(base::RcclHandler::get_instance()->Init(cfg_.tp_rank, cfg_.tp_size, id));
  // With a communicator of one rank the all-reduces still run (as identities): the collective,
  // its side stream and its events are then exercised, graph capture included, on a single GPU.
  comm_ready_ = cfg_.tp_size > 1 || id != nullptr;
  return base::error::Success();
}

base::Status LlamaModel::peer_create(size_t max_floats, void* handle64) {
  if (peer_) return base::error::InvalidArgument("peer mailboxes already created");
  if (!handle64 || max_floats == 0) return base::error::InvalidArgument("bad peer mailbox request");
  KX_RET(kx_peer_create(&peer_, cfg_.tp_size, cfg_.tp_rank, max_floats));
  peer_cap_ = max_floats;
  KX_RET(kx_peer_handle(peer_, handle64));
  return base::error::Success();
}

base::Status LlamaModel::peer_connect(const void* handles) {
  if (!peer_ || peer_connected_ || peer_mapped_)
    return base::error::InvalidArgument("peer_create() first, connect once");
  KX_RET(kx_peer_connect(peer_, handles));
  peer_connected_ = peer_mapped_ = true;
  // a step captured earlier holds the other collective
  if (graph_) {
    kx_graph_destroy(graph_);
    graph_ = nullptr;
  }
  return base::error::Success();
}

void LlamaModel::peer_disable() {
  if (!peer_connected_) return;
  peer_connected_ = false;
  if (graph_) {
    kx_graph_destroy(graph_);
    graph_ = nullptr;
  }
}

base::Status LlamaModel::peer_timed_out(bool* flag) const {
  if (!peer_ || !flag) return base::error::InvalidArgument("no peer mailboxes");
  int32_t f = 0;
  KX_RET(kx_peer_timed_out(peer_, &f));
  *flag = f != 0;
  return base::error::Success();
}

double LlamaModel::linear_weight_bytes_per_token() const {
  if (!quant_) return 0.0;
  double b = 0;
  const int gs = cfg_.group_size;
  for (const auto& l : layers_) {
    b += gemv_bytes(l.qkv.rows, l.qkv.cols, gs) + gemv_bytes(l.wo.rows, l.wo.cols, gs) +
         gemv_bytes(l.w13.rows, l.w13.cols, gs) + gemv_bytes(l.w2.rows, l.w2.cols, gs);
  }
  return b + gemv_bytes(cls_.rows, cls_.cols, gs);
}

// ---------------- reference-structured path ----------------
void LlamaModel::build_layers() {
  const TransformerConfig& c = cfg_;
  ref_ = std::make_unique<RefLayers>();
  RefLayers& R = *ref_;
  const base::DataType wdt = quant_ ? base::DataType::kDataTypeInt8 : base::DataType::kDataTypeFp32;
  const size_t esz = quant_ ? 1 : 4;
  auto make = [&](const Linear& l, int32_t row_off, int32_t rows, bool all_reduce) {
    auto m = std::make_shared<op::MatmulLayer>(kHip, rows, l.cols, quant_, l.bias != nullptr);
    m->set_cuda_config(hip_);
    if (l.bias) {
      int32_t n = rows;
      STATUS_CHECK(m->set_bias(0, n, l.bias + row_off, kHip));
    }
    // row-parallel sums go through the model's own all-reduce (RCCL on the side stream, or the
    // loopback rendezvous of the test rig)
    if (all_reduce) m->set_reducer([this](float* data, size_t count) { return this->all_reduce(data, count); });
    uint8_t* w = static_cast<uint8_t*>(l.w) + (size_t)row_off * l.cols * esz;
    if (quant_) {
      m->set_group_size(c.group_size);
      // weights and scales live in separate device arrays here, so bind them one by one
      // (set_weight(idx, Tensor) + set_scales) instead of the one-blob form
      STATUS_CHECK(m->set_weight(0, view(wdt, {rows, l.cols}, w)));
      const int32_t lg = l.cols / c.group_size;
      m->set_scales(view(base::DataType::kDataTypeFp32, {rows * lg}, l.s + (size_t)row_off * lg));
    } else {
      STATUS_CHECK(m->set_weight(0, view(wdt, {rows, l.cols}, w)));
    }
    m->set_tp_config(c.tp_size, c.tp_rank, all_reduce);
    return m;
  };
  for (int l = 0; l < c.layer_num; ++l) {
    const LayerWeights& lw = layers_[l];
    R.wq.push_back(make(lw.qkv, 0, c.q_dim_local, false));
    R.wk.push_back(make(lw.qkv, c.q_dim_local, c.kv_dim_local, false));
    R.wv.push_back(make(lw.qkv, c.q_dim_local + c.kv_dim_local, c.kv_dim_local, false));
    R.wo.push_back(make(lw.wo, 0, c.dim, true));
    R.w1.push_back(make(lw.w13, 0, c.hidden_local, false));
    R.w3.push_back(make(lw.w13, c.hidden_local, c.hidden_local, false));
    R.w2.push_back(make(lw.w2, 0, c.dim, true));
  }
  R.cls = make(cls_, 0, c.vocab_local, false);
  auto norm = [&](float* w) {
    auto n = std::make_shared<op::RmsNormLayer>(kHip, c.dim);
    n->set_cuda_config(hip_);
    n->set_eps(c.rms_eps);
    STATUS_CHECK(n->set_weight(0, view(base::DataType::kDataTypeFp32, {c.dim}, w)));
    return n;
  };
  // order as llama3.cpp:296-304: attention norms, ffn norms, final norm
  for (int l = 0; l < c.layer_num; ++l) R.rmsnorm.push_back(norm(layers_[l].attn_norm));
  for (int l = 0; l < c.layer_num; ++l) R.rmsnorm.push_back(norm(layers_[l].ffn_norm));
  R.rmsnorm.push_back(norm(final_norm_));
  R.rope = std::make_shared<op::RoPELayer>(kHip, c.q_dim_local, c.kv_dim_local, c.head_size);
  R.rope->set_rope_variant(c.rope_variant);
  R.mha = std::make_shared<op::MultiHeadAttention>(kHip, 0, c.kv_mul, c.kv_dim_local, c.seq_len,
                                                   c.head_local, c.head_size);
  R.add = std::make_shared<op::VecAddLayer>(kHip);
  R.swiglu = std::make_shared<op::SwiGLULayer>(kHip, c.hidden_local);
  R.embedding = std::make_shared<op::EmbeddingLayer>(kHip, c.dim, c.seq_len, c.vocab_size);
  STATUS_CHECK(R.embedding->set_weight(
      0, view(base::DataType::kDataTypeFp32, {c.vocab_size, c.dim}, emb_)));
  for (op::Layer* l : std::initializer_list<op::Layer*>{R.rope.get(), R.mha.get(), R.add.get(),
                                                        R.swiglu.get(), R.embedding.get()})
    l->set_cuda_config(hip_);

  // activation buffers, named as the reference's buffer map (llama3.cpp:460-536)
  using B = ModelBufferType;
  auto f32 = [&](std::vector<int32_t> dims, float* p) {
    return view(base::DataType::kDataTypeFp32, std::move(dims), p);
  };
  R.buffers[B::kInputEmbeddings] = f32({1, c.dim}, x_);
  R.buffers[B::kOutputRMSNorm] = f32({c.dim}, rms_);
  R.buffers[B::kFFNRMSNorm] = f32({c.dim}, rms_);
  R.buffers[B::kW2Output] = f32({c.dim}, tmp_dim_);
  R.buffers[B::kAttnOutput] = f32({c.dim}, tmp_dim_);
  R.buffers[B::kQuery] = f32({c.q_dim_local}, q_);
  R.buffers[B::kOutputMHA] = f32({c.q_dim_local}, att_);
  R.buffers[B::kW1Output] = f32({c.hidden_local}, h_);
  R.buffers[B::kW3Output] = f32({c.hidden_local}, h3_);
  R.buffers[B::kSinCache] = f32({c.head_size * c.seq_len}, sin_);
  R.buffers[B::kCosCache] = f32({c.head_size * c.seq_len}, cos_);
  R.buffers[B::kForwardOutput] = f32({c.vocab_local}, logits_ + c.vocab_row0);
  R.buffers[B::kKeyCache] = f32({c.layer_num, c.seq_len, c.kv_dim_local},
                                static_cast<float*>(kv_->get_raw_buffer_k()));
  R.buffers[B::kValueCache] = f32({c.layer_num, c.seq_len, c.kv_dim_local},
                                  static_cast<float*>(kv_->get_raw_buffer_v()));
  R.buffers[B::kScoreStorage] =
      view(base::DataType::kDataTypeFp32, {c.head_local, c.seq_len}, hip_->attn_workspace);
  auto cpu = base::CPUDeviceAllocatorFactory::get_instance();
  R.pos_tensor = tensor::Tensor(base::DataType::kDataTypeInt32, 1, true, cpu);
  R.token_tensor = tensor::Tensor(base::DataType::kDataTypeInt32, 1, true, cpu);
  R.token_num = tensor::Tensor(base::DataType::kDataTypeInt32, 1, true, cpu);
}

base::Status LlamaModel::forward_layers(int32_t token, int32_t pos, float* logits_host) {
  using B = ModelBufferType;
  RefLayers& R = *ref_;
  const TransformerConfig& c = cfg_;
  if (pos < 0 || pos >= c.seq_len) return base::error::InvalidArgument("pos outside seq_len");
  if (c.kv_f16) return base::error::FunctionNotImplement("the layer-by-layer loop reads an fp32 KV cache");
  auto& buf = R.buffers;
  R.pos_tensor.index<int32_t>(0) = pos;
  R.token_tensor.index<int32_t>(0) = token;
  tensor::Tensor x = view(base::DataType::kDataTypeFp32, {c.dim}, x_);
  x_valid_ = false;  // this loop uses x_ for its own token
  // embedding (llama3.cpp:615-635)
  ST_RET(R.embedding->forward(R.token_tensor, R.token_num, buf[B::kInputEmbeddings]));
  for (int32_t l = 0; l < c.layer_num; ++l) {
    // attention_rms (llama3.cpp:637-646)
    ST_RET(R.rmsnorm[l]->forward(x, buf[B::kOutputRMSNorm]));
    // attention_qkv (:648-677): k and v are written straight into their cache row
    auto [k_ptr, v_ptr] = kv_->get_cache_ptr(l);
    tensor::Tensor key = view(base::DataType::kDataTypeFp32, {c.kv_dim_local},
                              static_cast<float*>(k_ptr) + (size_t)pos * c.kv_dim_local);
    tensor::Tensor val = view(base::DataType::kDataTypeFp32, {c.kv_dim_local},
                              static_cast<float*>(v_ptr) + (size_t)pos * c.kv_dim_local);
    ST_RET(R.wq[l]->forward(buf[B::kOutputRMSNorm], buf[B::kQuery]));
    ST_RET(R.wk[l]->forward(buf[B::kOutputRMSNorm], key));
    ST_RET(R.wv[l]->forward(buf[B::kOutputRMSNorm], val));
    ST_RET(R.rope->forward(buf[B::kQuery], key, R.pos_tensor, buf[B::kSinCache], buf[B::kCosCache],
                           tensor::Tensor{}));
    // attention_mha (:716-743)
    R.mha->set_pos(pos);
    R.mha->set_layer_idx(l);
    ST_RET(R.mha->forward(buf[B::kQuery], buf[B::kScoreStorage], buf[B::kKeyCache],
                          buf[B::kValueCache], buf[B::kOutputMHA]));
    ST_RET(R.wo[l]->forward(buf[B::kOutputMHA], buf[B::kAttnOutput]));
    // feed_forward (:746-788)
    ST_RET(R.add->forward(x, buf[B::kAttnOutput], x));
    ST_RET(R.rmsnorm[l + c.layer_num]->forward(x, buf[B::kFFNRMSNorm]));
    ST_RET(R.w1[l]->forward(buf[B::kFFNRMSNorm], buf[B::kW1Output]));
    ST_RET(R.w3[l]->forward(buf[B::kFFNRMSNorm], buf[B::kW3Output]));
    ST_RET(R.swiglu->forward(buf[B::kW1Output], buf[B::kW3Output], buf[B::kW1Output]));
    ST_RET(R.w2[l]->forward(buf[B::kW1Output], buf[B::kW2Output]));
    ST_RET(R.add->forward(x, buf[B::kW2Output], x));
  }
  // cls_logits (:790-799)
  ST_RET(R.rmsnorm[2 * c.layer_num]->forward(x, x));
  ST_RET(R.cls->forward(x, buf[B::kForwardOutput]));
  ST_RET(gather_logits());
  if (logits_host) {
    KX_RET(kx_stream_sync(hip_->stream));
    KX_RET(kx_memcpy_d2h(logits_host, logits_, (size_t)c.vocab_size * 4, nullptr));
  }
  return base::error::Success();
}

// ---------------- fused path ----------------
base::Status LlamaModel::gemv(const Linear& l, int mode, const float* x, const float* norm_w,
                              float* y0, float* y1, float* y2, int32_t k0, int32_t k1,
                              const float* residual, void* stream) {
  kx_gemv_w8_desc d{};
  d.mode = mode;
  d.group_size = cfg_.group_size;
  d.M = l.cols;
  d.K = l.rows;
  d.x = x;
  d.w = static_cast<const int8_t*>(l.w);
  d.scales = l.s;
  d.norm_w = norm_w;
  d.norm_eps = cfg_.rms_eps;
  d.y0 = y0; d.y1 = y1; d.y2 = y2;
  d.k0 = k0; d.k1 = k1;
  d.bias = l.bias;
  d.residual = residual;
  if (mode == KX_GEMV_QKV_ROPE) {
    d.sin_cache = sin_;
    d.cos_cache = cos_;
    d.pos_dev = pos_dev_;
    d.rope_variant = cfg_.rope_variant;
    d.head_size = cfg_.head_size;
    d.kv_pos_stride = cfg_.kv_dim_local;
    d.kv_f16 = cfg_.kv_f16 ? 1 : 0;
  }
  if (prof_on_) {
    ++prof_launches_;
    prof_bytes_sum_ += gemv_bytes(l.rows, l.cols, cfg_.group_size);
  }
  KX_RET(kx_gemv_w8_fused_f32(&d, stream));
  return base::error::Success();
}

base::Status LlamaModel::all_reduce(float* buf, size_t count) {
  ST_RET(all_reduce_begin(buf, count));
  return all_reduce_end();
}

// The two halves of the side-stream all-reduce (matmul.cpp:93-116, events reused): begin() makes
// the comm stream wait for everything the compute stream has been given so far and queues the
// collective there; the compute stream is NOT held, so what it is given next runs beside the
// collective.  end() makes the compute stream wait for every collective queued since.
base::Status LlamaModel::all_reduce_begin(float* buf, size_t count) {
  if (loop_) return loop_->all_reduce(cfg_.tp_rank, buf, count, hip_->stream);  // test rig: in line
  if (peer_connected_ && count <= peer_cap_) {  // one launch on the compute stream; nothing pending
    KX_RET(kx_peer_allreduce_sum_f32(peer_, buf, count, hip_->stream));
    return base::error::Success();
  }
  if (cfg_.tp_size <= 1 && !comm_ready_) return base::error::Success();
  if (!comm_ready_) return base::error::InternalError("tensor-parallel model without init_comm()");
  auto comm = base::RcclHandler::get_instance();
  KX_RET(kx_event_record(hip_->compute_done, hip_->stream));
  KX_RET(kx_stream_wait_event(hip_->comm_stream, hip_->compute_done));
  ST_RET(comm->AllReduce(buf, buf, count, base::DataType::kDataTypeFp32, hip_->comm_stream));
  comm_pending_ = true;
  return base::error::Success();
}

base::Status LlamaModel::all_reduce_end() {
  if (!comm_pending_) return base::error::Success();
  comm_pending_ = false;
  KX_RET(kx_event_record(hip_->comm_done, hip_->comm_stream));
  KX_RET(kx_stream_wait_event(hip_->stream, hip_->comm_done));
  return base::error::Success();
}

base::Status LlamaModel::launch_step(void* st, bool sample) {
  const TransformerConfig& c = cfg_;
  if (!quant_) return base::error::FunctionNotImplement("the fused path needs int8 weights");
  const bool all = !prof_on_;  // profiling keeps only the GEMV launches
  // x_ holds the embedding of history[pos]: left there by the previous step's last launch, or
  // gathered here after the host changed the state (never inside a captured step: decode()
  // makes it valid before it captures or replays)
  if (all && !x_valid_) KX_RET(kx_embedding_dpos_f32(history_dev_, pos_dev_, emb_, c.dim, c.vocab_size, x_, st));
  x_valid_ = false;  // the layers overwrite it; classify_and_pick() makes it valid again
  const int32_t ql = c.q_dim_local, kl = c.kv_dim_local;
  // the cache is zero-filled when it is made and rows only ever receive finite numbers: rows above
  // the position may be read (with zero weight)
  const int32_t attn_fixed = attn_bound_ >= 0 && attn_bound_ < c.seq_len;
  const int32_t attn_bound = attn_fixed ? attn_bound_ : c.seq_len - 1;
  for (int32_t l = 0; l < c.layer_num; ++l) {
    const LayerWeights& lw = layers_[l];
    auto [k_ptr, v_ptr] = kv_->get_cache_ptr(l);
    ST_RET(gemv(lw.qkv, KX_GEMV_QKV_ROPE, x_, lw.attn_norm, q_, static_cast<float*>(k_ptr),
                static_cast<float*>(v_ptr), ql, ql + kl, nullptr, st));
    if (all) {
      // while the attention launch's last workgroups merge the splits, the finished ones read the
      // head of Wo's row blocks and scales into the XCD caches where the Wo GEMV's workgroups
      // will want them
      const int32_t rpb = kx_gemv_w8_block_rows(lw.wo.cols, lw.wo.rows, c.group_size);
      const size_t wo_bytes = (size_t)lw.wo.rows * lw.wo.cols;
      const void* n0 = rpb > 0 ? lw.wo.w : nullptr;
      const void* n1 = rpb > 0 ? lw.wo.s : nullptr;
      const size_t g0 = (size_t)rpb * lw.wo.cols, b1 = wo_bytes / c.group_size * sizeof(float),
                   g1 = g0 / c.group_size * sizeof(float);
      if (c.kv_f16)
        KX_RET(kx_mha_decode_kv16_ex(q_, kv_->get_raw_buffer_k(), kv_->get_raw_buffer_v(), att_,
                                        attn_bound, pos_dev_, attn_fixed, c.head_local, c.kv_mul, c.head_size, kl,
                                        c.seq_len, l, hip_->attn_workspace, hip_->attn_workspace_bytes,
                                        n0, wo_bytes, g0, n1, b1, g1, st));
      else
        KX_RET(kx_mha_decode_dpos_ex_f32(q_, static_cast<float*>(kv_->get_raw_buffer_k()),
                                            static_cast<float*>(kv_->get_raw_buffer_v()), att_, pos_dev_,
                                            attn_bound, attn_fixed, c.head_local, c.kv_mul, c.head_size, kl,
                                            c.seq_len, l, hip_->attn_workspace, hip_->attn_workspace_bytes,
                                            n0, wo_bytes, g0, n1, b1, g1, st));
    }
    // row-parallel: every rank holds a partial sum; rank 0 carries the residual into the sum
    const float* res = c.plan.carries_residual ? x_ : nullptr;
    ST_RET(gemv(lw.wo, KX_GEMV_PLAIN, att_, nullptr, x_, nullptr, nullptr, 0, 0, res, st));
    if (all) ST_RET(all_reduce(x_, c.dim));
    ST_RET(gemv(lw.w13, KX_GEMV_GATE_UP, x_, lw.ffn_norm, h_, nullptr, nullptr, 0, 0, nullptr, st));
    ST_RET(gemv(lw.w2, KX_GEMV_PLAIN, h_, nullptr, x_, nullptr, nullptr, 0, 0, res, st));
    if (all) ST_RET(all_reduce(x_, c.dim));
  }
  if (!all) return gemv(cls_, KX_GEMV_PLAIN, x_, final_norm_, logits_ + c.vocab_row0, nullptr, nullptr, 0,
                        0, nullptr, st);
  // greedy sample; a prompt step keeps the given next token and only advances the position
  return classify_and_pick(x_, sample, st);
}

base::Status LlamaModel::classify_and_pick(const float* x_row, bool write_history, void* st) {
  const TransformerConfig& c = cfg_;
  float* local = logits_ + c.vocab_row0;
  // the step's last launch also gathers the next step's input row: emb[history[pos + 1]]
  const bool embed = quant_ && !prof_on_;
  const float* table = embed ? emb_ : nullptr;
  const bool one_rank_route = !c.cls_split && !(comm_ready_ && c.tp_size == 1);
  if (one_rank_route && !pick_unfused_ && getenv("KX_NO_PICK_FUSE") == nullptr) {  // (the variable: A/B runs)
    // classifier GEMV, greedy pick, bookkeeping and the next embedding row as one launch; shapes
    // it does not take fall through to the separate launches for good
    kx_gemv_w8_desc d{};
    d.mode = KX_GEMV_PLAIN;
    d.group_size = c.group_size;
    d.M = cls_.cols;
    d.K = cls_.rows;
    d.x = x_row;
    d.w = static_cast<const int8_t*>(cls_.w);
    d.scales = cls_.s;
    d.norm_w = final_norm_;
    d.norm_eps = c.rms_eps;
    d.y0 = local;
    d.k0 = d.k1 = cls_.rows;
    d.bias = cls_.bias;
    const int rc = kx_gemv_w8_classify_pick_f32(&d, token_dev_, pos_dev_, history_dev_, c.seq_len + 1, write_history,
                                                table, c.dim, c.vocab_size, x_, step_ws_,
                                                kx_step_workspace_bytes(), st);
    if (rc == KX_OK) {
      x_valid_ = embed;
      return base::error::Success();
    }
    if (rc != KX_EUNSUPPORTED) KX_RET(rc);
    pick_unfused_ = true;
  }
  ST_RET(gemv(cls_, KX_GEMV_PLAIN, x_row, final_norm_, local, nullptr, nullptr, 0, 0, nullptr, st));
  // a communicator of one rank takes the candidate route too, so that route (its extra launches
  // and its collective) can be exercised, graph capture included, on a single GPU
  if (one_rank_route) {
    KX_RET(kx_decode_step_end_embed(local, c.vocab_local, token_dev_, pos_dev_, history_dev_, c.seq_len + 1,
                                    write_history, table, c.dim, c.vocab_size, x_, step_ws_,
                                    kx_step_workspace_bytes(), st));
    x_valid_ = embed;
    return base::error::Success();
  }
  // this rank's best {value, global index} into slot (1 + rank); all-gather the slots; same pick everywhere
  float* slots = cand_ + 2;
  KX_RET(kx_argmax_candidate_f32(local, c.vocab_local, c.vocab_row0, slots + 2 * c.tp_rank, step_ws_,
                                 kx_step_workspace_bytes(), st));
  ST_RET(all_gather(slots, 2));
  KX_RET(kx_decode_step_end_candidates_embed(slots, c.tp_size, token_dev_, pos_dev_, history_dev_,
                                             c.seq_len + 1, write_history, table, c.dim, c.vocab_size, x_, st));
  x_valid_ = embed;
  return base::error::Success();
}

base::Status LlamaModel::all_gather(float* buf, size_t count) {
  if (loop_) return loop_->all_gather(cfg_.tp_rank, buf, count, hip_->stream);
  if (peer_connected_ && count <= peer_cap_) {
    KX_RET(kx_peer_allgather_f32(peer_, buf, count, hip_->stream));
    return base::error::Success();
  }
  if (cfg_.tp_size <= 1 && !comm_ready_) return base::error::Success();
  if (!comm_ready_) return base::error::InternalError("tensor-parallel model without init_comm()");
  auto comm = base::RcclHandler::get_instance();
  KX_RET(kx_event_record(hip_->compute_done, hip_->stream));
  KX_RET(kx_stream_wait_event(hip_->comm_stream, hip_->compute_done));
  ST_RET(comm->AllGather(buf + (size_t)cfg_.tp_rank * count, buf, count, base::DataType::kDataTypeFp32,
                         hip_->comm_stream));
  KX_RET(kx_event_record(hip_->comm_done, hip_->comm_stream));
  KX_RET(kx_stream_wait_event(hip_->stream, hip_->comm_done));
  return base::error::Success();
}

base::Status LlamaModel::gather_logits() {
  if (!cfg_.cls_split) return base::error::Success();
  return all_gather(logits_, (size_t)cfg_.vocab_local);
}

base::Status LlamaModel::set_state(int32_t token, int32_t pos) {
  if (pos < 0 || pos >= cfg_.seq_len) return base::error::InvalidArgument("pos outside seq_len");
  KX_RET(kx_stream_sync(hip_->stream));
  KX_RET(kx_memcpy_h2d(token_dev_, &token, 4, nullptr));
  KX_RET(kx_memcpy_h2d(pos_dev_, &pos, 4, nullptr));
  KX_RET(kx_memcpy_h2d(history_dev_ + pos, &token, 4, nullptr));
  x_valid_ = false;
  return base::error::Success();
}

base::Status LlamaModel::step_logits(int32_t token, int32_t pos, float* logits_host) {
  ST_RET(set_state(token, pos));
  attn_bound_ = pos;
  base::Status launched = launch_step(hip_->stream, true);
  attn_bound_ = -1;
  ST_RET(launched);
  ST_RET(gather_logits());
  KX_RET(kx_stream_sync(hip_->stream));
  if (logits_host) KX_RET(kx_memcpy_d2h(logits_host, logits_, (size_t)cfg_.vocab_size * 4, nullptr));
  return base::error::Success();
}

base::Status LlamaModel::decode(int32_t steps, bool use_graph) {
  if (steps <= 0) return base::error::Success();
  int32_t pos = 0;
  KX_RET(kx_stream_sync(hip_->stream));
  KX_RET(kx_memcpy_d2h(&pos, pos_dev_, 4, nullptr));
  if (pos + steps > cfg_.seq_len)
    return base::error::InvalidArgument("decode would run past seq_len");
  void* st = hip_->stream;
  if (use_graph && loop_)
    return base::error::InvalidArgument("the loopback communicator cannot be captured into a graph");
  // the captured step has no embedding launch: its input row is left by the step before it
  if (use_graph && !x_valid_ && quant_) {
    KX_RET(kx_embedding_dpos_f32(history_dev_, pos_dev_, emb_, cfg_.dim, cfg_.vocab_size, x_, st));
    x_valid_ = true;
  }
  auto step_at = [&](int32_t at) {  // one eager step at a known position
    attn_bound_ = at;
    base::Status s = launch_step(st, true);
    attn_bound_ = -1;
    return s;
  };
  if (!use_graph) {
    for (int32_t i = 0; i < steps; ++i) ST_RET(step_at(pos + i));
    return base::error::Success();
  }
  const char* span_env = getenv("KX_ATTN_SPAN");  // tests and A/B runs
  const int32_t span = span_env ? std::max(1, atoi(span_env)) : kAttnSpan;
  while (steps > 0) {
    // The captured step serves the positions up to graph_bound_ (its attention launches are cut
    // for that row and read every row up to it); past them, or with a context that has become
    // shorter by more than a span, the step is captured again.
    if (graph_ && (pos > graph_bound_ || pos + 2 * span <= graph_bound_)) {
      kx_graph_destroy(graph_);
      graph_ = nullptr;
    }
    if (!graph_) {
      // one eager step first: lazy per-kernel setup (e.g. raising a kernel's LDS limit) must
      // not happen while the stream is capturing
      ST_RET(step_at(pos));
      ++pos;
      if (--steps == 0) break;
      const int32_t bound = std::min(pos + span - 1, cfg_.seq_len - 1);
      KX_RET(kx_graph_begin_capture(st));
      attn_bound_ = bound;
      base::Status s = launch_step(st, true);
      attn_bound_ = -1;
      void* g = nullptr;
      const int rc = kx_graph_end_capture(st, &g);
      if (!s) return s;
      KX_RET(rc);
      graph_ = g;
      graph_bound_ = bound;
    }
    const int32_t n = std::min(steps, graph_bound_ - pos + 1);
    for (int32_t i = 0; i < n; ++i) KX_RET(kx_graph_launch(graph_, st));
    steps -= n;
    pos += n;
  }
  return base::error::Success();
}

base::Status LlamaModel::prefill_sequential(const int32_t* tokens, int32_t n, int32_t pos0,
                                            float* logits_host) {
  // The prompt is laid into the token history; each row then runs the fused decode step without
  // sampling -- exactly the reference's prompt phase, one token per step (demo/main.cpp:18-41).
  KX_RET(kx_stream_sync(hip_->stream));
  KX_RET(kx_memcpy_h2d(history_dev_ + pos0, tokens, (size_t)n * 4, nullptr));
  KX_RET(kx_memcpy_h2d(pos_dev_, &pos0, 4, nullptr));
  x_valid_ = false;
  for (int32_t i = 0; i < n; ++i) {
    attn_bound_ = pos0 + i;
    base::Status s = launch_step(hip_->stream, false);
    attn_bound_ = -1;
    ST_RET(s);
  }
  ST_RET(gather_logits());
  KX_RET(kx_stream_sync(hip_->stream));
  // leave the state ready for decode(): the sampled token of the last prompt row
  int32_t next = 0;
  KX_RET(kx_memcpy_d2h(&next, token_dev_, 4, nullptr));
  KX_RET(kx_memcpy_h2d(history_dev_ + pos0 + n, &next, 4, nullptr));
  x_valid_ = false;  // the last row's step end ran before history[pos0 + n] was known
  if (logits_host) KX_RET(kx_memcpy_d2h(logits_host, logits_, (size_t)cfg_.vocab_size * 4, nullptr));
  return base::error::Success();
}

base::Status LlamaModel::prefill(const int32_t* tokens, int32_t n, int32_t pos0, float* logits_host,
                                 bool exact) {
  const TransformerConfig& c = cfg_;
  if (!tokens || n <= 0 || pos0 < 0 || pos0 + n > c.seq_len)
    return base::error::InvalidArgument("prompt does not fit in seq_len");
  if (!quant_) return base::error::FunctionNotImplement("the fused path needs int8 weights");
  if (exact || n < 4) return prefill_sequential(tokens, n, pos0, logits_host);

  // Batched prompt: rows go through the MFMA GEMMs and the flash-attention kernel in chunks.
  void* st = hip_->stream;
  const int32_t ql = c.q_dim_local, kl = c.kv_dim_local, hl = c.hidden_local, dim = c.dim;
  int32_t chunk = n < kPrefillChunk ? n : kPrefillChunk;
  {
    // A prompt of 193..304 rows on a model WITHOUT fp16 weight images (policy 0, or images that do
    // not fit) runs as two pieces of at most 192 rows through the few-rows kernel, which streams the
    // int8 weights once per piece (Llama-3-8B: 224 / 256 / 288 rows in 11.0 / 11.4 / 12.6 ms against
    // 18.5 / 18.9 / 20 ms for the int8 single-pass kernel those lengths got before).  From 305 rows
    // converting the weights per pass and running the fp16 GEMMs is faster (320 / 352 / 384 rows:
    // 12.8 / 13.2 / 13.3 ms against 13.2 / 14.0 / 14.6 ms in pieces, one box; equal at 304); with
    // kept images the fp16 GEMMs win from 193 rows.
    const int gs0 = c.group_size;
    const bool attn16 = c.head_size == 32 || c.head_size == 64 || c.head_size == 128;
    const bool rows_ok = gs0 == 64 && dim % 256 == 0 && ql % 256 == 0 && hl % 256 == 0 && attn16 &&
                         getenv("KX_PREFILL_NO_ROWS_KERNEL") == nullptr && getenv("KX_PREFILL_ROWS_MAX") == nullptr;
    const bool f16_shapes = (gs0 & (gs0 - 1)) == 0 && gs0 >= 16 && dim % 64 == 0 && ql % 64 == 0 && hl % 64 == 0 &&
                            dim % gs0 == 0 && ql % gs0 == 0 && hl % gs0 == 0 && attn16;
    if (rows_ok && n > 192 && n <= kPiecesMaxRows && getenv("KX_PREFILL_NO_PIECES") == nullptr) {
      if (f16_shapes) ST_RET(reserve_weight_images());  // the decision a 193+ row prompt makes anyway
      if (weight_images_ != 1) {
        const int32_t pieces = (n + 191) / 192;
        chunk = ((n + pieces - 1) / pieces + 15) / 16 * 16;
        if (chunk > 192) chunk = 192;
      }
    }
  }
  // scratch: X, XN [chunk][dim]; Q, ATT [chunk][ql]; H13 [chunk][2hl]; HH [chunk][hl]; then the
  // fp16 workspace of the two-pass GEMM, sized for the widest matrix of a layer
  // with a 16-bit cache the K/V rows of a chunk pass through fp32 scratch (RoPE runs there) and
  // are rounded into the cache afterwards: KR, VR [chunk][kl]
  const size_t act_bytes =
      (size_t)chunk * (2 * (size_t)dim + 2 * (size_t)ql + 3 * (size_t)hl + (c.kv_f16 ? 2 * (size_t)kl : 0)) * 4;
  const size_t act_pad = (act_bytes + 255) / 256 * 256;
  size_t gemm_ws = 0;
  for (size_t b : {kx_gemm_w8_workspace_bytes(chunk, dim, ql + 2 * kl),
                   kx_gemm_w8_workspace_bytes(chunk, ql, dim),
                   kx_gemm_w8_workspace_bytes(chunk, dim, 2 * hl),
                   kx_gemm_w8_workspace_bytes(chunk, hl, dim)})
    gemm_ws = b > gemm_ws ? b : gemm_ws;
  // and the slabs + tickets through which the GEMM combines reductions it cut into parts
  const size_t gemm_pad = (gemm_ws + 255) / 256 * 256;
  const size_t split_ws_bytes = kx_gemm_split_workspace_bytes();
  const size_t need = act_pad + gemm_pad + split_ws_bytes;
  if (need > pf_bytes_) {
    KX_RET(kx_stream_sync(st));
    dev_free(pf_);
    pf_bytes_ = 0;
    void* p = nullptr;
    KX_RET(kx_malloc(&p, need));
    pf_ = static_cast<float*>(p);
    pf_bytes_ = need;
  }
  float* X = pf_;
  float* XN = X + (size_t)chunk * dim;
  float* Q = XN + (size_t)chunk * dim;
  float* ATT = Q + (size_t)chunk * ql;
  float* H13 = ATT + (size_t)chunk * ql;
  float* HH = H13 + (size_t)chunk * 2 * hl;
  float* KR = HH + (size_t)chunk * hl;  // only with a 16-bit cache
  float* VR = KR + (size_t)chunk * kl;
  void* gws = reinterpret_cast<char*>(pf_) + act_pad;
  void* split_ws = reinterpret_cast<char*>(pf_) + act_pad + gemm_pad;
  const int gs = c.group_size;
  KX_RET(kx_stream_sync(st));
  KX_RET(kx_memset(split_ws, 0, 4096, st));  // arrival tickets start at zero (the scratch layout moves with the chunk size)
  KX_RET(kx_memcpy_h2d(history_dev_ + pos0, tokens, (size_t)n * 4, nullptr));
  float* kbase = static_cast<float*>(kv_->get_raw_buffer_k());  // half rows when c.kv_f16
  float* vbase = static_cast<float*>(kv_->get_raw_buffer_v());
  const float* res = c.plan.carries_residual ? X : nullptr;
  // attention over the cache in either element type; out16 selects fp16 output rows
  auto attention = [&](float* out32, void* out16, int32_t rows_, int32_t last_pos, int32_t layer) -> int {
    const float scale = 1.0f / std::sqrt((float)c.head_size);
    if (c.kv_f16)
      return kx_flash_attention_kv16(Q, kbase, vbase, out32, out16, rows_, c.head_local, c.kv_head_local,
                                     c.head_size, c.seq_len, last_pos, layer, scale, 1,
                                     hip_->attn_workspace, hip_->attn_workspace_bytes, st);
    if (out16)
      return kx_flash_attention_f16out(Q, kbase, vbase, out16, rows_, c.head_local, c.kv_head_local,
                                       c.head_size, c.seq_len, last_pos, layer, scale, 1,
                                       hip_->attn_workspace, hip_->attn_workspace_bytes, st);
    return kx_flash_attention_f32(Q, kbase, vbase, out32, rows_, c.head_local, c.kv_head_local,
                                  c.head_size, c.seq_len, last_pos, layer, scale, 1,
                                  hip_->attn_workspace, hip_->attn_workspace_bytes, st);
  };
  // files the chunk's finished K/V rows into a 16-bit cache
  auto file_kv16 = [&](void* k_ptr, void* v_ptr, int32_t p0_, int32_t rows_) -> int {
    uint16_t* k16 = static_cast<uint16_t*>(k_ptr) + (size_t)p0_ * kl;
    uint16_t* v16 = static_cast<uint16_t*>(v_ptr) + (size_t)p0_ * kl;
    if (int rc = kx_f32_to_f16_rows(KR, kl, k16, kl, kl, rows_, st)) return rc;
    return kx_f32_to_f16_rows(VR, kl, v16, kl, kl, rows_, st);
  };
  // shapes the fp16 kernels are built for (LDS-DMA K-steps of 64, 32-unit SwiGLU blocks)
  const bool f16_ok = (gs & (gs - 1)) == 0 && gs >= 16 && dim % 64 == 0 && ql % 64 == 0 &&
                      hl % 64 == 0 && dim % gs == 0 && ql % gs == 0 && hl % gs == 0 &&
                      (c.head_size == 32 || c.head_size == 64 || c.head_size == 128);
  // A/B switches (tools and tests), read once per call
  const char* env_rows_max = getenv("KX_PREFILL_ROWS_MAX");
  const bool env_no_rows = getenv("KX_PREFILL_NO_ROWS_KERNEL") != nullptr;
  const bool env_unfused = getenv("KX_PREFILL_ROWS_UNFUSED") != nullptr;
  const char* env_f16_min = getenv("KX_PREFILL_F16_MIN_ROWS");
  // the fp16-output attention kernel the 16-bit forms end in exists for these head sizes only
  const bool attn16_ok = c.head_size == 32 || c.head_size == 64 || c.head_size == 128;
  int32_t rows = 0;
  for (int32_t p0 = pos0; p0 < pos0 + n; p0 += rows) {
    rows = pos0 + n - p0 < chunk ? pos0 + n - p0 : chunk;
    KX_RET(kx_embedding_f32(history_dev_ + p0, rows, emb_, dim, c.vocab_size, X, st));
    // per-layer GEMM time of the two forms on the 8B matrices ties at 512..640 rows; the fp16 form
    // also saves the separate SwiGLU pass
    // ... while the fp16 weight images have to be made per pass.  A model that keeps its images
    // (prefill_weight_cache, decided here at the first prompt that is not run by the few-rows
    // kernel below and has 32+ rows) pays the conversion once
    // per matrix, and from then on the fp16 form wins from 32 rows up (Llama-3-8B, 32 / 128 / 384
    // rows: 6.2 / 9.1 / 11.7 ms against 16.9 / 18.4 / 22.4 ms for the int8 single-pass kernel).
    // Which form a prompt takes depends on its length and on that one decision only, never on
    // which prompts came before it.
    // Up to 192 rows the int8 weights are streamed once against all rows (kx_gemm_w8_rows_f16act):
    // no fp16 image involved (Llama-3-8B, 16 / 32 / 64 / 96 / 128 / 192 rows: 3.2 / 3.6 / 4.6 / 5.6 /
    // 5.8 / 7.4 ms against 16.4 / 6.0 / 7.0 / 8.3 / 9.0 / 10.7 ms for the tiled kernels).
    const int32_t few_rows_max = env_rows_max ? atoi(env_rows_max) : 192;
    const bool few_rows = rows <= few_rows_max && gs == 64 && dim % 256 == 0 && ql % 256 == 0 && hl % 256 == 0 &&
                          attn16_ok && !env_no_rows;
    // SwiGLU, and bias + RoPE + K/V placement, ride in the few-rows products' launches where the shapes
    // allow (whole 128-row tiles of q, k and v rows)
    const bool rows_fused = few_rows && !env_unfused;
    const bool rows_fused_rope = rows_fused && ql % 128 == 0 && kl % 128 == 0;
    if (!few_rows && f16_ok && rows >= 32) ST_RET(reserve_weight_images());
    int32_t f16_min_rows = weight_images_ == 1 ? 32 : kPiecesMaxRows + 1;
    if (env_f16_min) f16_min_rows = atoi(env_f16_min);
    const bool f16_chunk = !few_rows && f16_ok && rows >= f16_min_rows;
    for (int32_t l = 0; l < c.layer_num; ++l) {
      LayerWeights& lw = layers_[l];
      auto [k_ptr, v_ptr] = kv_->get_cache_ptr(l);
      float* krow = c.kv_f16 ? KR : static_cast<float*>(k_ptr) + (size_t)p0 * kl;
      float* vrow = c.kv_f16 ? VR : static_cast<float*>(v_ptr) + (size_t)p0 * kl;
      const float* bq = lw.qkv.bias;  // q | k | v biases of the layer, or none
      const float* bk = bq ? bq + ql : nullptr;
      const float* bv = bq ? bq + ql + kl : nullptr;
      if (few_rows) {
        // fp16 activation rows (XN / ATT / HH reused as fp16 rows), fp32 results; bias and RoPE,
        // SwiGLU and the residual sums as in the other forms
        auto rows_gemm = [&](const void* x16, int64_t ldx, Linear& lin, float* y0, int64_t ld0, float* y1,
                             int64_t ld1, float* y2, int64_t ld2, int32_t k0, int32_t k1, const float* resid,
                             int32_t Mm, int32_t Kk) -> int {
          return kx_gemm_w8_rows_f16act(x16, ldx, static_cast<const int8_t*>(lin.w), lin.s, gs, y0, ld0, y1, ld1,
                                        y2, ld2, k0, k1, resid, dim, rows, Mm, Kk, split_ws, split_ws_bytes, st);
        };
        KX_RET(kx_rmsnorm_f16out(X, lw.attn_norm, XN, dim, rows, c.rms_eps, st));
        if (rows_fused_rope) {
          // bias, RoPE and the K/V row placement on the finished totals, in the product's launch
          KX_RET(kx_gemm_w8_rows_qkv_rope_f16act(XN, dim, static_cast<const int8_t*>(lw.qkv.w), lw.qkv.s, gs, Q, ql,
                                                 krow, kl, vrow, kl, ql, kl, c.head_size, c.rope_variant, sin_,
                                                 cos_, p0, bq, rows, dim, split_ws, split_ws_bytes, st));
        } else {
          KX_RET(rows_gemm(XN, dim, lw.qkv, Q, ql, krow, kl, vrow, kl, ql, ql + kl, nullptr, dim, lw.qkv.rows));
          KX_RET(kx_rope_bias_f32(c.rope_variant, ql, kl, c.head_size, Q, krow, kl, vrow, kl, bq, bk, bv,
                                  p0, rows, sin_, cos_, st));
        }
        if (c.kv_f16) KX_RET(file_kv16(k_ptr, v_ptr, p0, rows));
        KX_RET(attention(nullptr, ATT, rows, p0 + rows - 1, l));
        KX_RET(rows_gemm(ATT, ql, lw.wo, X, dim, nullptr, 0, nullptr, 0, dim, dim, res, ql, dim));
        ST_RET(all_reduce(X, (size_t)rows * dim));
        KX_RET(kx_rmsnorm_f16out(X, lw.ffn_norm, XN, dim, rows, c.rms_eps, st));
        if (rows_fused) {
          KX_RET(kx_gemm_w8_rows_swiglu_f16act(XN, dim, static_cast<const int8_t*>(lw.w13.w), lw.w13.s, gs, HH, hl,
                                               rows, dim, hl, split_ws, split_ws_bytes, st));
        } else {
          KX_RET(rows_gemm(XN, dim, lw.w13, H13, 2 * hl, nullptr, 0, nullptr, 0, 2 * hl, 2 * hl, nullptr, dim,
                           2 * hl));
          KX_RET(kx_swiglu_rows_f16out(H13, 2 * hl, hl, HH, hl, rows, st));
        }
        KX_RET(rows_gemm(HH, hl, lw.w2, X, dim, nullptr, 0, nullptr, 0, dim, dim, res, hl, dim));
        ST_RET(all_reduce(X, (size_t)rows * dim));
        continue;
      }
      if (f16_chunk) {
        // fp16 weight image: the matrix's own kept one (made by the first pass over it), else scratch
        auto gemm16 = [&](const void* x16, Linear& lin, int mode, void* h16, float* y0, int64_t ld0,
                          float* y1, int64_t ld1, float* y2, int64_t ld2, int32_t k0, int32_t k1,
                          const float* resid, int64_t ldr, int32_t Mm, int32_t Kk, int32_t T_ = 0,
                          bool scratch_holds_it = false) -> int {
          void* wsp = lin.w16 ? lin.w16 : gws;
          const size_t wsb = lin.w16 ? lin.w16_bytes : gemm_ws;
          const bool ready = lin.w16 ? lin.w16_ready : scratch_holds_it;
          const int rc = kx_gemm_w8_f16act_split_ws(x16, static_cast<const int8_t*>(lin.w), lin.s, gs,
                                                    mode | (ready ? KX_GEMM_WEIGHTS_READY : 0), h16, y0, ld0,
                                                    y1, ld1, y2, ld2, k0, k1, resid, ldr, T_ > 0 ? T_ : rows,
                                                    Mm, Kk, wsp, wsb, split_ws, split_ws_bytes, st);
          if (rc == 0 && lin.w16) lin.w16_ready = true;
          return rc;
        };
        // Row-parallel product (Wo, W2) under tensor parallelism: the rows go through the GEMM in
        // sub-blocks, and the all-reduce of sub-block j runs on the side stream while the GEMM of
        // sub-block j+1 runs here; only the last sub-block's reduce is waited for with nothing beside it.
        auto row_parallel = [&](const void* x16, int64_t x_ld_halves, Linear& lin, int32_t Mm) -> base::Status {
          const bool reducing = loop_ || comm_ready_ || c.tp_size > 1;
          int32_t nsub = reducing ? prefill_subblocks_ : 1;
          if (nsub > rows / 1024) nsub = rows / 1024;
          if (nsub < 1) nsub = 1;
          const int32_t per = ((rows + nsub - 1) / nsub + 255) / 256 * 256;
          for (int32_t r0 = 0, j = 0; r0 < rows; r0 += per, ++j) {
            const int32_t nr = rows - r0 < per ? rows - r0 : per;
            const char* xin = static_cast<const char*>(x16) + (size_t)r0 * x_ld_halves * 2;
            KX_RET(gemm16(xin, lin, KX_GEMM_ROUTE_F32, nullptr, X + (size_t)r0 * dim, dim, nullptr, 0, nullptr,
                          0, dim, dim, res ? res + (size_t)r0 * dim : nullptr, dim, Mm, dim, nr, j > 0));
            ST_RET(all_reduce_begin(X + (size_t)r0 * dim, (size_t)nr * dim));
          }
          return all_reduce_end();
        };
        // long chunks keep the GEMM inputs in fp16: the norm and the attention kernel write fp16,
        // SwiGLU runs in the epilogue of the W1|W3 GEMM; XN / ATT / HH are reused as fp16 rows
        KX_RET(kx_rmsnorm_f16out(X, lw.attn_norm, XN, dim, rows, c.rms_eps, st));
        if (c.head_size == 64 || c.head_size == 128) {
          // bias, RoPE and the K/V row placement ride in the epilogue of the Wq|Wk|Wv product
          Linear& lin = lw.qkv;
          void* wsp = lin.w16 ? lin.w16 : gws;
          const size_t wsb = lin.w16 ? lin.w16_bytes : gemm_ws;
          const bool ready = lin.w16 && lin.w16_ready;
          KX_RET(kx_gemm_w8_f16act_qkv_rope(XN, static_cast<const int8_t*>(lin.w), lin.s, gs, ready, Q, ql, krow,
                                            kl, vrow, kl, ql, kl, c.head_size, c.rope_variant, sin_, cos_, p0,
                                            bq, rows, dim, wsp, wsb, split_ws, split_ws_bytes, st));
          if (lin.w16) lin.w16_ready = true;
        } else {
          KX_RET(gemm16(XN, lw.qkv, KX_GEMM_ROUTE_F32, nullptr, Q, ql, krow, kl, vrow, kl, ql, ql + kl,
                        nullptr, 0, dim, lw.qkv.rows));
          KX_RET(kx_rope_bias_f32(c.rope_variant, ql, kl, c.head_size, Q, krow, kl, vrow, kl, bq, bk, bv,
                                  p0, rows, sin_, cos_, st));
        }
        if (c.kv_f16) KX_RET(file_kv16(k_ptr, v_ptr, p0, rows));
        KX_RET(attention(nullptr, ATT, rows, p0 + rows - 1, l));
        ST_RET(row_parallel(ATT, ql, lw.wo, ql));
        KX_RET(kx_rmsnorm_f16out(X, lw.ffn_norm, XN, dim, rows, c.rms_eps, st));
        KX_RET(gemm16(XN, lw.w13, KX_GEMM_SWIGLU_F16, HH, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, nullptr,
                      0, dim, 2 * hl));
        ST_RET(row_parallel(HH, hl, lw.w2, hl));
        continue;
      }
      KX_RET(kx_rmsnorm_f32(X, lw.attn_norm, XN, dim, rows, c.rms_eps, st));
      KX_RET(kx_gemm_w8_f32_ws(XN, dim, static_cast<const int8_t*>(lw.qkv.w), lw.qkv.s, gs, Q, ql,
                               krow, kl, vrow, kl, ql, ql + kl, nullptr, 0, rows, dim,
                               lw.qkv.rows, gws, gemm_ws, st));
      KX_RET(kx_rope_bias_f32(c.rope_variant, ql, kl, c.head_size, Q, krow, kl, vrow, kl, bq, bk, bv,
                              p0, rows, sin_, cos_, st));
      if (c.kv_f16) KX_RET(file_kv16(k_ptr, v_ptr, p0, rows));
      KX_RET(attention(ATT, nullptr, rows, p0 + rows - 1, l));
      KX_RET(kx_gemm_w8_f32_ws(ATT, ql, static_cast<const int8_t*>(lw.wo.w), lw.wo.s, gs, X, dim,
                               nullptr, 0, nullptr, 0, dim, dim, res, dim, rows, ql, dim, gws,
                               gemm_ws, st));
      ST_RET(all_reduce(X, (size_t)rows * dim));
      KX_RET(kx_rmsnorm_f32(X, lw.ffn_norm, XN, dim, rows, c.rms_eps, st));
      KX_RET(kx_gemm_w8_f32_ws(XN, dim, static_cast<const int8_t*>(lw.w13.w), lw.w13.s, gs, H13,
                               2 * hl, nullptr, 0, nullptr, 0, 2 * hl, 2 * hl, nullptr, 0, rows, dim,
                               2 * hl, gws, gemm_ws, st));
      KX_RET(kx_swiglu_rows_f32(H13, 2 * hl, hl, HH, hl, rows, st));
      KX_RET(kx_gemm_w8_f32_ws(HH, hl, static_cast<const int8_t*>(lw.w2.w), lw.w2.s, gs, X, dim,
                               nullptr, 0, nullptr, 0, dim, dim, res, dim, rows, hl, dim, gws,
                               gemm_ws, st));
      ST_RET(all_reduce(X, (size_t)rows * dim));
    }
  }
  // classifier on the last row only (cls_logits, llama3.cpp:790-799), then the greedy pick
  const int32_t last_pos = pos0 + n - 1;
  KX_RET(kx_memcpy_h2d(pos_dev_, &last_pos, 4, st));
  ST_RET(classify_and_pick(X + (size_t)(rows - 1) * dim, true, st));
  ST_RET(gather_logits());
  KX_RET(kx_stream_sync(st));
  if (logits_host) KX_RET(kx_memcpy_d2h(logits_host, logits_, (size_t)c.vocab_size * 4, nullptr));
  return base::error::Success();
}

base::Status LlamaModel::read_history(int32_t* out, int32_t n) const {
  if (!out || n < 0 || n > cfg_.seq_len + 1) return base::error::InvalidArgument("bad history range");
  KX_RET(kx_stream_sync(hip_->stream));
  KX_RET(kx_memcpy_d2h(out, history_dev_, (size_t)n * 4, nullptr));
  return base::error::Success();
}

base::Status LlamaModel::read_state(int32_t* token, int32_t* pos) const {
  KX_RET(kx_stream_sync(hip_->stream));
  if (token) KX_RET(kx_memcpy_d2h(token, token_dev_, 4, nullptr));
  if (pos) KX_RET(kx_memcpy_d2h(pos, pos_dev_, 4, nullptr));
  return base::error::Success();
}

// IEEE half -> float on the host (test / inspection helper only)
static float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: normalise
      int e = -1;
      do { ++e; man <<= 1; } while (!(man & 0x400u));
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

base::Status LlamaModel::read_kv(int32_t layer, int32_t rows, float* k_host, float* v_host) const {
  if (layer < 0 || layer >= cfg_.layer_num || rows < 0 || rows > cfg_.seq_len)
    return base::error::InvalidArgument("bad kv range");
  auto [k_ptr, v_ptr] = kv_->get_cache_ptr(layer);
  KX_RET(kx_stream_sync(hip_->stream));
  const size_t count = (size_t)rows * cfg_.kv_dim_local;
  if (!cfg_.kv_f16) {
    if (k_host) KX_RET(kx_memcpy_d2h(k_host, k_ptr, count * 4, nullptr));
    if (v_host) KX_RET(kx_memcpy_d2h(v_host, v_ptr, count * 4, nullptr));
    return base::error::Success();
  }
  std::vector<uint16_t> tmp(count);
  for (int which = 0; which < 2; ++which) {
    float* dst = which ? v_host : k_host;
    if (!dst || count == 0) continue;
    KX_RET(kx_memcpy_d2h(tmp.data(), which ? v_ptr : k_ptr, count * 2, nullptr));
    for (size_t i = 0; i < count; ++i) dst[i] = half_bits_to_float(tmp[i]);
  }
  return base::error::Success();
}

base::Status LlamaModel::sync() const {
  KX_RET(kx_stream_sync(hip_->stream));
  return base::error::Success();
}

base::Status LlamaModel::profile_allreduce(size_t count, int32_t iters, double* ms) {
  if (!ms || iters <= 0 || count == 0) return base::error::InvalidArgument("bad all-reduce profile request");
  if (!(comm_ready_ || peer_connected_) || loop_)
    return base::error::InvalidArgument("needs an RCCL communicator or connected peer mailboxes");
  void* st = hip_->stream;
  void* buf = nullptr;
  KX_RET(kx_malloc(&buf, count * sizeof(float)));
  KX_RET(kx_memset(buf, 0, count * sizeof(float), st));
  void *e0 = nullptr, *e1 = nullptr;
  KX_RET(kx_event_create(&e0));
  KX_RET(kx_event_create(&e1));
  base::Status s = base::error::Success();
  for (int32_t i = 0; i < 3 && s; ++i) s = all_reduce(static_cast<float*>(buf), count);  // warm-up
  if (s) {
    KX_RET(kx_stream_sync(st));
    KX_RET(kx_event_record(e0, st));
    for (int32_t i = 0; i < iters && s; ++i) s = all_reduce(static_cast<float*>(buf), count);
    KX_RET(kx_event_record(e1, st));
    KX_RET(kx_event_sync(e1));
    float t = 0.f;
    KX_RET(kx_event_elapsed_ms(e0, e1, &t));
    *ms = (double)t / iters;
  }
  kx_event_destroy(e0);
  kx_event_destroy(e1);
  kx_free(buf);
  return s;
}

base::Status LlamaModel::profile_gemv(int32_t steps, int64_t* launches, double* bytes, double* ms) {
  // The W8 GEMV launches of ONE decode step, in model order and on the model's own buffers, are
  // captured alone into a graph (attention, embedding, sampling and collectives left out) and the
  // graph is replayed `steps` times between two HIP events on the compute stream: the average is
  // the launch-to-launch time of the kernel in its dependent chain, with every weight byte
  // streamed cold from HBM (one pass over the weights exceeds the 256 MiB Infinity Cache).
  if (!quant_) return base::error::FunctionNotImplement("the fused path needs int8 weights");
  if (steps <= 0) return base::error::InvalidArgument("steps must be positive");
  void* st = hip_->stream;
  KX_RET(kx_stream_sync(st));
  prof_on_ = true;
  prof_launches_ = 0;
  prof_bytes_sum_ = 0;
  base::Status s = launch_step(st, true);  // eager once (lazy kernel setup), also counts bytes
  const int64_t n_step = prof_launches_;
  const double b_step = prof_bytes_sum_;
  void* g = nullptr;
  if (s) {
    KX_RET(kx_stream_sync(st));
    KX_RET(kx_graph_begin_capture(st));
    s = launch_step(st, true);
    const int rc = kx_graph_end_capture(st, &g);
    if (s && rc) s = base::error::InternalError("graph capture failed");
  }
  prof_on_ = false;
  x_valid_ = false;  // the profiled launches ran on the model's own buffers
  ST_RET(s);
  void *e0 = nullptr, *e1 = nullptr;
  KX_RET(kx_event_create(&e0));
  KX_RET(kx_event_create(&e1));
  KX_RET(kx_graph_launch(g, st));  // warm replay
  KX_RET(kx_event_record(e0, st));
  for (int32_t i = 0; i < steps; ++i) KX_RET(kx_graph_launch(g, st));
  KX_RET(kx_event_record(e1, st));
  KX_RET(kx_stream_sync(st));
  float e = 0;
  KX_RET(kx_event_elapsed_ms(e0, e1, &e));
  kx_event_destroy(e0);
  kx_event_destroy(e1);
  (void)kx_graph_destroy(g);
  if (launches) *launches = n_step * steps;
  if (bytes) *bytes = b_step * steps;
  if (ms) *ms = e;
  return base::error::Success();
}

}  // namespace model
