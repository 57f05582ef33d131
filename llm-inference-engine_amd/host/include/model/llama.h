// Llama-family token loop on the HIP device.
//
// Host orchestration of the hot path, re-written rather than mirrored: the reference's
// LLama2Model (kuiper/source/model/llama3.cpp) hand-unrolls one token per predict() call with
// one launch per op and reads the position on the host.  Two paths are offered here:
//
//   forward_layers()  the reference-structured path: one op::Layer::forward per op, through the
//                     kernel registry (attention_rms / attention_qkv / attention_mha /
//                     feed_forward / cls_logits, llama3.cpp:637-799).  This is the drop-in
//                     surface; it exists so the registry and the layers are exercised end to end.
//   decode()          the fused path: per layer [RMSNorm+Wq|Wk|Wv+RoPE+KV write] -> attention ->
//                     [Wo+residual] -> [RMSNorm+W1|W3+SwiGLU] -> [W2+residual]; token id and
//                     position live on the device, the whole step is captured once in a HIP graph
//                     and replayed, so the host is out of the loop.
//   prefill()         T prompt rows at once (MFMA GEMMs + flash attention), new capability; it
//                     leaves the KV cache and last-row logits that T decode steps would.
//
// Tensor parallelism: column-parallel Wq/Wk/Wv/W1/W3, row-parallel Wo/W2 with an all-reduce after
// each (the hook at kuiper/source/op/matmul.cpp:81-127), one process per GPU.
#ifndef KX_HOST_MODEL_LLAMA_H_
#define KX_HOST_MODEL_LLAMA_H_
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "base/kv_cache_manager.h"
#include "base/loopback_comm.h"
#include "base/rccl_handler.h"
#include "op/flash_attention.h"
#include "op/matmul.h"
#include "op/mha.h"
#include "op/support_layers.h"
#include "model/shard_plan.h"

namespace model {

// Header of the exporter's files: 7 x int32 (+ group_size for quant files).
// kuiper/include/model/config.h:4-15, tools/export_llama.py:91-93,148-150.
struct ModelConfig {
  int32_t dim = 0;
  int32_t hidden_dim = 0;
  int32_t layer_num = 0;
  int32_t head_num = 0;
  int32_t kv_head_num = 0;
  int32_t vocab_size = 0;
  int32_t seq_len = 0;
};

struct TransformerConfig {
  int32_t dim = 0, hidden_dim = 0, layer_num = 0, head_num = 0, kv_head_num = 0;
  int32_t vocab_size = 0, seq_len = 0;
  int32_t group_size = 0;    // 0 = fp32 weights
  int32_t rope_variant = 0;  // KX_ROPE_*; the reference's LLAMA3/QWEN2 compile-time switch
  float rope_theta = 10000.f;
  float rms_eps = 1e-5f;
  int32_t tp_size = 1, tp_rank = 0;
  // 16-bit KV cache: rows are kept as IEEE half (one rounding per stored element); the fused token
  // loop and the batched prompt pass support it, the layer-by-layer loop (fp32 views) does not
  bool kv_f16 = false;
  // Qwen2-style attention: fp32 biases on the q/k/v projections.  The fp32 files carry them
  // (tools/export_qwen2.py:102-110, qwen2.cpp:306-333); quant files carry none (qwen2.cpp:192-217),
  // so a quant model starts with zero biases and set_qkv_bias() may attach them.
  bool qkv_bias = false;
  // Long prompts run on fp16 images of the int8 weights (scale applied, one rounding).  0: made
  // per prompt pass in scratch and dropped; 1: kept per matrix after the first pass (2 bytes per
  // weight on top of the int8 copy: 15 GB for Llama-3-8B); -1: kept when they fit in a third of
  // the free device memory at the first long prompt.
  int32_t prefill_weight_cache = -1;
  // derived (model.cpp:132-134)
  int32_t kv_dim = 0, kv_mul = 0, head_size = 0;
  // per-rank sizes under tensor parallelism
  int32_t q_dim_local = 0, kv_dim_local = 0, hidden_local = 0, head_local = 0, kv_head_local = 0;
  // classifier: rows [vocab_row0, vocab_row0 + vocab_local) on this rank.  Split over the ranks
  // when the vocabulary divides (each rank streams 1/tp of the largest matrix of the model; the
  // greedy pick travels as one {value, index} candidate per rank), else replicated.
  int32_t vocab_local = 0, vocab_row0 = 0;
  bool cls_split = false;
  ShardPlan plan;
  base::Status finalize();
};

class LlamaModel {
 public:
  LlamaModel();
  ~LlamaModel();
  LlamaModel(const LlamaModel&) = delete;
  LlamaModel& operator=(const LlamaModel&) = delete;

  // Allocates weights, activation buffers, RoPE tables and the KV cache on the current device.
  base::Status init(const TransformerConfig& cfg);

  // ---- weights ----
  // Seeded random weights generated on the device: int8 uniform in [-127,127] with scales
  // U(1e-4, 3e-4) (or fp32 U(-0.035,0.035) for group_size 0), fp32 embedding, norm weights 1.
  base::Status init_synthetic(uint64_t seed);
  // The exporter's byte layouts: legacy fp32 (tools/export_llama.py:79-131) or quant
  // (:134-210), chosen by cfg.group_size; reader logic of model.cpp:52-122, llama3.cpp:201-440.
  // Tensor-parallel ranks take their shard of every matrix.
  base::Status load_blob(const void* blob, size_t bytes);
  // q/k/v biases of one layer from host arrays of the UNSHARDED sizes ([dim], [kv_dim], [kv_dim];
  // a tensor-parallel rank keeps its heads' rows).  Needs cfg.qkv_bias.
  base::Status set_qkv_bias(int32_t layer, const float* bq, const float* bk, const float* bv);
  // Host copy of one weight as stored on this rank.  name: wq wk wv wo w1 w2 w3 cls (int8 or fp32
  // data), *_s (scales), emb, attn_norm, ffn_norm, final_norm, bq bk bv (biases, cfg.qkv_bias).
  base::Status read_weight(const std::string& name, int32_t layer, void* host_out, size_t bytes) const;

  base::Status init_comm(const base::CommUniqueId* id);
  // Peer mailboxes (kx_peer_*): the token loop's reduces and candidate gathers of up to max_floats
  // floats run as one launch on the compute stream instead of an RCCL call on the side stream;
  // larger messages (prompt blocks) keep the RCCL path.  peer_create() on every rank, the 64-byte
  // handles exchanged by the caller, peer_connect() with all of them by rank, a host barrier.
  base::Status peer_create(size_t max_floats, void* handle64);
  base::Status peer_connect(const void* handles);
  base::Status peer_timed_out(bool* flag) const;
  void peer_disable();  // back to the RCCL path (the mailboxes stay mapped)
  // Test rig: the ranks of a tensor-parallel model as objects of one process (one thread each),
  // summing through a host rendezvous instead of RCCL.  Eager launches only.
  void join_loopback(std::shared_ptr<base::LoopbackGroup> group) { loop_ = std::move(group); }
  // Row sub-blocks per row-parallel GEMM of a long prompt pass under tensor parallelism (>= 1;
  // 1 = one reduce after the whole GEMM, nothing overlapped).
  void set_prefill_subblocks(int32_t n) { prefill_subblocks_ = n < 1 ? 1 : n; }
  // fp16 weight images of the long-prompt GEMMs: 0 made per pass in scratch, 1 kept, -1 kept if
  // they fit a third of the free memory.  Kept images are given back when the policy turns to 0;
  // the next long prompt decides again.
  void set_prefill_weight_cache(int32_t policy);
  size_t weight_image_bytes() const;

  // ---- reference-structured path ----
  base::Status forward_layers(int32_t token, int32_t pos, float* logits_host);

  // ---- fused path ----
  base::Status set_state(int32_t token, int32_t pos);          // device token id + position
  base::Status step_logits(int32_t token, int32_t pos, float* logits_host);  // one fused step
  base::Status decode(int32_t steps, bool use_graph);           // greedy, host out of the loop
  base::Status read_history(int32_t* out, int32_t n) const;     // token at each position
  base::Status read_state(int32_t* token, int32_t* pos) const;
  // Prompt rows pos0..pos0+n-1.  Default: batched (MFMA GEMMs + flash attention, fp16 operand
  // rounding).  exact = true feeds the rows one by one through the fp32 decode step instead.
  base::Status prefill(const int32_t* tokens, int32_t n, int32_t pos0, float* logits_host,
                       bool exact = false);

  base::Status read_kv(int32_t layer, int32_t rows, float* k_host, float* v_host) const;
  base::Status sync() const;

  // Timing of the W8 GEMV launches of a decode step (replayed alone from a graph, `steps` times,
  // between two HIP events on the compute stream): launches, algorithmic bytes, milliseconds.
  base::Status profile_gemv(int32_t steps, int64_t* launches, double* bytes, double* ms);
  // `iters` back-to-back all-reduces of `count` floats on the path the token loop uses (side stream
  // between two events); ms = average per call.  A collective: every rank must call it.
  base::Status profile_allreduce(size_t count, int32_t iters, double* ms);

  const TransformerConfig& config() const { return cfg_; }
  std::shared_ptr<kernel::CudaConfig> hip_config() const { return hip_; }
  double linear_weight_bytes_per_token() const;  // algorithmic bytes of all GEMVs of one step

 private:
  struct Linear {  // one (possibly stacked) matrix per layer on the device
    void* w = nullptr;       // int8 or fp32 [rows][cols]
    float* s = nullptr;      // scales (quant only)
    float* bias = nullptr;   // optional fp32 [rows]
    void* w16 = nullptr;     // kept fp16 image for the long-prompt GEMMs (see prefill_weight_cache)
    size_t w16_bytes = 0;
    bool w16_ready = false;
    int32_t rows = 0, cols = 0;
    size_t w_bytes() const;
  };
  struct LayerWeights {
    Linear qkv, wo, w13, w2;
    float* attn_norm = nullptr;
    float* ffn_norm = nullptr;
  };

  base::Status alloc_linear(Linear& l, int32_t rows, int32_t cols);
  void free_linear(Linear& l);
  base::Status fill_linear_rows(Linear& l, int32_t row_off, int32_t rows, int64_t g_ld,
                                int64_t g_row0, int64_t g_col0, uint64_t seed);
  base::Status upload_matrix(Linear& l, int32_t dst_row_off, const void* src_w, const float* src_s,
                             int32_t g_rows, int32_t g_cols, int32_t row0, int32_t rows,
                             int32_t col0, int32_t cols);
  static constexpr int32_t kPrefillChunk = 4096;  // prompt rows per batched pass
  static constexpr int32_t kPiecesMaxRows = 304;  // longest prompt cut into few-rows pieces (no fp16 images)
  base::Status reserve_weight_images();
  void drop_weight_images(bool free_memory);
  int weight_images_ = 0;  // 0 not decided yet, 1 in use, -1 not available
  base::Status prefill_sequential(const int32_t* tokens, int32_t n, int32_t pos0, float* logits_host);
  void build_layers();
  base::Status launch_step(void* stream, bool with_events);
  base::Status gemv(const Linear& l, int mode, const float* x, const float* norm_w, float* y0,
                    float* y1, float* y2, int32_t k0, int32_t k1, const float* residual,
                    void* stream);
  base::Status all_reduce(float* buf, size_t count);
  base::Status all_reduce_begin(float* buf, size_t count);
  base::Status all_reduce_end();
  base::Status all_gather(float* buf, size_t count_per_rank);  // in place, slice r at buf + r*count
  // final norm + classifier on `x_row`, then the greedy pick and the step bookkeeping
  base::Status classify_and_pick(const float* x_row, bool write_history, void* stream);
  base::Status gather_logits();  // split classifier: every rank's slice into logits_ (a collective)
  float* cand_ = nullptr;        // [1 + tp][2]: this rank's candidate, then everyone's
  bool comm_pending_ = false;
  int32_t prefill_subblocks_ = 4;  // row sub-blocks of a row-parallel prompt GEMM (reduce j beside GEMM j+1)

  TransformerConfig cfg_;
  bool quant_ = false;
  bool comm_ready_ = false;
  void* peer_ = nullptr;        // peer-mailbox group; used once connected
  size_t peer_cap_ = 0;         // floats a peer collective carries at most
  bool peer_connected_ = false;  // collectives within the mailbox size take the peer path
  bool peer_mapped_ = false;
  std::shared_ptr<base::LoopbackGroup> loop_;
  std::shared_ptr<kernel::CudaConfig> hip_;
  std::vector<LayerWeights> layers_;
  Linear cls_;
  float* emb_ = nullptr;
  float* final_norm_ = nullptr;
  std::unique_ptr<base::KVCacheManager> kv_;

  // activations (device)
  float *x_ = nullptr, *q_ = nullptr, *att_ = nullptr, *h_ = nullptr, *logits_ = nullptr;
  float *rms_ = nullptr, *h3_ = nullptr, *tmp_dim_ = nullptr;  // layer-path scratch
  float *sin_ = nullptr, *cos_ = nullptr;
  int32_t *token_dev_ = nullptr, *pos_dev_ = nullptr, *history_dev_ = nullptr;
  float* step_ws_ = nullptr;  // argmax partials of the step-end kernels
  // prefill scratch, grown on demand
  float* pf_ = nullptr;
  size_t pf_bytes_ = 0;

  void* graph_ = nullptr;
  // Upper bound of the position the step launches being issued will read from pos_dev_ (-1 = none:
  // seq_len - 1, cut by the position itself): the decode attention is cut for the bound and reads
  // the position as a mask only, so its cache addresses do not wait for it.  A step issued eagerly
  // knows its position; a captured step serves kAttnSpan positions (graph_bound_: the last one)
  // and is captured again past them, or when the context has become much shorter.
  int32_t attn_bound_ = -1;
  bool pick_unfused_ = false;  // the classifier's shape does not take the launch that also ends the step
  int32_t graph_bound_ = -1;
  static constexpr int32_t kAttnSpan = 256;

  // reference-structured path
  struct RefLayers;
  std::unique_ptr<RefLayers> ref_;

  // profiling (profile_gemv)
  bool prof_on_ = false;
  bool x_valid_ = false;  // x_ holds the embedding of history[pos] (left by the previous step's end)
  int64_t prof_launches_ = 0;
  double prof_bytes_sum_ = 0;
};

}  // namespace model
#endif
