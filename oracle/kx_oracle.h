/*
 * kx_oracle.h -- CPU restatement of the reference's kernel arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or
 * calls this file; it is the checker that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg compare the HIP path against.
 *
 * Every function names the reference source it restates (paths relative to
 * the reference tree).  Storage types follow the reference (fp32 activations,
 * int8 weights, fp32 scales).  Dot products and softmax sums are accumulated
 * in double and rounded to fp32 once on store: the reference's own CPU kernels
 * sit on Armadillo/BLAS whose summation order is unspecified, so the oracle
 * takes the order-free value and the tests state a tolerance around it.
 */
#ifndef KX_ORACLE_H_
#define KX_ORACLE_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RoPE pairing conventions selected at compile time in the reference
 * (kuiper/source/op/kernels/cpu/rope_kernel.cpp:3,43,83). */
enum { KXO_ROPE_INTERLEAVED = 0, KXO_ROPE_HALF_SPLIT = 1 };

/* tools/export_llama.py:49-73 (quantize_q80): groups run over the flattened
 * row-major matrix; scale = max|w|/127 (fp32); q = round-half-even(w/scale). */
void kxo_quantize_q80(const float* w, size_t n, int32_t group_size, int8_t* q, float* scales);

/* kuiper/source/op/kernels/cuda/matmul_kernel.cu:56-87: y[p] = sum_i (x[i]*s[(p*M+i)/gs])*(float)w[p*M+i].
 * threads<=1 runs serial; otherwise OpenMP over output rows. */
void kxo_gemv_w8(const float* x, const int8_t* w, const float* scales, int32_t group_size, float* y,
                 int32_t M, int32_t K, int32_t threads);

/* kuiper/source/op/kernels/cpu/matmul_kernel.cpp:20-40: y = (W[K,M] . x[M]) * scale. */
void kxo_gemv_f32(const float* x, const float* w, float* y, int32_t M, int32_t K, float scale,
                  int32_t threads);

/* Batched forms: X[T,M] row-major -> Y[T,K]; each row is one kxo_gemv_*. */
void kxo_gemm_w8(const float* X, const int8_t* w, const float* scales, int32_t group_size, float* Y,
                 int32_t T, int32_t M, int32_t K, int32_t threads);
void kxo_gemm_f32(const float* X, const float* w, float* Y, int32_t T, int32_t M, int32_t K,
                  int32_t threads);

/* kuiper/source/op/kernels/cpu/mha_kernel.cpp:5-61 with softmax_kernel.cpp:4-15 and
 * scale_sum_kernel.cpp:14-21.  score may be NULL; when given it receives the
 * probabilities at score[h*seq_len + t], t<=pos, as the reference leaves them. */
void kxo_mha(int32_t pos, int32_t head_num, int32_t layer_index, int32_t seq_len, int32_t kv_dim,
             int32_t kv_mul, int32_t head_size, float* mha_out, const float* query, float* score,
             const float* key_cache, const float* value_cache);

/* FlashAttention layer semantics (kuiper/source/op/flash_attention.cpp:40-65 interface):
 * Tq query rows [Tq][head_num][head_size]; row i sits at absolute position
 * pos-(Tq-1)+i and attends cache rows 0..that position (causal) or 0..pos (not
 * causal); GQA kv head = h / (head_num/kv_head_num).  Tq==1 equals kxo_mha. */
void kxo_flash_attention(const float* query, const float* key_cache, const float* value_cache,
                         float* out, int32_t Tq, int32_t head_num, int32_t kv_head_num,
                         int32_t head_size, int32_t seq_len, int32_t pos, int32_t layer_idx,
                         float softmax_scale, int32_t is_causal, int32_t threads);

/* kuiper/source/op/kernels/cpu/rmsnorm_kernel.cpp:24-32. */
void kxo_rmsnorm(const float* in, const float* weight, float* out, int32_t dim, float eps);

/* kuiper/source/op/kernels/cpu/rope_kernel.cpp:4-16 / 44-56 / 84-96 (theta is the only difference). */
void kxo_sin_cos_cache(int32_t head_size, int32_t max_seq_len, float theta, float* sin_cache,
                       float* cos_cache);
/* rope_kernel.cpp:98-121 (interleaved) and :18-42 / :58-82 (half split). */
void kxo_rope(int32_t variant, int32_t dim, int32_t kv_dim, int32_t head_size, float* q, float* k,
              int32_t pos, const float* sin_cache, const float* cos_cache);

/* kuiper/source/op/kernels/cpu/swiglu_kernel.cpp:21-22: out = in1*sigmoid(in1)*in2. */
void kxo_swiglu(const float* in1, const float* in2, float* out, int32_t n);
/* kuiper/source/op/kernels/cpu/add_kernel.cpp:18. */
void kxo_add(const float* a, const float* b, float* out, int32_t n);
/* kuiper/source/op/kernels/cpu/emb_kernel.cpp:14-27. */
void kxo_embedding(const int32_t* tokens, int32_t n_tokens, const float* table, int32_t dim,
                   float* out);
/* kuiper/source/sampler/argmax_sampler.cpp:6-8 (std::max_element: first maximum wins). */
int32_t kxo_argmax(const float* logits, int32_t n);

/* ---- whole-token graph: kuiper/source/model/llama3.cpp:155-175,637-677,689-799 ---- */
typedef struct {
  int32_t dim, hidden_dim, layer_num, head_num, kv_head_num, vocab_size, seq_len;
  int32_t group_size;    /* 0 = fp32 weights, else int8 group quant */
  int32_t rope_variant;  /* KXO_ROPE_* */
  float rope_theta;
  float rms_eps;
} kxo_config;

typedef struct {
  kxo_config cfg;
  /* per-layer matrices, layer-major; exactly one of f32 / (q8, sc) set per matrix */
  const float *wq_f, *wk_f, *wv_f, *wo_f, *w1_f, *w2_f, *w3_f, *cls_f;
  const int8_t *wq_q, *wk_q, *wv_q, *wo_q, *w1_q, *w2_q, *w3_q, *cls_q;
  const float *wq_s, *wk_s, *wv_s, *wo_s, *w1_s, *w2_s, *w3_s, *cls_s;
  const float* emb;       /* [vocab][dim] */
  const float* attn_norm; /* [L][dim] */
  const float* ffn_norm;  /* [L][dim] */
  const float* final_norm;
  /* optional q/k/v biases, [L][dim] / [L][kv_dim] / [L][kv_dim]; NULL = none.  Added to the
   * matmul output as MatmulLayer does (kuiper/source/op/matmul.cpp:76-79); the Qwen2 fp32 files
   * carry them (kuiper/source/model/qwen2.cpp:306-333, tools/export_qwen2.py:102-110). */
  const float *bq, *bk, *bv;
} kxo_weights;

typedef struct kxo_state kxo_state;

/* KV cache layout: K block then V block, each [layer][seq_len][kv_dim] fp32
 * (what every kernel indexes: cpu/mha_kernel.cpp:10, qwen2.cpp:473-476). */
kxo_state* kxo_state_new(const kxo_config* cfg);
void kxo_state_free(kxo_state* st);
float* kxo_state_key_cache(kxo_state* st);
float* kxo_state_value_cache(kxo_state* st);
float* kxo_state_logits(kxo_state* st);

/* One reference token step at position pos; logits left in kxo_state_logits. */
void kxo_forward_token(const kxo_weights* w, kxo_state* st, int32_t token, int32_t pos,
                       int32_t threads);

/* Parse the exporter's on-disk layouts held in memory (tools/export_llama.py:79-131 legacy fp32,
 * :134-210 quant; reader: kuiper/source/model/model.cpp:52-122, llama3.cpp:201-305,307-440).
 * Returns 0 on success. Pointers alias the blob. */
int kxo_parse_legacy_f32(const void* blob, size_t bytes, kxo_weights* out);
/* The Qwen2 exporter's fp32 layout (tools/export_qwen2.py:79-134; reader qwen2.cpp:290-427): as
 * the legacy fp32 layout, but each layer's wq / wk / wv matrix is followed by its bias vector.
 * De-interleaves into owned storage (never freed: test-sized models). */
int kxo_parse_qwen2_f32(const void* blob, size_t bytes, kxo_weights* out);
int kxo_parse_legacy_q8(const void* blob, size_t bytes, kxo_weights* out);

#ifdef __cplusplus
}
#endif
#endif
