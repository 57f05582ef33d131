/*
 * kx_host.h -- C view of the C++17 host layer (libkx_host.so) for FFI callers (the Python tests
 * and benchmark driver).  The host layer itself is C++: base::, tensor::, op::, kernel::,
 * model:: under host/include; this header only exposes the model driver.
 * Every call returns 0 on success; kxm_last_error() describes the last failure.
 */
#ifndef KX_HOST_H_
#define KX_HOST_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct kxm_config {
  int32_t dim, hidden_dim, layer_num, head_num, kv_head_num, vocab_size, seq_len;
  int32_t group_size;   /* 0 = fp32 weights */
  int32_t rope_variant; /* 0 interleaved, 1 half split */
  float rope_theta;
  float rms_eps;
  int32_t tp_size, tp_rank;
  int32_t kv_f16;       /* non-zero: 16-bit (IEEE half) KV cache instead of the reference's fp32 one */
  int32_t qkv_bias;     /* non-zero: q/k/v projections carry fp32 biases (Qwen2 family) */
  int32_t prefill_weight_cache; /* fp16 weight images for long prompts: 0 per pass, 1 kept, -1 kept if they fit */
} kxm_config;

const char* kxm_last_error(void);
void* kxm_create(const kxm_config* cfg);
void kxm_destroy(void* model);
int kxm_init_synthetic(void* model, uint64_t seed);
int kxm_load_blob(void* model, const void* blob, size_t bytes);
/* q/k/v biases of one layer, host arrays of the unsharded sizes [dim], [kv_dim], [kv_dim] */
int kxm_set_qkv_bias(void* model, int32_t layer, const float* bq, const float* bk, const float* bv);
int kxm_read_weight(void* model, const char* name, int32_t layer, void* host_out, size_t bytes);
/* Tensor-parallel shard plan (pure host logic, callable without a GPU): {row0, rows, col0, cols}
 * for wq, wk, wv, wo, w1, w2, w3 in that order; -1 when the geometry does not divide. */
int kxm_shard_plan(int32_t dim, int32_t hidden_dim, int32_t head_num, int32_t kv_head_num,
                   int32_t tp_size, int32_t tp_rank, int32_t* slices28, int32_t* carries_residual);
int kxm_comm_unique_id(void* id128);
int kxm_init_comm(void* model, const void* id128);
/* peer mailboxes for the token loop's small collectives (LlamaModel::peer_create / peer_connect) */
int kxm_peer_create(void* model, uint64_t max_floats, void* handle64);
int kxm_peer_connect(void* model, const void* handles);
int kxm_peer_timed_out(void* model, int32_t* flag);
int kxm_peer_disable(void* model);
/* Row sub-blocks of the row-parallel prompt GEMMs under tensor parallelism: the all-reduce of one
 * sub-block runs on the side stream beside the GEMM of the next (1 = no overlap; default 4). */
int kxm_set_prefill_subblocks(void* model, int32_t n);
/* Policy of the fp16 weight images (TransformerConfig.prefill_weight_cache) changed on a live model:
 * 0 gives kept images back.  kxm_weight_image_bytes: device memory the kept images hold now. */
int kxm_set_prefill_weight_cache(void* model, int32_t policy);
int kxm_weight_image_bytes(void* model, uint64_t* bytes);
/* Test rig: tensor-parallel ranks as objects of one process (one host thread per rank), summing
 * through a host rendezvous instead of RCCL; eager launches only. */
void* kxm_loopback_create(int32_t ranks);
void kxm_loopback_destroy(void* group);
int kxm_join_loopback(void* model, void* group);
int kxm_forward_layers(void* model, int32_t token, int32_t pos, float* logits_host);
int kxm_set_state(void* model, int32_t token, int32_t pos);
int kxm_step_logits(void* model, int32_t token, int32_t pos, float* logits_host);
int kxm_decode(void* model, int32_t steps, int32_t use_graph);
/* exact = 0: batched MFMA prompt pass; exact = 1: one fp32 decode step per prompt row */
int kxm_prefill(void* model, const int32_t* tokens, int32_t n, int32_t pos0, float* logits_host,
                int32_t exact);
int kxm_read_history(void* model, int32_t* out, int32_t n);
int kxm_read_state(void* model, int32_t* token, int32_t* pos);
int kxm_read_kv(void* model, int32_t layer, int32_t rows, float* k_host, float* v_host);
int kxm_sync(void* model);
int kxm_profile_gemv(void* model, int32_t steps, int64_t* launches, double* bytes, double* ms);
/* Average milliseconds of one all-reduce of `count` floats on the token loop's own path (side
 * stream between two events), `iters` calls back to back.  A collective: every rank calls it. */
int kxm_profile_allreduce(void* model, uint64_t count, int32_t iters, double* ms);
double kxm_linear_bytes_per_token(void* model);
void* kxm_stream(void* model);

/* Layer-level checks used by the tests: run one op::MatmulLayer / op::MultiHeadAttention /
 * op::FlashAttention forward on DeviceType::kDeviceHIP from host arrays. */
int kxm_test_matmul_layer(const float* x, const void* w, const float* scales, int32_t group_size,
                          const float* bias, int32_t dim0, int32_t dim1, int32_t rows, float* y);
int kxm_test_mha_layer(const float* q, const float* kcache, const float* vcache, int32_t layers,
                       int32_t seq_len, int32_t head_num, int32_t kv_mul, int32_t head_size,
                       int32_t layer_idx, int32_t pos, float* out);
/* The same layer with NO reserved workspace and a garbage-filled score tensor (the state a
 * drop-in caller hands over), run twice in a row. */
int kxm_test_mha_layer_no_workspace(const float* q, const float* kcache, const float* vcache,
                                    int32_t layers, int32_t seq_len, int32_t head_num,
                                    int32_t kv_mul, int32_t head_size, int32_t layer_idx,
                                    int32_t pos, float* out);
/* 1 when a row-parallel MatmulLayer (tp_size 2) without a communicator refuses to run, 0 when it
 * returned success. */
int kxm_test_matmul_tp_without_comm(void);
int kxm_test_flash_layer(const float* q, const float* kcache, const float* vcache, int32_t layers,
                         int32_t seq_len, int32_t head_num, int32_t kv_mul, int32_t head_size,
                         int32_t layer_idx, int32_t pos, int32_t tq, float* out);
/* base::KVCacheManager layout probe: byte offsets of (layer) K and V pointers from the K base and
 * the arena size. */
int kxm_test_kv_layout(int32_t layers, int32_t seq_len, int32_t kv_head_num, int32_t head_size,
                       int32_t layer_idx, int64_t* k_off, int64_t* v_off, int64_t* total_bytes);
/* Registry contract: returns 1 when every getter hands back a kernel for kDeviceHIP. */
int kxm_test_registry(void);
/* One decode-attention call assembled from the registry's matmul / softmax / scale-sum kernels the
 * way the reference's CPU mha_kernel assembles it (cpu/mha_kernel.cpp:10-60); out receives
 * [head_num*head_size] results followed by the same values through the scale kernel (x 2). */
int kxm_test_mha_from_registry_pieces(const float* q, const float* kcache, const float* vcache,
                                      int32_t layers, int32_t seq_len, int32_t head_num,
                                      int32_t kv_mul, int32_t head_size, int32_t layer_idx,
                                      int32_t pos, float* out);

#ifdef __cplusplus
}
#endif
#endif
