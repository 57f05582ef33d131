/*
 * kx_oracle.c -- CPU restatement of the reference's kernel arithmetic.
 * TEST INFRASTRUCTURE ONLY (see kx_oracle.h).  Plain C99; OpenMP is used only
 * when the caller passes threads > 1 (the all-cores CPU baseline).
 *
 * Parity pin: checked by tests/test_oracle_golden.py against (a) the known-answer
 * vectors of the reference's own tests (test/test_op/test_cu_matmul.cpp:10-106,
 * test/test_op/test_load.cpp:97-105) and (b) fixtures under tests/golden/ produced
 * by importing the reference's tools/model.py and tools/export_llama.py, and (c) the half-split
 * RoPE / GQA / bias fixtures g6, g7 produced by sending Hugging Face models through the
 * reference's tools/export_llama3.py and tools/export_qwen2.py (tests/golden/make_golden_hf.py).
 */
#include "kx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int clamp_threads(int32_t threads) {
#ifdef _OPENMP
  if (threads < 1) return 1;
  return threads;
#else
  (void)threads;
  return 1;
#endif
}

/* ---- quantiser: tools/export_llama.py:49-73 ---- */
void kxo_quantize_q80(const float* w, size_t n, int32_t group_size, int8_t* q, float* scales) {
  size_t groups = n / (size_t)group_size;
  for (size_t g = 0; g < groups; ++g) {
    const float* wg = w + g * (size_t)group_size;
    float wmax = 0.f;
    for (int32_t i = 0; i < group_size; ++i) {
      float a = fabsf(wg[i]);
      if (a > wmax) wmax = a;
    }
    float scale = wmax / 127.0f;
    scales[g] = scale;
    for (int32_t i = 0; i < group_size; ++i) {
      /* an all-zero group: 0/0 in the exporter; the int8 it stores for that is 0 */
      float quant = scale > 0.f ? wg[i] / scale : 0.f;
      /* torch.round: half to even == nearbyintf under the default rounding mode */
      q[g * (size_t)group_size + i] = (int8_t)nearbyintf(quant);
    }
  }
}

/* ---- W8 GEMV: cuda/matmul_kernel.cu:68-74 ---- */
static float dot_w8_row(const float* x, const int8_t* wrow, const float* scales, size_t row_base,
                        int32_t group_size, int32_t M) {
  double acc = 0.0;
  for (int32_t i = 0; i < M; ++i) {
    size_t weight_idx = row_base + (size_t)i;
    size_t group_idx = weight_idx / (size_t)group_size;
    float xs = x[i] * scales[group_idx];
    float term = xs * (float)wrow[i];
    acc += (double)term;
  }
  return (float)acc;
}

void kxo_gemv_w8(const float* x, const int8_t* w, const float* scales, int32_t group_size, float* y,
                 int32_t M, int32_t K, int32_t threads) {
  int nt = clamp_threads(threads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
  for (int32_t p = 0; p < K; ++p) {
    size_t row_base = (size_t)p * (size_t)M;
    y[p] = dot_w8_row(x, w + row_base, scales, row_base, group_size, M);
  }
}

/* ---- fp32 GEMV: cpu/matmul_kernel.cpp:37-40 ---- */
void kxo_gemv_f32(const float* x, const float* w, float* y, int32_t M, int32_t K, float scale,
                  int32_t threads) {
  int nt = clamp_threads(threads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
  for (int32_t p = 0; p < K; ++p) {
    const float* wrow = w + (size_t)p * (size_t)M;
    double acc = 0.0;
    for (int32_t i = 0; i < M; ++i) acc += (double)(x[i] * wrow[i]);
    y[p] = (float)acc * scale;
  }
}

void kxo_gemm_w8(const float* X, const int8_t* w, const float* scales, int32_t group_size, float* Y,
                 int32_t T, int32_t M, int32_t K, int32_t threads) {
  for (int32_t t = 0; t < T; ++t)
    kxo_gemv_w8(X + (size_t)t * M, w, scales, group_size, Y + (size_t)t * K, M, K, threads);
}

void kxo_gemm_f32(const float* X, const float* w, float* Y, int32_t T, int32_t M, int32_t K,
                  int32_t threads) {
  for (int32_t t = 0; t < T; ++t)
    kxo_gemv_f32(X + (size_t)t * M, w, Y + (size_t)t * K, M, K, 1.f, threads);
}

/* ---- attention for one (query row, head): cpu/mha_kernel.cpp:19-60 ---- */
static void attend_one_head(const float* q_head, const float* k_base, const float* v_base,
                            int32_t last, int32_t kv_dim, int32_t head_size, float scale,
                            float* out_head, float* score /* >= last+1 floats */) {
  /* scores: q . k[t] * scale  (mha_kernel.cpp:28-39 via matmul_kernel_cpu with scale) */
  for (int32_t t = 0; t <= last; ++t) {
    const float* k = k_base + (size_t)t * kv_dim;
    double acc = 0.0;
    for (int32_t d = 0; d < head_size; ++d) acc += (double)(q_head[d] * k[d]);
    score[t] = (float)acc * scale;
  }
  /* softmax_kernel.cpp:8-14 */
  float mx = score[0];
  for (int32_t t = 1; t <= last; ++t)
    if (score[t] > mx) mx = score[t];
  double sum = 0.0;
  for (int32_t t = 0; t <= last; ++t) {
    score[t] = expf(score[t] - mx);
    sum += (double)score[t];
  }
  float fsum = (float)sum;
  for (int32_t t = 0; t <= last; ++t) score[t] = score[t] / fsum;
  /* scale_sum_kernel.cpp:17-21: out = sum_t p[t] * v[t] */
  for (int32_t d = 0; d < head_size; ++d) {
    double acc = 0.0;
    for (int32_t t = 0; t <= last; ++t) acc += (double)(score[t] * v_base[(size_t)t * kv_dim + d]);
    out_head[d] = (float)acc;
  }
}

void kxo_mha(int32_t pos, int32_t head_num, int32_t layer_index, int32_t seq_len, int32_t kv_dim,
             int32_t kv_mul, int32_t head_size, float* mha_out, const float* query, float* score,
             const float* key_cache, const float* value_cache) {
  size_t layer_offset = (size_t)layer_index * (size_t)seq_len * (size_t)kv_dim; /* mha_kernel.cpp:10 */
  float scale = 1.f / sqrtf((float)head_size);                                   /* :11 */
  float* tmp = NULL;
  if (!score) tmp = (float*)malloc(sizeof(float) * (size_t)(pos + 1));
  for (int32_t h = 0; h < head_num; ++h) {
    size_t head_off = (size_t)(h / kv_mul) * (size_t)head_size; /* :29,:53 */
    float* sc = score ? score + (size_t)h * seq_len : tmp;
    attend_one_head(query + (size_t)h * head_size, key_cache + layer_offset + head_off,
                    value_cache + layer_offset + head_off, pos, kv_dim, head_size, scale,
                    mha_out + (size_t)h * head_size, sc);
  }
  free(tmp);
}

void kxo_flash_attention(const float* query, const float* key_cache, const float* value_cache,
                         float* out, int32_t Tq, int32_t head_num, int32_t kv_head_num,
                         int32_t head_size, int32_t seq_len, int32_t pos, int32_t layer_idx,
                         float softmax_scale, int32_t is_causal, int32_t threads) {
  int32_t kv_mul = head_num / kv_head_num;
  int32_t kv_dim = kv_head_num * head_size;
  size_t layer_offset = (size_t)layer_idx * (size_t)seq_len * (size_t)kv_dim;
  int nt = clamp_threads(threads);
  (void)nt;
  int64_t jobs = (int64_t)Tq * head_num;
#pragma omp parallel num_threads(nt) if (nt > 1)
  {
    float* sc = (float*)malloc(sizeof(float) * (size_t)(pos + 1));
#pragma omp for schedule(dynamic, 4)
    for (int64_t j = 0; j < jobs; ++j) {
      int32_t i = (int32_t)(j / head_num);
      int32_t h = (int32_t)(j % head_num);
      int32_t last = is_causal ? pos - (Tq - 1) + i : pos;
      size_t head_off = (size_t)(h / kv_mul) * (size_t)head_size;
      attend_one_head(query + ((size_t)i * head_num + h) * head_size,
                      key_cache + layer_offset + head_off, value_cache + layer_offset + head_off,
                      last, kv_dim, head_size, softmax_scale,
                      out + ((size_t)i * head_num + h) * head_size, sc);
    }
    free(sc);
  }
}

/* ---- rmsnorm: cpu/rmsnorm_kernel.cpp:30-32 ---- */
void kxo_rmsnorm(const float* in, const float* weight, float* out, int32_t dim, float eps) {
  double ss = 0.0;
  for (int32_t i = 0; i < dim; ++i) ss += (double)(in[i] * in[i]);
  float mean = (float)(ss / (double)dim) + eps;
  float rs = 1.f / sqrtf(mean);
  for (int32_t i = 0; i < dim; ++i) out[i] = weight[i] * (rs * in[i]);
}

/* ---- rope tables: cpu/rope_kernel.cpp:84-96 (theta 1e4), :4-16 (5e5), :44-56 (1e6) ---- */
void kxo_sin_cos_cache(int32_t head_size, int32_t max_seq_len, float theta, float* sin_cache,
                       float* cos_cache) {
  for (int32_t pos = 0; pos < max_seq_len; ++pos) {
    for (int32_t hd = 0; hd < head_size; ++hd) {
      float freq = 1.0f / powf(theta, (float)hd / (float)head_size);
      float val = (float)pos * freq;
      sin_cache[(size_t)pos * head_size + hd] = sinf(val);
      cos_cache[(size_t)pos * head_size + hd] = cosf(val);
    }
  }
}

void kxo_rope(int32_t variant, int32_t dim, int32_t kv_dim, int32_t head_size, float* q, float* k,
              int32_t pos, const float* sin_cache, const float* cos_cache) {
  if (variant == KXO_ROPE_INTERLEAVED) {
    /* rope_kernel.cpp:105-120 */
    for (int32_t i = 0; i < dim; i += 2) {
      int32_t hd = i % head_size;
      float fci = sin_cache[(size_t)pos * head_size + hd];
      float fcr = cos_cache[(size_t)pos * head_size + hd];
      int32_t rotn = i < kv_dim ? 2 : 1;
      for (int32_t v = 0; v < rotn; ++v) {
        float* vec = v == 0 ? q : k;
        float v0 = vec[i], v1 = vec[i + 1];
        vec[i] = v0 * fcr - v1 * fci;
        vec[i + 1] = v0 * fci + v1 * fcr;
      }
    }
  } else {
    /* rope_kernel.cpp:25-41 */
    int32_t half = head_size / 2;
    for (int32_t i = 0; i < dim; i += head_size) {
      for (int32_t hd = 0; hd < half; ++hd) {
        float fci = sin_cache[(size_t)pos * head_size + hd * 2];
        float fcr = cos_cache[(size_t)pos * head_size + hd * 2];
        int32_t rotn = i < kv_dim ? 2 : 1;
        for (int32_t v = 0; v < rotn; ++v) {
          float* vec = v == 0 ? q : k;
          float v0 = vec[i + hd], v1 = vec[i + hd + half];
          vec[i + hd] = v0 * fcr - v1 * fci;
          vec[i + hd + half] = v0 * fci + v1 * fcr;
        }
      }
    }
  }
}

void kxo_swiglu(const float* in1, const float* in2, float* out, int32_t n) {
  for (int32_t i = 0; i < n; ++i) {
    float g = in1[i] * (1.0f / (1.0f + expf(-in1[i])));
    out[i] = g * in2[i];
  }
}

void kxo_add(const float* a, const float* b, float* out, int32_t n) {
  for (int32_t i = 0; i < n; ++i) out[i] = a[i] + b[i];
}

void kxo_embedding(const int32_t* tokens, int32_t n_tokens, const float* table, int32_t dim,
                   float* out) {
  for (int32_t i = 0; i < n_tokens; ++i)
    memcpy(out + (size_t)i * dim, table + (size_t)tokens[i] * dim, sizeof(float) * (size_t)dim);
}

int32_t kxo_argmax(const float* logits, int32_t n) {
  int32_t best = 0;
  for (int32_t i = 1; i < n; ++i)
    if (logits[i] > logits[best]) best = i;
  return best;
}

/* ---- token graph ---- */
struct kxo_state {
  kxo_config cfg;
  int32_t kv_dim, head_size, kv_mul;
  float *x, *rms, *q, *mha, *attn, *h1, *h3, *w2o, *logits, *score;
  float *kcache, *vcache, *sin_cache, *cos_cache;
};

static float* zalloc_f(size_t n) { return (float*)calloc(n ? n : 1, sizeof(float)); }

kxo_state* kxo_state_new(const kxo_config* cfg) {
  kxo_state* st = (kxo_state*)calloc(1, sizeof(kxo_state));
  st->cfg = *cfg;
  st->head_size = cfg->dim / cfg->head_num;                    /* model.cpp:134 */
  st->kv_dim = (cfg->dim * cfg->kv_head_num) / cfg->head_num;  /* model.cpp:132 */
  st->kv_mul = cfg->head_num / cfg->kv_head_num;               /* model.cpp:133 */
  size_t cache = (size_t)cfg->layer_num * cfg->seq_len * st->kv_dim;
  st->x = zalloc_f(cfg->dim);
  st->rms = zalloc_f(cfg->dim);
  st->q = zalloc_f(cfg->dim);
  st->mha = zalloc_f(cfg->dim);
  st->attn = zalloc_f(cfg->dim);
  st->h1 = zalloc_f(cfg->hidden_dim);
  st->h3 = zalloc_f(cfg->hidden_dim);
  st->w2o = zalloc_f(cfg->dim);
  st->logits = zalloc_f(cfg->vocab_size);
  st->score = zalloc_f((size_t)cfg->head_num * cfg->seq_len);
  st->kcache = zalloc_f(cache);
  st->vcache = zalloc_f(cache);
  st->sin_cache = zalloc_f((size_t)cfg->seq_len * st->head_size);
  st->cos_cache = zalloc_f((size_t)cfg->seq_len * st->head_size);
  kxo_sin_cos_cache(st->head_size, cfg->seq_len, cfg->rope_theta, st->sin_cache, st->cos_cache);
  return st;
}

void kxo_state_free(kxo_state* st) {
  if (!st) return;
  free(st->x); free(st->rms); free(st->q); free(st->mha); free(st->attn); free(st->h1);
  free(st->h3); free(st->w2o); free(st->logits); free(st->score); free(st->kcache);
  free(st->vcache); free(st->sin_cache); free(st->cos_cache);
  free(st);
}

float* kxo_state_key_cache(kxo_state* st) { return st->kcache; }
float* kxo_state_value_cache(kxo_state* st) { return st->vcache; }
float* kxo_state_logits(kxo_state* st) { return st->logits; }

static void linear(const kxo_config* c, const float* wf, const int8_t* wq, const float* ws,
                   size_t layer, int32_t K, int32_t M, const float* x, float* y, int32_t threads) {
  size_t off = layer * (size_t)K * (size_t)M;
  if (c->group_size > 0) {
    /* each matrix was quantised on its own: group index restarts per matrix (llama3.cpp:211-216) */
    kxo_gemv_w8(x, wq + off, ws + off / (size_t)c->group_size, c->group_size, y, M, K, threads);
  } else {
    kxo_gemv_f32(x, wf + off, y, M, K, 1.f, threads);
  }
}

void kxo_forward_token(const kxo_weights* w, kxo_state* st, int32_t token, int32_t pos,
                       int32_t threads) {
  const kxo_config* c = &w->cfg;
  int32_t dim = c->dim, hid = c->hidden_dim, kvd = st->kv_dim;
  kxo_embedding(&token, 1, w->emb, dim, st->x); /* llama3.cpp:615-635, model.cpp:268-293 */
  for (int32_t l = 0; l < c->layer_num; ++l) {
    /* attention_rms llama3.cpp:637-646 */
    kxo_rmsnorm(st->x, w->attn_norm + (size_t)l * dim, st->rms, dim, c->rms_eps);
    /* attention_qkv :648-677 ; K/V rows written in place in the cache (model.cpp:251-262) */
    float* krow = st->kcache + ((size_t)l * c->seq_len + pos) * kvd;
    float* vrow = st->vcache + ((size_t)l * c->seq_len + pos) * kvd;
    linear(c, w->wq_f, w->wq_q, w->wq_s, l, dim, dim, st->rms, st->q, threads);
    linear(c, w->wk_f, w->wk_q, w->wk_s, l, kvd, dim, st->rms, krow, threads);
    linear(c, w->wv_f, w->wv_q, w->wv_s, l, kvd, dim, st->rms, vrow, threads);
    /* bias after the matmul, before RoPE (matmul.cpp:76-79; qwen2.cpp attention_qkv) */
    if (w->bq) kxo_add(st->q, w->bq + (size_t)l * dim, st->q, dim);
    if (w->bk) kxo_add(krow, w->bk + (size_t)l * kvd, krow, kvd);
    if (w->bv) kxo_add(vrow, w->bv + (size_t)l * kvd, vrow, kvd);
    kxo_rope(c->rope_variant, dim, kvd, st->head_size, st->q, krow, pos, st->sin_cache,
             st->cos_cache);
    /* attention_mha :716-743 */
    kxo_mha(pos, c->head_num, l, c->seq_len, kvd, st->kv_mul, st->head_size, st->mha, st->q,
            st->score, st->kcache, st->vcache);
    linear(c, w->wo_f, w->wo_q, w->wo_s, l, dim, dim, st->mha, st->attn, threads);
    /* feed_forward :746-788 */
    kxo_add(st->x, st->attn, st->x, dim);
    kxo_rmsnorm(st->x, w->ffn_norm + (size_t)l * dim, st->rms, dim, c->rms_eps);
    linear(c, w->w1_f, w->w1_q, w->w1_s, l, hid, dim, st->rms, st->h1, threads);
    linear(c, w->w3_f, w->w3_q, w->w3_s, l, hid, dim, st->rms, st->h3, threads);
    kxo_swiglu(st->h1, st->h3, st->h1, hid);
    linear(c, w->w2_f, w->w2_q, w->w2_s, l, dim, hid, st->h1, st->w2o, threads);
    kxo_add(st->x, st->w2o, st->x, dim);
  }
  /* cls_logits :790-799 */
  kxo_rmsnorm(st->x, w->final_norm, st->x, dim, c->rms_eps);
  linear(c, w->cls_f, w->cls_q, w->cls_s, 0, c->vocab_size, dim, st->x, st->logits, threads);
}

/* ---- on-disk layouts ---- */
static int read_header(const int32_t* h, kxo_config* c, int* shared) {
  c->dim = h[0]; c->hidden_dim = h[1]; c->layer_num = h[2]; c->head_num = h[3];
  c->kv_head_num = h[4];
  *shared = h[5] > 0;              /* model.cpp:138-142 */
  c->vocab_size = abs(h[5]);       /* model.cpp:150 */
  c->seq_len = h[6];
  if (c->dim <= 0 || c->head_num <= 0 || c->kv_head_num <= 0 || c->layer_num <= 0) return -1;
  return 0;
}

int kxo_parse_legacy_f32(const void* blob, size_t bytes, kxo_weights* out) {
  memset(out, 0, sizeof(*out));
  if (bytes < 28) return -1;
  int shared = 0;
  kxo_config* c = &out->cfg;
  if (read_header((const int32_t*)blob, c, &shared)) return -1;
  c->group_size = 0;
  size_t dim = c->dim, hid = c->hidden_dim, L = c->layer_num, V = c->vocab_size;
  size_t hs = dim / c->head_num, kvd = hs * c->kv_head_num;
  const float* p = (const float*)((const char*)blob + 28); /* model.cpp:110-111 */
  /* order: tools/export_llama.py:96-127 ; offsets mirrored at llama3.cpp:315-440 */
  out->emb = p;            p += V * dim;
  out->attn_norm = p;      p += L * dim;
  out->wq_f = p;           p += L * dim * dim;
  out->wk_f = p;           p += L * kvd * dim;
  out->wv_f = p;           p += L * kvd * dim;
  out->wo_f = p;           p += L * dim * dim;
  out->ffn_norm = p;       p += L * dim;
  out->w1_f = p;           p += L * hid * dim;
  out->w2_f = p;           p += L * dim * hid;
  out->w3_f = p;           p += L * hid * dim;
  out->final_norm = p;     p += dim;
  p += (size_t)c->seq_len * hs; /* freqs_cos + freqs_sin, skipped: llama3.cpp:384-385 */
  out->cls_f = shared ? out->emb : p;
  if (!shared) p += V * dim;
  if ((size_t)((const char*)p - (const char*)blob) > bytes) return -2;
  return 0;
}

int kxo_parse_qwen2_f32(const void* blob, size_t bytes, kxo_weights* out) {
  memset(out, 0, sizeof(*out));
  if (bytes < 28) return -1;
  int shared = 0;
  kxo_config* c = &out->cfg;
  if (read_header((const int32_t*)blob, c, &shared)) return -1;
  c->group_size = 0;
  size_t dim = c->dim, hid = c->hidden_dim, L = c->layer_num, V = c->vocab_size;
  size_t hs = dim / c->head_num, kvd = hs * c->kv_head_num;
  const float* p = (const float*)((const char*)blob + 28);
  const float* end = (const float*)((const char*)blob + bytes);
  /* order: tools/export_qwen2.py:96-130; offsets mirrored at qwen2.cpp:298-385 */
  out->emb = p;            p += V * dim;
  out->attn_norm = p;      p += L * dim;
  size_t rows[3] = {dim, kvd, kvd};
  float* wdst[3];
  float* bdst[3];
  for (int m = 0; m < 3; ++m) {
    wdst[m] = (float*)malloc(L * rows[m] * dim * sizeof(float));
    bdst[m] = (float*)malloc(L * rows[m] * sizeof(float));
    for (size_t l = 0; l < L; ++l) {
      if (p + rows[m] * dim + rows[m] > end) return -2;
      memcpy(wdst[m] + l * rows[m] * dim, p, rows[m] * dim * sizeof(float));
      p += rows[m] * dim;
      memcpy(bdst[m] + l * rows[m], p, rows[m] * sizeof(float));
      p += rows[m];
    }
  }
  out->wq_f = wdst[0]; out->wk_f = wdst[1]; out->wv_f = wdst[2];
  out->bq = bdst[0];   out->bk = bdst[1];   out->bv = bdst[2];
  out->wo_f = p;           p += L * dim * dim;
  out->ffn_norm = p;       p += L * dim;
  out->w1_f = p;           p += L * hid * dim;
  out->w2_f = p;           p += L * dim * hid;
  out->w3_f = p;           p += L * hid * dim;
  out->final_norm = p;     p += dim;
  p += (size_t)c->seq_len * hs; /* freqs_cos + freqs_sin, skipped: qwen2.cpp:373-374 */
  out->cls_f = shared ? out->emb : p;
  if (!shared) p += V * dim;
  if (p > end) return -2;
  return 0;
}

int kxo_parse_legacy_q8(const void* blob, size_t bytes, kxo_weights* out) {
  memset(out, 0, sizeof(*out));
  if (bytes < 32) return -1;
  int shared = 0;
  kxo_config* c = &out->cfg;
  const int32_t* h = (const int32_t*)blob;
  if (read_header(h, c, &shared)) return -1;
  c->group_size = h[7]; /* model.cpp:74-80 */
  if (c->group_size <= 0) return -1;
  /* A tied classifier has no int8 copy in the file (export_llama.py:183-192); the reference then
   * points the int8 layer at the fp32 embedding bytes (llama3.cpp:273-286).  Here the classifier
   * is what the exporter would have written for an untied copy of the table: quantize_q80 of the
   * fp32 embedding, below. */
  size_t dim = c->dim, hid = c->hidden_dim, L = c->layer_num, V = c->vocab_size;
  size_t hs = dim / c->head_num, kvd = hs * c->kv_head_num, gs = c->group_size;
  const char* p = (const char*)blob + 32; /* model.cpp:113-114 */
  /* each matrix: int8 block immediately followed by its fp32 scales (op/layer.cpp:209-214);
   * all layers of one kind are consecutive (llama3.cpp:210-271).  The file interleaves
   * [q(layer0) s(layer0) q(layer1) s(layer1) ...]; the oracle wants layer-major q and
   * layer-major s, so de-interleave into owned storage (never freed: test-sized models). */
  size_t shapes[7][2] = {{dim, dim}, {kvd, dim}, {kvd, dim}, {dim, dim},
                         {hid, dim}, {dim, hid}, {hid, dim}};
  int8_t* qdst[7];
  float* sdst[7];
  for (int m = 0; m < 7; ++m) {
    size_t n = shapes[m][0] * shapes[m][1];
    if (n % gs) return -4;
    qdst[m] = (int8_t*)malloc(L * n);
    sdst[m] = (float*)malloc(L * (n / gs) * sizeof(float));
    for (size_t l = 0; l < L; ++l) {
      if ((size_t)(p - (const char*)blob) + n + (n / gs) * 4 > bytes) return -2;
      memcpy(qdst[m] + l * n, p, n);
      memcpy(sdst[m] + l * (n / gs), p + n, (n / gs) * sizeof(float));
      p += n + (n / gs) * sizeof(float);
    }
  }
  out->wq_q = qdst[0]; out->wq_s = sdst[0];
  out->wk_q = qdst[1]; out->wk_s = sdst[1];
  out->wv_q = qdst[2]; out->wv_s = sdst[2];
  out->wo_q = qdst[3]; out->wo_s = sdst[3];
  out->w1_q = qdst[4]; out->w1_s = sdst[4];
  out->w2_q = qdst[5]; out->w2_s = sdst[5];
  out->w3_q = qdst[6]; out->w3_s = sdst[6];
  int8_t* cq = NULL;
  float* cs = NULL;
  {
    size_t n = V * dim;
    if (n % gs) return -4;
    cq = (int8_t*)malloc(n);
    cs = (float*)malloc((n / gs) * sizeof(float));
    if (!shared) {
      if ((size_t)(p - (const char*)blob) + n + (n / gs) * 4 > bytes) return -2;
      memcpy(cq, p, n);
      memcpy(cs, p + n, (n / gs) * sizeof(float));
      p += n + (n / gs) * sizeof(float);
    }
    out->cls_q = cq; out->cls_s = cs;
  }
  /* fp32 tail: embedding, attention norms, ffn norms, final norm (export_llama.py:191-203;
   * llama3.cpp:288-304).  May be unaligned in the blob: copy out. */
  size_t tail = V * dim + 2 * L * dim + dim;
  if ((size_t)(p - (const char*)blob) + tail * 4 > bytes) return -2;
  float* t = (float*)malloc(tail * sizeof(float));
  memcpy(t, p, tail * sizeof(float));
  out->emb = t;
  out->attn_norm = t + V * dim;
  out->ffn_norm = out->attn_norm + L * dim;
  out->final_norm = out->ffn_norm + L * dim;
  if (shared) kxo_quantize_q80(out->emb, V * dim, (int32_t)gs, cq, cs);
  return 0;
}
