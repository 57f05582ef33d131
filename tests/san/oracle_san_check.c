/* Sanitizer driver for the CPU oracle (built by tests/test_sanitizers.py with
 * -fsanitize=address,undefined together with oracle/kx_oracle.c): every oracle kernel on sizes that
 * end exactly at their buffers, and the three file parsers + token loop on the golden blobs given
 * on the command line as  <kind>:<path>  (kind f = legacy fp32, b = fp32 with q/k/v biases,
 * q = quant).  Heap buffers of exact size, so a stray read or write is an ASan report. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kx_oracle.h"

static float* fvec(size_t n, unsigned seed) {
  float* p = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) {
    seed = seed * 1664525u + 1013904223u;
    p[i] = ((float)(seed >> 8) / 8388608.0f) - 1.0f;
  }
  return p;
}

static int run_model(const char* kind, const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); return 1; }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* blob = malloc((size_t)n);  /* exact size: a parser that reads past the file is caught */
  if (fread(blob, 1, (size_t)n, f) != (size_t)n) return 1;
  fclose(f);
  kxo_weights w;
  int rc = kind[0] == 'q' ? kxo_parse_legacy_q8(blob, (size_t)n, &w)
         : kind[0] == 'b' ? kxo_parse_qwen2_f32(blob, (size_t)n, &w)
                          : kxo_parse_legacy_f32(blob, (size_t)n, &w);
  if (rc) { fprintf(stderr, "parse %s rc %d\n", path, rc); return 1; }
  w.cfg.rope_variant = kind[1] == 'h' ? KXO_ROPE_HALF_SPLIT : KXO_ROPE_INTERLEAVED;
  w.cfg.rope_theta = 10000.0f;
  w.cfg.rms_eps = 1e-5f;
  kxo_state* st = kxo_state_new(&w.cfg);
  int tok = 1;
  for (int pos = 0; pos < w.cfg.seq_len; ++pos) {  /* up to the LAST cache row */
    kxo_forward_token(&w, st, tok, pos, pos & 1 ? 2 : 1);
    tok = kxo_argmax(kxo_state_logits(st), w.cfg.vocab_size);
  }
  kxo_state_free(st);
  /* a truncated file must be refused, not read past */
  for (long cut = 1; cut < n; cut += n / 7 + 1) {
    void* part = malloc((size_t)cut);
    memcpy(part, blob, (size_t)cut);
    kxo_weights w2;
    if (kind[0] == 'q') (void)kxo_parse_legacy_q8(part, (size_t)cut, &w2);
    else if (kind[0] == 'b') (void)kxo_parse_qwen2_f32(part, (size_t)cut, &w2);
    else (void)kxo_parse_legacy_f32(part, (size_t)cut, &w2);
    free(part);
  }
  free(blob);
  printf("model %s ok\n", path);
  return 0;
}

int main(int argc, char** argv) {
  /* kernels */
  const int K = 37, M = 192, gs = 64;
  float* x = fvec(M, 1);
  float* wf = fvec((size_t)K * M, 2);
  int8_t* q = (int8_t*)malloc((size_t)K * M);
  float* s = (float*)malloc(sizeof(float) * K * M / gs);
  float* y = (float*)malloc(sizeof(float) * K);
  kxo_quantize_q80(wf, (size_t)K * M, gs, q, s);
  kxo_gemv_w8(x, q, s, gs, y, M, K, 1);
  kxo_gemv_w8(x, q, s, gs, y, M, K, 3);
  kxo_gemv_f32(x, wf, y, M, K, 0.5f, 2);
  float* X = fvec(3 * M, 3);
  float* Y = (float*)malloc(sizeof(float) * 3 * K);
  kxo_gemm_w8(X, q, s, gs, Y, 3, M, K, 2);
  kxo_gemm_f32(X, wf, Y, 3, M, K, 2);
  /* attention at the last row of a 2-layer cache */
  const int H = 6, KVH = 2, hs = 16, S = 9, kvd = KVH * hs;
  float* kc = fvec((size_t)2 * S * kvd, 4);
  float* vc = fvec((size_t)2 * S * kvd, 5);
  float* qq = fvec((size_t)4 * H * hs, 6);
  float* out = (float*)malloc(sizeof(float) * 4 * H * hs);
  float* score = (float*)malloc(sizeof(float) * H * S);
  kxo_mha(S - 1, H, 1, S, kvd, H / KVH, hs, out, qq, score, kc, vc);
  kxo_mha(0, H, 0, S, kvd, H / KVH, hs, out, qq, NULL, kc, vc);
  kxo_flash_attention(qq, kc, vc, out, 4, H, KVH, hs, S, S - 1, 1, 0.25f, 1, 2);
  kxo_flash_attention(qq, kc, vc, out, 4, H, KVH, hs, S, S - 1, 1, 0.25f, 0, 1);
  /* rope tables and both pairings at the last position */
  float* sn = (float*)malloc(sizeof(float) * S * hs);
  float* cs = (float*)malloc(sizeof(float) * S * hs);
  kxo_sin_cos_cache(hs, S, 10000.0f, sn, cs);
  float* rq = fvec((size_t)H * hs, 7);
  float* rk = fvec((size_t)kvd, 8);
  kxo_rope(KXO_ROPE_INTERLEAVED, H * hs, kvd, hs, rq, rk, S - 1, sn, cs);
  kxo_rope(KXO_ROPE_HALF_SPLIT, H * hs, kvd, hs, rq, rk, S - 1, sn, cs);
  kxo_rmsnorm(x, wf, y, K, 1e-5f);
  kxo_swiglu(x, wf, y, K);
  kxo_add(x, wf, y, K);
  int32_t toks[2] = {0, K - 1};
  float* e = (float*)malloc(sizeof(float) * 2 * M);
  kxo_embedding(toks, 2, wf, M, e);
  (void)kxo_argmax(y, K);
  free(x); free(wf); free(q); free(s); free(y); free(X); free(Y); free(kc); free(vc); free(qq);
  free(out); free(score); free(sn); free(cs); free(rq); free(rk); free(e);
  printf("kernels ok\n");
  int bad = 0;
  for (int i = 1; i < argc; ++i) {
    char* colon = strchr(argv[i], ':');
    if (!colon) continue;
    *colon = 0;
    bad += run_model(argv[i], colon + 1);
  }
  return bad;
}
