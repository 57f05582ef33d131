// Support kernels of the token path: RMSNorm, RoPE (+ table), SwiGLU, add, embedding gather,
// greedy argmax.  Each is O(dim) bytes per token next to O(dim^2) weight bytes, so they are kept
// simple, vectorised and correct; the reference twins are cited per kernel.
#define KX_FILE_ID 7
#include "kx_common.hpp"
#include "kx_step.hpp"

namespace kx {

// Block-wide sum for blocks of up to 1024 threads; red must hold 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  __syncthreads();
  return t;
}

// The same normalisation with the result rounded to fp16 (activations of the prefill GEMMs).
__global__ void rmsnorm_f16out_kernel(const float* __restrict__ in, const float* __restrict__ weight,
                                      _Float16* __restrict__ out, int dim, float eps KX_BPARAM(3)) {
  __shared__ float red[16];
  const f32x4* x4 = reinterpret_cast<const f32x4*>(in + (size_t)blockIdx.x * dim);
  const f32x4* w4 = reinterpret_cast<const f32x4*>(weight);
  f16x4* o4 = reinterpret_cast<f16x4*>(out + (size_t)blockIdx.x * dim);
  float ss = 0.f;
  for (int i = threadIdx.x; i < dim / 4; i += blockDim.x) {
    const f32x4 v = *KX_CKP(0, x4 + i);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = block_sum(ss, red);
  const float rs = 1.0f / sqrtf(ss / (float)dim + eps);
  for (int i = threadIdx.x; i < dim / 4; i += blockDim.x) {
    const f32x4 v = *KX_CKP(0, x4 + i), w = *KX_CKP(1, w4 + i);
    f16x4 h;
    h[0] = sat_f16(w.x * (rs * v.x)); h[1] = sat_f16(w.y * (rs * v.y));
    h[2] = sat_f16(w.z * (rs * v.z)); h[3] = sat_f16(w.w * (rs * v.w));
    *KX_CKP(2, o4 + i) = h;
  }
}

// row_rmsnorm_f32 (kuiper/source/op/kernels/cuda/rmsnorm_kernel.cu:62-108):
// out = weight * (x * rsqrt(mean(x^2) + eps)); one workgroup per row.
__global__ void rmsnorm_kernel(const float* __restrict__ in, const float* __restrict__ weight,
                               float* __restrict__ out, int dim, float eps KX_BPARAM(3)) {
  __shared__ float red[16];
  const float* x = in + (size_t)blockIdx.x * dim;
  float* o = out + (size_t)blockIdx.x * dim;
  float ss = 0.f;
  const bool vec = (dim % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)o % 16 == 0) &&
                   ((uintptr_t)weight % 16 == 0);
  if (vec) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (int i = threadIdx.x; i < dim / 4; i += blockDim.x) {
      const f32x4 v = *KX_CKP(0, x4 + i);
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  } else {
    for (int i = threadIdx.x; i < dim; i += blockDim.x) ss += *KX_CKP(0, x + i) * x[i];
  }
  ss = block_sum(ss, red);
  const float rs = 1.0f / sqrtf(ss / (float)dim + eps);
  if (vec) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(weight);
    f32x4* o4 = reinterpret_cast<f32x4*>(o);
    for (int i = threadIdx.x; i < dim / 4; i += blockDim.x) {
      const f32x4 v = *KX_CKP(0, x4 + i), w = *KX_CKP(1, w4 + i);
      *KX_CKP(2, o4 + i) = f32x4{w.x * (rs * v.x), w.y * (rs * v.y), w.z * (rs * v.z), w.w * (rs * v.w)};
    }
  } else {
    for (int i = threadIdx.x; i < dim; i += blockDim.x) *KX_CKP(2, o + i) = *KX_CKP(1, weight + i) * (rs * x[i]);
  }
}

// sin_cos_calc (cuda/rope_kernel.cu:38-49): table[pos][head_dim], freq = theta^(-head_dim/hs).
__global__ void sin_cos_kernel(int head_size, int max_seq_len, float theta, float* sin_cache,
                               float* cos_cache KX_BPARAM(2)) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= head_size * max_seq_len) return;
  const int pos = idx / head_size, hd = idx % head_size;
  const float freq = 1.0f / powf(theta, (float)hd / (float)head_size);
  const float val = (float)pos * freq;
  *KX_CKP(0, sin_cache + idx) = sinf(val);
  *KX_CKP(1, cos_cache + idx) = cosf(val);
}

// rope_kernel_cu_fp32 (cuda/rope_kernel.cu:5-36 half-split, :98-128 interleaved).
// One thread per rotation pair of q; pairs inside kv_dim also rotate k.
// Optional q/k/v biases (added ahead of the rotation, as MatmulLayer adds them ahead of the RoPE
// layer: matmul.cpp:76-79, qwen2.cpp attention_qkv); v is touched only to receive its bias.
struct RopeBias {
  const float* bq;
  const float* bk;
  const float* bv;
  float* v;
  int v_stride;
};

template <int VARIANT>
__global__ void rope_kernel(int dim, int kv_dim, int head_size, float* q, float* k, int k_stride,
                            int pos, int rows, const float* __restrict__ sin_cache,
                            const float* __restrict__ cos_cache, const RopeBias b KX_BPARAM(8)) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;
  if (pair >= dim / 2 || row >= rows) return;
  const int p = pos + row;
  int i0, i1, tab;
  if (VARIANT == KX_ROPE_INTERLEAVED) {
    i0 = pair * 2;
    i1 = i0 + 1;
    tab = i0 % head_size;
  } else {
    const int half = head_size / 2;
    const int head = pair / half, hd = pair % half;
    i0 = head * head_size + hd;
    i1 = i0 + half;
    tab = hd * 2;
  }
  const float fci = *KX_CKP(2, sin_cache + ((size_t)p * head_size + tab));
  const float fcr = *KX_CKP(3, cos_cache + ((size_t)p * head_size + tab));
  float* qr = q + (size_t)row * dim;
  float q0 = *KX_CKP(0, qr + i0), q1 = *KX_CKP(0, qr + i1);
  if (b.bq) {
    q0 += *KX_CKP(4, b.bq + i0);
    q1 += *KX_CKP(4, b.bq + i1);
  }
  qr[i0] = q0 * fcr - q1 * fci;
  qr[i1] = q0 * fci + q1 * fcr;
  const int head_base = VARIANT == KX_ROPE_INTERLEAVED ? i0 : (i0 / head_size) * head_size;
  if (head_base < kv_dim) {
    float* kr = k + (size_t)row * k_stride;
    float k0 = *KX_CKP(1, kr + i0), k1 = *KX_CKP(1, kr + i1);
    if (b.bk) {
      k0 += *KX_CKP(5, b.bk + i0);
      k1 += *KX_CKP(5, b.bk + i1);
    }
    kr[i0] = k0 * fcr - k1 * fci;
    kr[i1] = k0 * fci + k1 * fcr;
    if (b.bv) {
      float* vr = b.v + (size_t)row * b.v_stride;
      *KX_CKP(7, vr + i0) += *KX_CKP(6, b.bv + i0);
      *KX_CKP(7, vr + i1) += *KX_CKP(6, b.bv + i1);
    }
  }
}

// swiglu_kernel_cu_fp32 (cuda/swiglu_kernel.cu:4-22)
__global__ void swiglu_kernel(const float* __restrict__ a, const float* __restrict__ b,
                              float* __restrict__ out, int n KX_BPARAM(3)) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = *KX_CKP(0, a + i);
  *KX_CKP(2, out + i) = g * (1.0f / (1.0f + expf(-g))) * *KX_CKP(1, b + i);
}

// Row-batched SwiGLU over a stacked [gate | up] activation: out[t][h] = silu(in[t][h]) *
// in[t][half + h] (the prefill form of swiglu_kernel_cu_fp32).
template <typename OutT>
__global__ void swiglu_rows_kernel(const float* __restrict__ in, int64_t ld_in, int half,
                                   OutT* __restrict__ out, int64_t ld_out, int rows KX_BPARAM(2)) {
  const int64_t n4 = (int64_t)rows * (half / 4);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / (half / 4);
    const int h = (int)(i - t * (half / 4)) * 4;
    const f32x4 g = *KX_CKP(0, reinterpret_cast<const f32x4*>(in + t * ld_in + h));
    const f32x4 u = *KX_CKP(0, reinterpret_cast<const f32x4*>(in + t * ld_in + half + h));
    f32x4 o;
    o.x = g.x * (1.0f / (1.0f + expf(-g.x))) * u.x;
    o.y = g.y * (1.0f / (1.0f + expf(-g.y))) * u.y;
    o.z = g.z * (1.0f / (1.0f + expf(-g.z))) * u.z;
    o.w = g.w * (1.0f / (1.0f + expf(-g.w))) * u.w;
    if constexpr (sizeof(OutT) == 4) {
      *KX_CKP(1, reinterpret_cast<f32x4*>(out + t * ld_out + h)) = o;
    } else {
      f16x4 o16;
      o16[0] = sat_f16(o.x); o16[1] = sat_f16(o.y); o16[2] = sat_f16(o.z); o16[3] = sat_f16(o.w);
      *KX_CKP(1, reinterpret_cast<f16x4*>(out + t * ld_out + h)) = o16;
    }
  }
}

// add_kernel_cu_fp32 (cuda/add_kernel.cu:4-12)
__global__ void add_kernel(const float* a, const float* b, float* out, int n KX_BPARAM(3)) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) *KX_CKP(2, out + i) = *KX_CKP(0, a + i) + *KX_CKP(1, b + i);
}

// emb_kernel_cu_fp32 (cuda/emb_kernel.cu:3-21): one workgroup per token row.
__global__ void embedding_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ table,
                                 float* __restrict__ out, int dim, int vocab KX_BPARAM(3)) {
  const int t = blockIdx.x;
  int tok = *KX_CKP(0, tokens + t);
  if (tok < 0 || tok >= vocab) tok = 0;  // the reference aborts on an out-of-range id
  const float* src = table + (size_t)tok * dim;
  float* dst = out + (size_t)t * dim;
  if (dim % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0)) {
    for (int i = threadIdx.x; i < dim / 4; i += blockDim.x)
      *KX_CKP(2, reinterpret_cast<f32x4*>(dst) + i) = *KX_CKP(1, reinterpret_cast<const f32x4*>(src) + i);
  } else {
    for (int i = threadIdx.x; i < dim; i += blockDim.x) *KX_CKP(2, dst + i) = *KX_CKP(1, src + i);
  }
}

// Same gather for one decode step with everything on the device: the token is history[*pos].
__global__ void embedding_dpos_kernel(const int32_t* __restrict__ history,
                                      const int32_t* __restrict__ pos,
                                      const float* __restrict__ table, float* __restrict__ out,
                                      int dim, int vocab KX_BPARAM(4)) {
  int tok = *KX_CKP(0, history + *KX_CKP(1, pos));
  if (tok < 0 || tok >= vocab) tok = 0;
  const float* src = table + (size_t)tok * dim;
  if (dim % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim / 4; i += gridDim.x * blockDim.x)
      *KX_CKP(3, reinterpret_cast<f32x4*>(out) + i) = *KX_CKP(2, reinterpret_cast<const f32x4*>(src) + i);
  } else {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x)
      *KX_CKP(3, out + i) = *KX_CKP(2, src + i);
  }
}

// argmax_kernel_fp32 (cuda/argmax_kernel.cu:49-71): first maximum wins.  Single workgroup.
__global__ void argmax_kernel(const float* __restrict__ x, int n, int32_t* out KX_BPARAM(2)) {
  __shared__ float sv[16];
  __shared__ int si[16];
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = *KX_CKP(0, x + i);
    if (v > best) {  // strictly greater: the lowest index of a thread's stripe wins ties
      best = v;
      bi = i;
    }
  }
  if (bi == 0x7fffffff && threadIdx.x < n) bi = threadIdx.x;  // all -inf / NaN stripe
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    argmax_merge(best, bi, ov, oi);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) {
    sv[wave] = best;
    si[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w) argmax_merge(best, bi, sv[w], si[w]);
    *KX_CKP(1, out) = bi == 0x7fffffff ? 0 : bi;
  }
}

// Greedy sampling + bookkeeping of one decode step, all on the device so that a captured token
// graph can be replayed without the host in the loop: next = argmax(logits) (first maximum
// wins, argmax_sampler.cpp:6-8); history[pos+1] = next; token = next; pos += 1.
// Stage 1: kStepBlocks workgroups each reduce a contiguous slice to one (value, index) pair.
// Stage 2: one workgroup reduces the pairs and does the bookkeeping.
__global__ __launch_bounds__(256) void step_pick_kernel(const StepArgs a) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int sm_last, sm_tok;
  const int per = (a.n + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(lo + per, a.n);
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float v = *KX_CK(a, SB_LOGITS, a.logits + i);
    if (v > best) {  // strictly greater: the lowest index of a thread's stripe wins ties
      best = v;
      bi = i;
    }
  }
  block_argmax(best, bi, sv, si);
  if (gridDim.x > 1) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(KX_CK(a, SB_PV, a.pv + blockIdx.x), best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(KX_CK(a, SB_PI, a.pi + blockIdx.x), bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // both stores acknowledged before the ticket
      sm_last = atomicInc(KX_CK(a, SB_TICKET, a.ticket), gridDim.x - 1u) == gridDim.x - 1u;
    }
    __syncthreads();
    if (!sm_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler order only: the loads below are device-scope
    best = -INFINITY;
    bi = 0x7fffffff;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x)
      argmax_merge(best, bi, __hip_atomic_load(KX_CK(a, SB_PV, a.pv + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   __hip_atomic_load(KX_CK(a, SB_PI, a.pi + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();  // sv / si are reused
    block_argmax(best, bi, sv, si);
  }
  if (a.cand) {
    if (threadIdx.x == 0) {
      *KX_CK(a, SB_CAND, a.cand) = best;
      *KX_CK(a, SB_CAND, a.cand + 1) = bi == 0x7fffffff ? -1.0f : (float)(a.index_base + bi);
    }
    return;
  }
  if (threadIdx.x == 0) step_bookkeeping(a, bi, &sm_tok);
  __syncthreads();
  step_embed(a, sm_tok);
}

// Tensor-parallel greedy pick over a row-split classifier: every rank reduces ITS slice of the
// logits to one candidate {value, global index (exact in fp32: < 2^24)} (step_pick_kernel with
// `cand`), the candidates are all-gathered, and every rank picks the same winner (largest value,
// lowest index on ties: the first maximum of the unsharded vector, argmax_sampler.cpp:6-8).
__global__ __launch_bounds__(256) void step_end_candidates_kernel(const float* __restrict__ cands,
                                                                  int ranks, const StepArgs a KX_BPARAM(1)) {
  __shared__ int sm_tok;
  if (threadIdx.x == 0) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int r = 0; r < ranks; ++r) {
      const float idx = *KX_CKP(0, cands + (2 * r + 1));
      if (idx >= 0.f) argmax_merge(best, bi, *KX_CKP(0, cands + 2 * r), (int)idx);
    }
    step_bookkeeping(a, bi, &sm_tok);
  }
  __syncthreads();
  step_embed(a, sm_tok);
}

// ---- the pieces the reference's CPU attention is assembled from (cpu/mha_kernel.cpp:19-60) ----
// scale_inplace_cpu (cpu/scale_kernel.cpp:3-8): x *= scale.
__global__ void scale_kernel(float scale, float* x, int n KX_BPARAM(1)) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) *KX_CKP(0, x + i) *= scale;
}

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  __syncthreads();
  return t;
}

// softmax_inplace_cpu (cpu/softmax_kernel.cpp:4-15): x = exp(x - max(x)) / sum; one workgroup.
__global__ void softmax_kernel(float* x, int n KX_BPARAM(1)) {
  __shared__ float red[16];
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, *KX_CKP(0, x + i));
  mx = block_max(mx, red);
  float sum = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float e = expf(x[i] - mx);
    x[i] = e;
    sum += e;
  }
  sum = block_sum(sum, red);
  for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = x[i] / sum;
}

// scale_sum_kernel_cpu (cpu/scale_sum_kernel.cpp:5-22): out[d] += sum_{i<=pos} scale[i] *
// value[i*stride + d], d < size; the terms are added in order of i, one thread per d.
__global__ void scale_sum_kernel(const float* __restrict__ value, const float* __restrict__ scale,
                                 float* out, int pos, int size, int stride KX_BPARAM(3)) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= size) return;
  float acc = *KX_CKP(2, out + d);
  for (int i = 0; i <= pos; ++i) acc = fmaf(*KX_CKP(1, scale + i), *KX_CKP(0, value + ((size_t)i * stride + d)), acc);
  out[d] = acc;
}

// quantize_q80 (tools/export_llama.py:49-73): per group of `gs` consecutive values
// scale = max|w| / 127, q = round_half_even(w / scale); one 64-lane wave per group.  IEEE division
// and rintf, so the bytes equal the exporter's; an all-zero group gives scale 0 and zeros.
__global__ void quantize_q80_kernel(const float* __restrict__ w, size_t groups, int gs,
                                    int8_t* __restrict__ q, float* __restrict__ scales KX_BPARAM(3)) {
  const int lane = threadIdx.x & 63;
  const size_t g = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (g >= groups) return;
  const float* wg = w + g * (size_t)gs;
  float wmax = 0.f;
  for (int i = lane; i < gs; i += 64) wmax = fmaxf(wmax, fabsf(*KX_CKP(0, wg + i)));
  wmax = wave_max(wmax);
  const float scale = __fdiv_rn(wmax, 127.0f);
  if (lane == 0) *KX_CKP(2, scales + g) = scale;
  for (int i = lane; i < gs; i += 64)
    *KX_CKP(1, q + (g * (size_t)gs + i)) = scale > 0.f ? (int8_t)rintf(__fdiv_rn(wg[i], scale)) : (int8_t)0;
}

// ---- seeded synthetic tensors, generated on the device ----
// Element (i, j) of the GLOBAL matrix gets a value that depends only on (seed, i*ld + j), so a
// tensor-parallel shard holds exactly the slice of the unsharded tensor.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void fill_hash_i8_kernel(int8_t* dst, int rows, int cols, int64_t ld, int64_t row0,
                                    int64_t col0, uint64_t seed KX_BPARAM(1)) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / cols, j = t - i * cols;
    const uint64_t h = mix64(seed ^ (uint64_t)((row0 + i) * ld + col0 + j) * 0xD1342543DE82EF95ull);
    *KX_CKP(0, dst + t) = (int8_t)((int)(h % 255ull) - 127);
  }
}
__global__ void fill_hash_f32_kernel(float* dst, int rows, int cols, int64_t ld, int64_t row0,
                                     int64_t col0, uint64_t seed, float lo, float hi KX_BPARAM(1)) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / cols, j = t - i * cols;
    const uint64_t h = mix64(seed ^ (uint64_t)((row0 + i) * ld + col0 + j) * 0xD1342543DE82EF95ull);
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
    *KX_CKP(0, dst + t) = lo + (hi - lo) * u;
  }
}

}  // namespace kx

using namespace kx;

extern "C" int kx_rmsnorm_f32(const float* in, const float* weight, float* out, int32_t dim,
                              int32_t rows, float eps, void* stream) {
  if (!in || !weight || !out || dim <= 0 || rows <= 0) return KX_EINVAL;
  const int threads = dim >= 4096 ? 512 : 256;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(threads), 0, (hipStream_t)stream, in, weight,
                     out, dim, eps KX_BPASS(3, in, weight, out));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_rmsnorm_f16out(const float* in, const float* weight, void* out16, int32_t dim,
                                 int32_t rows, float eps, void* stream) {
  if (!in || !weight || !out16 || dim <= 0 || rows <= 0) return KX_EINVAL;
  if (dim % 4 || ((uintptr_t)in | (uintptr_t)weight) % 16 || (uintptr_t)out16 % 8) return KX_EINVAL;
  const int threads = dim >= 4096 ? 512 : 256;
  hipLaunchKernelGGL(rmsnorm_f16out_kernel, dim3(rows), dim3(threads), 0, (hipStream_t)stream, in,
                     weight, static_cast<_Float16*>(out16), dim, eps KX_BPASS(3, in, weight, out16));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_sin_cos_cache_f32(int32_t head_size, int32_t max_seq_len, float theta,
                                    float* sin_cache, float* cos_cache, void* stream) {
  if (!sin_cache || !cos_cache || head_size <= 0 || max_seq_len <= 0) return KX_EINVAL;
  const int n = head_size * max_seq_len;
  hipLaunchKernelGGL(sin_cos_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     head_size, max_seq_len, theta, sin_cache, cos_cache KX_BPASS(2, sin_cache, cos_cache));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_rope_bias_f32(int32_t variant, int32_t dim, int32_t kv_dim, int32_t head_size,
                                float* q, float* k, int32_t k_stride, float* v, int32_t v_stride,
                                const float* bq, const float* bk, const float* bv, int32_t pos,
                                int32_t rows, const float* sin_cache, const float* cos_cache,
                                void* stream) {
  if (!q || !k || !sin_cache || !cos_cache) return KX_EINVAL;
  if (dim <= 0 || head_size <= 0 || head_size % 2 || dim % head_size || kv_dim % head_size ||
      kv_dim > dim || rows <= 0 || pos < 0)
    return KX_EINVAL;
  if (bv && !v) return KX_EINVAL;
  const kx::RopeBias b{bq, bk, bv, v, v_stride};
  const dim3 grid((dim / 2 + 127) / 128, rows);
  if (variant == KX_ROPE_INTERLEAVED) {
    hipLaunchKernelGGL(rope_kernel<KX_ROPE_INTERLEAVED>, grid, dim3(128), 0, (hipStream_t)stream,
                       dim, kv_dim, head_size, q, k, k_stride, pos, rows, sin_cache, cos_cache, b
                       KX_BPASS(8, q, k, sin_cache, cos_cache, bq, bk, bv, v));
  } else if (variant == KX_ROPE_HALF_SPLIT) {
    hipLaunchKernelGGL(rope_kernel<KX_ROPE_HALF_SPLIT>, grid, dim3(128), 0, (hipStream_t)stream, dim,
                       kv_dim, head_size, q, k, k_stride, pos, rows, sin_cache, cos_cache, b
                       KX_BPASS(8, q, k, sin_cache, cos_cache, bq, bk, bv, v));
  } else {
    return KX_EINVAL;
  }
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_rope_f32(int32_t variant, int32_t dim, int32_t kv_dim, int32_t head_size, float* q,
                           float* k, int32_t k_stride, int32_t pos, int32_t rows,
                           const float* sin_cache, const float* cos_cache, void* stream) {
  return kx_rope_bias_f32(variant, dim, kv_dim, head_size, q, k, k_stride, nullptr, 0, nullptr,
                          nullptr, nullptr, pos, rows, sin_cache, cos_cache, stream);
}

extern "C" int kx_swiglu_f32(const float* in1, const float* in2, float* out, int32_t n,
                             void* stream) {
  if (!in1 || !in2 || !out || n <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(swiglu_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in1,
                     in2, out, n KX_BPASS(3, in1, in2, out));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_add_f32(const float* a, const float* b, float* out, int32_t n, void* stream) {
  if (!a || !b || !out || n <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(add_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b,
                     out, n KX_BPASS(3, a, b, out));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_embedding_f32(const int32_t* tokens, int32_t n_tokens, const float* table,
                                int32_t dim, int32_t vocab_size, float* out, void* stream) {
  if (!tokens || !table || !out || n_tokens <= 0 || dim <= 0 || vocab_size <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(embedding_kernel, dim3(n_tokens), dim3(256), 0, (hipStream_t)stream, tokens,
                     table, out, dim, vocab_size KX_BPASS(3, tokens, table, out));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_argmax_f32(const float* logits, int32_t n, int32_t* out_index, void* stream) {
  if (!logits || !out_index || n <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, n,
                     out_index KX_BPASS(2, logits, out_index));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" size_t kx_step_workspace_bytes(void) { return kx::kStepWorkspaceBytes; }

static int step_pick_launch(kx::StepArgs a, void* workspace, size_t workspace_bytes, void* stream) {
  if (!workspace || workspace_bytes < kx_step_workspace_bytes()) return KX_EWORKSPACE;
  a.pv = static_cast<float*>(workspace);
  a.pi = reinterpret_cast<int32_t*>(a.pv + kx::kStepSlots);
  a.ticket = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + kx::kStepTicketOff);
  const int blocks = a.n < kx::kStepBlocks * 256 ? (a.n + 255) / 256 : kx::kStepBlocks;
  KX_BSET(a, SB_LOGITS, a.logits); KX_BSET(a, SB_PV, a.pv); KX_BSET(a, SB_PI, a.pi); KX_BSET(a, SB_TICKET, a.ticket);
  KX_BSET(a, SB_CAND, a.cand); KX_BSET(a, SB_TOKEN, a.token); KX_BSET(a, SB_POS, a.pos);
  KX_BSET(a, SB_HISTORY, a.history); KX_BSET(a, SB_TABLE, a.table); KX_BSET(a, SB_XOUT, a.x_out);
  hipLaunchKernelGGL(kx::step_pick_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_decode_step_end_embed(const float* logits, int32_t n, int32_t* token_dev,
                                        int32_t* pos_dev, int32_t* history_dev, int32_t history_len,
                                        int32_t write_history, const float* table, int32_t dim,
                                        int32_t vocab_size, float* x_out, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  if (!logits || !token_dev || !pos_dev || n <= 0) return KX_EINVAL;
  if (table && (!x_out || !history_dev || dim <= 0 || vocab_size <= 0)) return KX_EINVAL;
  kx::StepArgs a{};
  a.logits = logits; a.n = n;
  a.token = token_dev; a.pos = pos_dev; a.history = history_dev; a.history_len = history_len;
  a.write_history = write_history != 0;
  a.table = table; a.dim = dim; a.vocab = vocab_size; a.x_out = x_out;
  return step_pick_launch(a, workspace, workspace_bytes, stream);
}

extern "C" int kx_decode_step_end(const float* logits, int32_t n, int32_t* token_dev,
                                  int32_t* pos_dev, int32_t* history_dev, int32_t history_len,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  return kx_decode_step_end_embed(logits, n, token_dev, pos_dev, history_dev, history_len, 1, nullptr, 0,
                                  0, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int kx_argmax_candidate_f32(const float* logits, int32_t n, int32_t index_base,
                                       float* candidate, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  if (!logits || n <= 0 || index_base < 0 || !candidate) return KX_EINVAL;
  if ((int64_t)index_base + n > (1 << 24)) return KX_EUNSUPPORTED;  // indices travel as fp32
  kx::StepArgs a{};
  a.logits = logits; a.n = n; a.index_base = index_base; a.cand = candidate;
  return step_pick_launch(a, workspace, workspace_bytes, stream);
}

extern "C" int kx_decode_step_end_candidates_embed(const float* candidates, int32_t ranks,
                                                   int32_t* token_dev, int32_t* pos_dev,
                                                   int32_t* history_dev, int32_t history_len,
                                                   int32_t write_history, const float* table,
                                                   int32_t dim, int32_t vocab_size, float* x_out,
                                                   void* stream) {
  if (!candidates || ranks <= 0 || !token_dev || !pos_dev) return KX_EINVAL;
  if (table && (!x_out || !history_dev || dim <= 0 || vocab_size <= 0)) return KX_EINVAL;
  kx::StepArgs a{};
  a.token = token_dev; a.pos = pos_dev; a.history = history_dev; a.history_len = history_len;
  a.write_history = write_history != 0;
  a.table = table; a.dim = dim; a.vocab = vocab_size; a.x_out = x_out;
  KX_BSET(a, SB_LOGITS, nullptr); KX_BSET(a, SB_PV, nullptr); KX_BSET(a, SB_PI, nullptr); KX_BSET(a, SB_TICKET, nullptr);
  KX_BSET(a, SB_CAND, nullptr); KX_BSET(a, SB_TOKEN, a.token); KX_BSET(a, SB_POS, a.pos);
  KX_BSET(a, SB_HISTORY, a.history); KX_BSET(a, SB_TABLE, a.table); KX_BSET(a, SB_XOUT, a.x_out);
  hipLaunchKernelGGL(kx::step_end_candidates_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                     candidates, ranks, a KX_BPASS(1, candidates));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_decode_step_end_candidates(const float* candidates, int32_t ranks,
                                             int32_t* token_dev, int32_t* pos_dev,
                                             int32_t* history_dev, int32_t history_len, void* stream) {
  return kx_decode_step_end_candidates_embed(candidates, ranks, token_dev, pos_dev, history_dev,
                                             history_len, 1, nullptr, 0, 0, nullptr, stream);
}

extern "C" int kx_fill_hash_i8(int8_t* dst, int32_t rows, int32_t cols, int64_t ld, int64_t row0,
                               int64_t col0, uint64_t seed, void* stream) {
  if (!dst || rows <= 0 || cols <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(fill_hash_i8_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, dst, rows,
                     cols, ld, row0, col0, seed KX_BPASS(1, dst));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_fill_hash_f32(float* dst, int32_t rows, int32_t cols, int64_t ld, int64_t row0,
                                int64_t col0, uint64_t seed, float lo, float hi, void* stream) {
  if (!dst || rows <= 0 || cols <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(fill_hash_f32_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, dst, rows,
                     cols, ld, row0, col0, seed, lo, hi KX_BPASS(1, dst));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

// Reads `bytes` of memory and keeps nothing: brings them into the die-level cache ahead of the
// launch that will stream them (default cache policy, so the lines stay; the streaming kernels
// read with the non-temporal policy and leave them there for no one).  The sum of what was read
// goes to `sink` only if it equals a value it cannot practically take, which keeps the loads alive.
namespace kx {
__global__ __launch_bounds__(256) void touch_bytes_kernel(const i32x4* __restrict__ p, size_t n16,
                                                          int* __restrict__ sink KX_BPARAM(2)) {
  i32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x;
  for (; i + 768 < n16; i += stride) {
    const i32x4 a = *KX_CKP(0, p + i), b = *KX_CKP(0, p + i + 256), c = *KX_CKP(0, p + i + 512),
                d = *KX_CKP(0, p + i + 768);
    acc = acc ^ a ^ b ^ c ^ d;
  }
  for (; i < n16; i += 256) acc = acc ^ *KX_CKP(0, p + i);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x5a17c0de && acc.x == 0x7e57ab1e) *KX_CKP(1, sink) = acc.y;
}
}  // namespace kx

extern "C" int kx_touch_bytes(const void* ptr, size_t bytes, int32_t workgroups, int32_t* sink, void* stream) {
  if (!ptr || !sink || workgroups <= 0 || (uintptr_t)ptr % 16) return KX_EINVAL;
  if (bytes < 16) return KX_OK;
  hipLaunchKernelGGL(kx::touch_bytes_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const kx::i32x4*>(ptr), bytes / 16, sink KX_BPASS(2, ptr, sink));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_embedding_dpos_f32(const int32_t* history_dev, const int32_t* pos_dev,
                                     const float* table, int32_t dim, int32_t vocab_size,
                                     float* out, void* stream) {
  if (!history_dev || !pos_dev || !table || !out || dim <= 0 || vocab_size <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(embedding_dpos_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, history_dev,
                     pos_dev, table, out, dim, vocab_size KX_BPASS(4, history_dev, pos_dev, table, out));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_swiglu_rows_f32(const float* in, int64_t ld_in, int32_t half, float* out,
                                  int64_t ld_out, int32_t rows, void* stream) {
  if (!in || !out || half <= 0 || rows <= 0) return KX_EINVAL;
  if (half % 4 || ld_in % 4 || ld_out % 4 || ((uintptr_t)in | (uintptr_t)out) % 16) return KX_EINVAL;
  hipLaunchKernelGGL(swiglu_rows_kernel<float>, dim3(2048), dim3(256), 0, (hipStream_t)stream, in, ld_in,
                     half, out, ld_out, rows KX_BPASS(2, in, out));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_swiglu_rows_f16out(const float* in, int64_t ld_in, int32_t half, void* out16,
                                     int64_t ld_out, int32_t rows, void* stream) {
  if (!in || !out16 || half <= 0 || rows <= 0) return KX_EINVAL;
  if (half % 4 || ld_in % 4 || ld_out % 4 || (uintptr_t)in % 16 || (uintptr_t)out16 % 8) return KX_EINVAL;
  const int64_t n4 = (int64_t)rows * (half / 4);
  hipLaunchKernelGGL(swiglu_rows_kernel<_Float16>, dim3((unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048)),
                     dim3(256), 0, (hipStream_t)stream, in, ld_in, half, static_cast<_Float16*>(out16), ld_out,
                     rows KX_BPASS(2, in, out16));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_scale_f32(float scale, float* x, int32_t n, void* stream) {
  if (!x || n <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, scale, x, n KX_BPASS(1, x));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_softmax_f32(float* x, int32_t n, void* stream) {
  if (!x || n <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(softmax_kernel, dim3(1), dim3(n > 4096 ? 1024 : 256), 0, (hipStream_t)stream, x, n KX_BPASS(1, x));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_scale_sum_f32(const float* value, const float* scale, float* out, int32_t pos,
                                int32_t size, int32_t stride, void* stream) {
  if (!value || !scale || !out || pos < 0 || size <= 0 || stride < size) return KX_EINVAL;
  hipLaunchKernelGGL(scale_sum_kernel, dim3((size + 63) / 64), dim3(64), 0, (hipStream_t)stream, value,
                     scale, out, pos, size, stride KX_BPASS(3, value, scale, out));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_quantize_q80_f32(const float* w, size_t n, int32_t group_size, int8_t* q,
                                   float* scales, void* stream) {
  if (!w || !q || !scales || group_size <= 0 || n == 0 || n % (size_t)group_size) return KX_EINVAL;
  const size_t groups = n / (size_t)group_size;
  if ((groups + 3) / 4 > 0x7fffffffull) return KX_EINVAL;
  hipLaunchKernelGGL(quantize_q80_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, w, groups, group_size, q, scales KX_BPASS(3, w, q, scales));
  KX_LAUNCH_CHECK();
  return KX_OK;
}
