// Support kernels of the token path: RMSNorm, RoPE (+ table), SwiGLU, add, embedding gather,
// greedy argmax.  Each is O(dim) bytes per token next to O(dim^2) weight bytes, so they are kept
// simple, vectorised and correct; the reference twins are cited per kernel.
#include "kx_common.hpp"

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

// row_rmsnorm_f32 (kuiper/source/op/kernels/cuda/rmsnorm_kernel.cu:62-108):
// out = weight * (x * rsqrt(mean(x^2) + eps)); one workgroup per row.
__global__ void rmsnorm_kernel(const float* __restrict__ in, const float* __restrict__ weight,
                               float* __restrict__ out, int dim, float eps) {
  __shared__ float red[16];
  const float* x = in + (size_t)blockIdx.x * dim;
  float* o = out + (size_t)blockIdx.x * dim;
  float ss = 0.f;
  const bool vec = (dim % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)o % 16 == 0) &&
                   ((uintptr_t)weight % 16 == 0);
  if (vec) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (int i = threadIdx.x; i < dim / 4; i += blockDim.x) {
      const f32x4 v = x4[i];
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  } else {
    for (int i = threadIdx.x; i < dim; i += blockDim.x) ss += x[i] * x[i];
  }
  ss = block_sum(ss, red);
  const float rs = 1.0f / sqrtf(ss / (float)dim + eps);
  if (vec) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(weight);
    f32x4* o4 = reinterpret_cast<f32x4*>(o);
    for (int i = threadIdx.x; i < dim / 4; i += blockDim.x) {
      const f32x4 v = x4[i], w = w4[i];
      o4[i] = f32x4{w.x * (rs * v.x), w.y * (rs * v.y), w.z * (rs * v.z), w.w * (rs * v.w)};
    }
  } else {
    for (int i = threadIdx.x; i < dim; i += blockDim.x) o[i] = weight[i] * (rs * x[i]);
  }
}

// sin_cos_calc (cuda/rope_kernel.cu:38-49): table[pos][head_dim], freq = theta^(-head_dim/hs).
__global__ void sin_cos_kernel(int head_size, int max_seq_len, float theta, float* sin_cache,
                               float* cos_cache) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= head_size * max_seq_len) return;
  const int pos = idx / head_size, hd = idx % head_size;
  const float freq = 1.0f / powf(theta, (float)hd / (float)head_size);
  const float val = (float)pos * freq;
  sin_cache[idx] = sinf(val);
  cos_cache[idx] = cosf(val);
}

// rope_kernel_cu_fp32 (cuda/rope_kernel.cu:5-36 half-split, :98-128 interleaved).
// One thread per rotation pair of q; pairs inside kv_dim also rotate k.
template <int VARIANT>
__global__ void rope_kernel(int dim, int kv_dim, int head_size, float* q, float* k, int k_stride,
                            int pos, int rows, const float* __restrict__ sin_cache,
                            const float* __restrict__ cos_cache) {
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
  const float fci = sin_cache[(size_t)p * head_size + tab];
  const float fcr = cos_cache[(size_t)p * head_size + tab];
  float* qr = q + (size_t)row * dim;
  const float q0 = qr[i0], q1 = qr[i1];
  qr[i0] = q0 * fcr - q1 * fci;
  qr[i1] = q0 * fci + q1 * fcr;
  const int head_base = VARIANT == KX_ROPE_INTERLEAVED ? i0 : (i0 / head_size) * head_size;
  if (head_base < kv_dim) {
    float* kr = k + (size_t)row * k_stride;
    const float k0 = kr[i0], k1 = kr[i1];
    kr[i0] = k0 * fcr - k1 * fci;
    kr[i1] = k0 * fci + k1 * fcr;
  }
}

// swiglu_kernel_cu_fp32 (cuda/swiglu_kernel.cu:4-22)
__global__ void swiglu_kernel(const float* __restrict__ a, const float* __restrict__ b,
                              float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = a[i];
  out[i] = g * (1.0f / (1.0f + expf(-g))) * b[i];
}

// add_kernel_cu_fp32 (cuda/add_kernel.cu:4-12)
__global__ void add_kernel(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

// emb_kernel_cu_fp32 (cuda/emb_kernel.cu:3-21): one workgroup per token row.
__global__ void embedding_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ table,
                                 float* __restrict__ out, int dim, int vocab) {
  const int t = blockIdx.x;
  int tok = tokens[t];
  if (tok < 0 || tok >= vocab) tok = 0;  // the reference aborts on an out-of-range id
  const float* src = table + (size_t)tok * dim;
  float* dst = out + (size_t)t * dim;
  if (dim % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0)) {
    for (int i = threadIdx.x; i < dim / 4; i += blockDim.x)
      reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
  } else {
    for (int i = threadIdx.x; i < dim; i += blockDim.x) dst[i] = src[i];
  }
}

// argmax_kernel_fp32 (cuda/argmax_kernel.cu:49-71): first maximum wins.  Single workgroup.
__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}
__global__ void argmax_kernel(const float* __restrict__ x, int n, int32_t* out) {
  __shared__ float sv[16];
  __shared__ int si[16];
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = x[i];
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
    *out = bi == 0x7fffffff ? 0 : bi;
  }
}

}  // namespace kx

using namespace kx;

extern "C" int kx_rmsnorm_f32(const float* in, const float* weight, float* out, int32_t dim,
                              int32_t rows, float eps, void* stream) {
  if (!in || !weight || !out || dim <= 0 || rows <= 0) return KX_EINVAL;
  const int threads = dim >= 4096 ? 512 : 256;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(threads), 0, (hipStream_t)stream, in, weight,
                     out, dim, eps);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_sin_cos_cache_f32(int32_t head_size, int32_t max_seq_len, float theta,
                                    float* sin_cache, float* cos_cache, void* stream) {
  if (!sin_cache || !cos_cache || head_size <= 0 || max_seq_len <= 0) return KX_EINVAL;
  const int n = head_size * max_seq_len;
  hipLaunchKernelGGL(sin_cos_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     head_size, max_seq_len, theta, sin_cache, cos_cache);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_rope_f32(int32_t variant, int32_t dim, int32_t kv_dim, int32_t head_size, float* q,
                           float* k, int32_t k_stride, int32_t pos, int32_t rows,
                           const float* sin_cache, const float* cos_cache, void* stream) {
  if (!q || !k || !sin_cache || !cos_cache) return KX_EINVAL;
  if (dim <= 0 || head_size <= 0 || head_size % 2 || dim % head_size || kv_dim % head_size ||
      rows <= 0 || pos < 0)
    return KX_EINVAL;
  const dim3 grid((dim / 2 + 127) / 128, rows);
  if (variant == KX_ROPE_INTERLEAVED) {
    hipLaunchKernelGGL(rope_kernel<KX_ROPE_INTERLEAVED>, grid, dim3(128), 0, (hipStream_t)stream,
                       dim, kv_dim, head_size, q, k, k_stride, pos, rows, sin_cache, cos_cache);
  } else if (variant == KX_ROPE_HALF_SPLIT) {
    hipLaunchKernelGGL(rope_kernel<KX_ROPE_HALF_SPLIT>, grid, dim3(128), 0, (hipStream_t)stream, dim,
                       kv_dim, head_size, q, k, k_stride, pos, rows, sin_cache, cos_cache);
  } else {
    return KX_EINVAL;
  }
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_swiglu_f32(const float* in1, const float* in2, float* out, int32_t n,
                             void* stream) {
  if (!in1 || !in2 || !out || n <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(swiglu_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in1,
                     in2, out, n);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_add_f32(const float* a, const float* b, float* out, int32_t n, void* stream) {
  if (!a || !b || !out || n <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(add_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b,
                     out, n);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_embedding_f32(const int32_t* tokens, int32_t n_tokens, const float* table,
                                int32_t dim, int32_t vocab_size, float* out, void* stream) {
  if (!tokens || !table || !out || n_tokens <= 0 || dim <= 0 || vocab_size <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(embedding_kernel, dim3(n_tokens), dim3(256), 0, (hipStream_t)stream, tokens,
                     table, out, dim, vocab_size);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_argmax_f32(const float* logits, int32_t n, int32_t* out_index, void* stream) {
  if (!logits || !out_index || n <= 0) return KX_EINVAL;
  hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, n,
                     out_index);
  KX_LAUNCH_CHECK();
  return KX_OK;
}
