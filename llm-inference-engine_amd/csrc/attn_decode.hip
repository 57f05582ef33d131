// Single-token attention over the fp32 KV cache for gfx950 ("flash-decode").
//
// Stands in for multi_head_attention_kernel (kuiper/source/op/kernels/cuda/mha_kernel.cu:50-107)
// and its CPU twin (cpu/mha_kernel.cpp:5-61):
//   for each head h:  s[t] = (q_h . K[layer][t][kvh*hs..]) / sqrt(hs),  t <= pos
//                     p = softmax(s);  out_h = sum_t p[t] * V[layer][t][kvh*hs..]
// The reference gives one 256-thread block to each head and walks all t serially inside it.
// Here the (pos+1) cache rows are cut into 16-row tiles dealt over (kv head, split) workgroups
// so the whole chip streams the cache; one wavefront handles a tile for ALL kv_mul query heads
// that share the kv head, so K and V are read once per kv head, straight into VGPRs
// (128-bit loads).  Both products run on the exact-fp32 matrix instruction
// v_mfma_f32_16x16x4_f32 (an fp32 fma chain, bit-compatible with VALU fmaf): the score tile
// comes out with the key index on the register axis and the head on the lane, which is
// already the B-operand layout of the P.V product, so probabilities never cross lanes.
// Softmax is online (running max / sum); split results are merged by a second tiny kernel.
#include <math.h>

#include "kx_common.hpp"

namespace kx {

constexpr int kAttnWaves = 4;
constexpr int kMaxSplits = 64;

// workspace per (head, split): m, l, then O[hs]
__host__ __device__ inline size_t attn_ws_stride(int hs) { return (size_t)hs + 2; }

template <int HS>
__global__ __launch_bounds__(kAttnWaves * 64) void attn_decode_kernel(
    const float* __restrict__ q, const float* __restrict__ kc, const float* __restrict__ vc,
    float* __restrict__ out, float* __restrict__ ws, int last_val, const int32_t* __restrict__ pos_dev,
    int kv_mul, int kv_dim, float scale) {
  constexpr int NJ = HS / 16;                 // float4 K loads per lane per tile
  constexpr int NH = HS >= 64 ? HS / 64 : 1;  // 64-dim halves of the output
  constexpr int DLV = HS >= 64 ? 16 : HS / 4; // dim-lanes that carry data
  __shared__ float sm_m[kAttnWaves][16];
  __shared__ float sm_l[kAttnWaves][16];
  __shared__ float sm_o[kAttnWaves][16][HS];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int kvh = blockIdx.x, split = blockIdx.y;
  // The position may live in device memory so a captured launch can be replayed as it advances;
  // the grid is then sized for the longest context and surplus splits leave at once.
  const int last = pos_dev ? *pos_dev : last_val;
  const int ntiles = (last >> 4) + 1;
  const int tiles_per_split = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y;
  const int nsplit = (ntiles + tiles_per_split - 1) / tiles_per_split;
  if (split >= nsplit) return;
  const float* kbase = kc + (size_t)kvh * HS;
  const float* vbase = vc + (size_t)kvh * HS;

  // Q fragment: B operand, lane (head n, group g) holds Q[head][16j+4g+e]
  float qf[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < kv_mul)
      v = *reinterpret_cast<const f32x4*>(q + ((size_t)kvh * kv_mul + n) * HS + 16 * j + 4 * g);
    qf[j][0] = v.x; qf[j][1] = v.y; qf[j][2] = v.z; qf[j][3] = v.w;
  }

  f32x4 o[NH][4];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) o[h][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int t_begin = split * tiles_per_split;
  const int t_end = min(t_begin + tiles_per_split, ntiles);
  for (int t = t_begin + wave; t < t_end; t += kAttnWaves) {
    const int key0 = t << 4;
    // ---- K tile: lane (key n, g) loads K[key][16j+4g .. +4) ----
    f32x4 kf[NJ];
    {
      const float* kr = kbase + (size_t)min(key0 + n, last) * kv_dim + 4 * g;
#pragma unroll
      for (int j = 0; j < NJ; ++j) kf[j] = *reinterpret_cast<const f32x4*>(kr + 16 * j);
    }
    // ---- V tile: lane (dim-lane n, g) loads V[key0+4g+r][64h+4n .. +4) ----
    f32x4 vf[4][NH];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* vr = vbase + (size_t)min(key0 + 4 * g + r, last) * kv_dim + 4 * n;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        vf[r][h] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (n < DLV) vf[r][h] = *reinterpret_cast<const f32x4*>(vr + 64 * h);
      }
    }
    // ---- S^T[key][head] = K . Q^T ----
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].x, qf[j][0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].y, qf[j][1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].z, qf[j][2], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].w, qf[j][3], s, 0, 0, 0);
    }
    // lane (head n, g) now holds scores of keys key0+4g+r, r=0..3
    float sv[4] = {s.x * scale, s.y * scale, s.z * scale, s.w * scale};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (key0 + 4 * g + r > last) sv[r] = -INFINITY;
    float mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
    mx = fmaxf(mx, lane_xor16(mx));
    mx = fmaxf(mx, lane_xor32(mx));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);  // first tile: exp(-inf) = 0
    m_run = m_new;
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = expf(sv[r] - m_new);
    l_run = l_run * alpha + ((p[0] + p[1]) + (p[2] + p[3]));
    // The output tile keeps the head on the lane (column), so alpha is lane-local.
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[h][e] *= alpha;
    // ---- O^T[dim][head] += V^T . P^T ----
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        o[h][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][h].x, p[r], o[h][0], 0, 0, 0);
        o[h][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][h].y, p[r], o[h][1], 0, 0, 0);
        o[h][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][h].z, p[r], o[h][2], 0, 0, 0);
        o[h][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][h].w, p[r], o[h][3], 0, 0, 0);
      }
  }

  // ---- wave result -> LDS.  lane (head n, g'): o[h][e][r'] = O[head][64h + 4(4g'+r') + e] ----
  l_run += lane_xor16(l_run);
  l_run += lane_xor32(l_run);
  if (n < kv_mul) {
    if (g == 0) {
      sm_m[wave][n] = m_run;
      sm_l[wave][n] = l_run;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float vals[4] = {o[h][e].x, o[h][e].y, o[h][e].z, o[h][e].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 64 * h + 4 * (4 * g + r) + e;
          if (d < HS) sm_o[wave][n][d] = vals[r];
        }
      }
  }
  __syncthreads();

  // ---- merge the workgroup's waves; write the final row or a split partial ----
  for (int idx = threadIdx.x; idx < kv_mul * HS; idx += blockDim.x) {
    const int head = idx / HS, d = idx % HS;
    float M = sm_m[0][head];
#pragma unroll
    for (int w = 1; w < kAttnWaves; ++w) M = fmaxf(M, sm_m[w][head]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < kAttnWaves; ++w) {
      const float f = expf(sm_m[w][head] - M);
      L += sm_l[w][head] * f;
      O += sm_o[w][head][d] * f;
    }
    const int hq = kvh * kv_mul + head;
    if (nsplit == 1) {
      out[(size_t)hq * HS + d] = O / L;
    } else {
      float* dst = ws + ((size_t)hq * gridDim.y + split) * attn_ws_stride(HS);
      if (d == 0) {
        dst[0] = M;
        dst[1] = L;
      }
      dst[2 + d] = O;
    }
  }
}

// out[h][d] = sum_s O_s[d] e^{m_s-M} / sum_s l_s e^{m_s-M}
// One 256-thread workgroup per head: the split weights are formed once (one split per lane),
// then the O rows are summed with the splits dealt over thread groups so the loads of
// different splits are independent and in flight together.
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ ws,
                                                         float* __restrict__ out, int hs,
                                                         int nsplit_grid, int last_val,
                                                         const int32_t* __restrict__ pos_dev) {
  __shared__ float sf[kMaxSplits];
  __shared__ float red[4];
  __shared__ float part[256];
  const int hq = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int last = pos_dev ? *pos_dev : last_val;
  const int ntiles = (last >> 4) + 1;
  const int tps = (ntiles + nsplit_grid - 1) / nsplit_grid;
  const int nsplit = (ntiles + tps - 1) / tps;
  if (nsplit == 1) return;  // the attention kernel wrote the final rows itself
  const size_t stride = attn_ws_stride(hs);
  const float* base = ws + (size_t)hq * nsplit_grid * stride;
  // nsplit <= kMaxSplits = 64: wave 0 holds one split per lane
  if (wave == 0) {
    float m = -INFINITY, l = 0.f;
    if (lane < nsplit) {
      m = base[(size_t)lane * stride];
      l = base[(size_t)lane * stride + 1];
    }
    const float M = wave_max(m);
    const float f = lane < nsplit ? expf(m - M) : 0.f;
    sf[lane] = f;
    const float L = wave_sum(f * l);
    if (lane == 0) red[0] = L;
  }
  __syncthreads();
  const float inv_l = 1.0f / red[0];
  // thread (group g, dim d): group g sums splits g, g+G, ...
  const int G = 256 / hs > 0 ? 256 / hs : 1;
  const int g = tid / hs, d = tid % hs;
  float acc = 0.f;
  if (g < G) {
#pragma unroll 4
    for (int s = g; s < nsplit; s += G) acc = fmaf(base[(size_t)s * stride + 2 + d], sf[s], acc);
  }
  part[tid] = acc;
  __syncthreads();
  if (tid < hs) {
    float o = 0.f;
    for (int k = 0; k < G; ++k) o += part[k * hs + tid];
    out[(size_t)hq * hs + tid] = o * inv_l;
  }
}

// Any head size / any kv_mul: one workgroup per head, scores kept in a [head][seq_len] scratch
// (the reference's score tensor has exactly that shape, llama3.cpp:523-525).
__global__ void attn_decode_generic_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                           const float* __restrict__ vc, float* __restrict__ out,
                                           float* __restrict__ score, int last, int kv_mul,
                                           int kv_dim, int hs, int seq_len, float scale) {
  __shared__ float red[16];
  const int h = blockIdx.x;
  const float* qh = q + (size_t)h * hs;
  const size_t hoff = (size_t)(h / kv_mul) * hs;
  float* sc = score + (size_t)h * seq_len;
  float mx = -INFINITY;
  for (int t = threadIdx.x; t <= last; t += blockDim.x) {
    const float* kr = kc + (size_t)t * kv_dim + hoff;
    float acc = 0.f;
    for (int d = 0; d < hs; ++d) acc = fmaf(qh[d], kr[d], acc);
    acc *= scale;
    sc[t] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mx = fmaxf(mx, red[w]);
  __syncthreads();
  float sum = 0.f;
  for (int t = threadIdx.x; t <= last; t += blockDim.x) {
    const float e = expf(sc[t] - mx);
    sc[t] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();  // also publishes sc[] to the whole workgroup
  sum = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += red[w];
  for (int d = threadIdx.x; d < hs; d += blockDim.x) {
    float acc = 0.f;
    for (int t = 0; t <= last; ++t) acc = fmaf(sc[t], vc[(size_t)t * kv_dim + hoff + d], acc);
    out[(size_t)h * hs + d] = acc / sum;
  }
}

}  // namespace kx

extern "C" size_t kx_attn_workspace_bytes(int32_t head_num, int32_t head_size) {
  if (head_num <= 0 || head_size <= 0) return 0;
  return (size_t)head_num * kx::kMaxSplits * kx::attn_ws_stride(head_size) * sizeof(float);
}

static int mha_decode_launch(const float* query, const float* key_cache, const float* value_cache,
                             float* out, int32_t pos, const int32_t* pos_dev, int32_t head_num,
                             int32_t kv_mul, int32_t head_size, int32_t kv_dim, int32_t seq_len,
                             int32_t layer_index, void* workspace, size_t workspace_bytes,
                             void* stream) {
  using namespace kx;
  if (!query || !key_cache || !value_cache || !out) return KX_EINVAL;
  if (pos < 0 || pos >= seq_len || head_num <= 0 || kv_mul <= 0 || head_num % kv_mul) return KX_EINVAL;
  const int kvh = head_num / kv_mul;
  if (kv_dim != kvh * head_size) return KX_EINVAL;
  const size_t layer_off = (size_t)layer_index * (size_t)seq_len * (size_t)kv_dim;
  const float* kc = key_cache + layer_off;
  const float* vc = value_cache + layer_off;
  const float scale = 1.0f / sqrtf((float)head_size);  // mha_kernel.cu / cpu/mha_kernel.cpp:11
  // one MFMA column block holds the heads of a kv head; the tiled kernel is built for the
  // head sizes of the supported model families
  const bool tiled = kv_mul <= 16 && (head_size == 32 || head_size == 64 || head_size == 128) &&
                     (((uintptr_t)query | (uintptr_t)kc | (uintptr_t)vc) % 16 == 0) &&
                     kv_dim % 4 == 0;
  if (!tiled) {
    if (pos_dev) return KX_EUNSUPPORTED;
    if (!workspace || workspace_bytes < (size_t)head_num * seq_len * sizeof(float))
      return KX_EWORKSPACE;
    hipLaunchKernelGGL(attn_decode_generic_kernel, dim3(head_num), dim3(256), 0,
                       (hipStream_t)stream, query, kc, vc, out, reinterpret_cast<float*>(workspace),
                       pos, kv_mul, kv_dim, head_size, seq_len, scale);
    KX_LAUNCH_CHECK();
    return KX_OK;
  }

  // `pos` sizes the grid: the actual position when passed by value, the largest position the
  // launch may ever see when it is read from pos_dev.
  const int ntiles = (pos >> 4) + 1;
  // one tile per wave is the finest useful cut; keep the grid near 2 workgroups per CU
  int nsplit = (ntiles + kAttnWaves - 1) / kAttnWaves;
  const int cap_grid = (512 + kvh - 1) / kvh;
  if (nsplit > cap_grid) nsplit = cap_grid;
  if (nsplit > kMaxSplits) nsplit = kMaxSplits;
  const size_t per_split = (size_t)head_num * attn_ws_stride(head_size) * sizeof(float);
  if (nsplit > 1) {
    const size_t fit = workspace ? workspace_bytes / per_split : 0;
    if (fit < (size_t)nsplit) nsplit = fit < 1 ? 1 : (int)fit;
  }
  const int tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;  // no empty splits at `pos`
  float* ws = reinterpret_cast<float*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(kvh, nsplit), block(kAttnWaves * 64);
  switch (head_size) {
    case 32:
      hipLaunchKernelGGL(attn_decode_kernel<32>, grid, block, 0, s, query, kc, vc, out, ws, pos,
                         pos_dev, kv_mul, kv_dim, scale);
      break;
    case 64:
      hipLaunchKernelGGL(attn_decode_kernel<64>, grid, block, 0, s, query, kc, vc, out, ws, pos,
                         pos_dev, kv_mul, kv_dim, scale);
      break;
    case 128:
      hipLaunchKernelGGL(attn_decode_kernel<128>, grid, block, 0, s, query, kc, vc, out, ws, pos,
                         pos_dev, kv_mul, kv_dim, scale);
      break;
    default:
      return KX_EUNSUPPORTED;
  }
  KX_LAUNCH_CHECK();
  if (nsplit > 1) {
    hipLaunchKernelGGL(attn_merge_kernel, dim3(head_num), dim3(256), 0, s, ws, out, head_size,
                       nsplit, pos, pos_dev);
    KX_LAUNCH_CHECK();
  }
  return KX_OK;
}

extern "C" int kx_mha_decode_f32(const float* query, const float* key_cache,
                                 const float* value_cache, float* out, int32_t pos,
                                 int32_t head_num, int32_t kv_mul, int32_t head_size, int32_t kv_dim,
                                 int32_t seq_len, int32_t layer_index, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return mha_decode_launch(query, key_cache, value_cache, out, pos, nullptr, head_num, kv_mul,
                           head_size, kv_dim, seq_len, layer_index, workspace, workspace_bytes,
                           stream);
}

extern "C" int kx_mha_decode_dpos_f32(const float* query, const float* key_cache,
                                      const float* value_cache, float* out,
                                      const int32_t* pos_dev, int32_t max_pos, int32_t head_num,
                                      int32_t kv_mul, int32_t head_size, int32_t kv_dim,
                                      int32_t seq_len, int32_t layer_index, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (!pos_dev) return KX_EINVAL;
  return mha_decode_launch(query, key_cache, value_cache, out, max_pos, pos_dev, head_num, kv_mul,
                           head_size, kv_dim, seq_len, layer_index, workspace, workspace_bytes,
                           stream);
}
