// W8A32 group-quant GEMV for gfx950 (decode path).
//
//   y[p] = sum_i x[i] * scales[(p*M+i)/gs] * (float)w[p*M+i]
//
// Stands in for matmul_kernel_cu_fp32int8 (kuiper/source/op/kernels/cuda/matmul_kernel.cu:56-87).
// The reference walks a row with 128 threads reading single bytes at stride 128 and divides
// per element to find the group.  Here one 64-lane wavefront owns whole rows: every lane
// issues 128-bit non-temporal loads of 16 consecutive int8 (1 KiB per wave instruction, fully
// coalesced), the 16 elements of a lane always sit inside one quant group so its scale is one
// register, x is staged once per workgroup in LDS in a layout that makes the four 16-byte
// reads per weight load conflict-free, and the row sum is finished with DPP + swizzle.
//
// Rounding order differs from the reference only in where the scale is applied: the 16
// products x*w of a lane are summed first and the group scale multiplies that partial sum.
#include "kx_common.hpp"

namespace kx {

struct GemvW8Args {
  const float* x;
  const int8_t* w;
  const float* scales;
  float* y0;
  float* y1;
  float* y2;
  const float* bias;
  const float* residual;
  int32_t k0, k1;          // row routing: [0,k0)->y0, [k0,k1)->y1, [k1,K)->y2
  int32_t M, K;
  int32_t chunks;          // ceil(M / 1024)
  int32_t gs_shift;        // log2(group_size)
  int32_t groups_per_row;  // M / group_size
  int32_t rows_per_block;
  int32_t swiglu_half;     // >0: rows p and p+swiglu_half are gate/up of output p
};

__device__ __forceinline__ float dot16(const i32x4 wv, const f32x4 x0, const f32x4 x1,
                                       const f32x4 x2, const f32x4 x3) {
  float a = 0.f;
#define KX_B(word, sh) ((float)(int8_t)((word) >> (sh)))
  a = fmaf(x0.x, KX_B(wv.x, 0), a);
  a = fmaf(x0.y, KX_B(wv.x, 8), a);
  a = fmaf(x0.z, KX_B(wv.x, 16), a);
  a = fmaf(x0.w, KX_B(wv.x, 24), a);
  a = fmaf(x1.x, KX_B(wv.y, 0), a);
  a = fmaf(x1.y, KX_B(wv.y, 8), a);
  a = fmaf(x1.z, KX_B(wv.y, 16), a);
  a = fmaf(x1.w, KX_B(wv.y, 24), a);
  a = fmaf(x2.x, KX_B(wv.z, 0), a);
  a = fmaf(x2.y, KX_B(wv.z, 8), a);
  a = fmaf(x2.z, KX_B(wv.z, 16), a);
  a = fmaf(x2.w, KX_B(wv.z, 24), a);
  a = fmaf(x3.x, KX_B(wv.w, 0), a);
  a = fmaf(x3.y, KX_B(wv.w, 8), a);
  a = fmaf(x3.z, KX_B(wv.w, 16), a);
  a = fmaf(x3.w, KX_B(wv.w, 24), a);
#undef KX_B
  return a;
}

__device__ __forceinline__ void store_row(const GemvW8Args& a, int row, float v) {
  if (a.bias) v += a.bias[row];
  float* dst;
  int r = row;
  if (row < a.k0) {
    dst = a.y0;
  } else if (row < a.k1) {
    dst = a.y1;
    r = row - a.k0;
  } else {
    dst = a.y2;
    r = row - a.k1;
  }
  if (a.residual) v += a.residual[row];
  dst[r] = v;
}

// R rows per wave at a time, U chunk loads per row in flight.  GU: rows p and p+swiglu_half are
// the gate and up rows of output p (fused [W1;W3] + SwiGLU).
template <int R, int U, bool GU>
__global__ void gemv_w8_kernel(const GemvW8Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs = reinterpret_cast<f32x4*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nwaves = blockDim.x >> 6;
  const int M = a.M;
  const int row_begin = blockIdx.x * a.rows_per_block;
  const int row_end = min(row_begin + a.rows_per_block, GU ? a.swiglu_half : a.K);

  // ---- stage x: slot (c*4+p)*64+l  <-  x[c*1024 + l*16 + p*4 .. +4) ; zero past M ----
  {
    const f32x4* xg = reinterpret_cast<const f32x4*>(a.x);
    const int n4 = a.chunks * 256;
    const int m4 = M >> 2;
    for (int i = tid; i < n4; i += blockDim.x) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < m4) v = xg[i];
      const int c = i >> 8, l = (i >> 2) & 63, p = i & 3;
      xs[((c * 4 + p) << 6) + l] = v;
    }
  }
  __syncthreads();

  const int col_lane = lane * 16;
  for (int r0 = row_begin + wave * R; r0 < row_end; r0 += nwaves * R) {
    constexpr int R2 = GU ? R : 1;
    constexpr int U2 = GU ? U : 1;
    float total[R];
    float total2[R2];  // up-projection rows (fused gate/up)
#pragma unroll
    for (int r = 0; r < R; ++r) total[r] = 0.f;
#pragma unroll
    for (int r = 0; r < R2; ++r) total2[r] = 0.f;

    for (int c0 = 0; c0 < a.chunks; c0 += U) {
      i32x4 wv[R][U];
      float sc[R][U];
      i32x4 wv2[R2][U2];
      float sc2[R2][U2];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int row = min(r0 + r, row_end - 1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int col = (c0 + u) * 1024 + col_lane;
          if (col < M) {
            const size_t off = (size_t)row * (size_t)M + (size_t)col;
            wv[r][u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(a.w + off));
            sc[r][u] = __builtin_nontemporal_load(
                a.scales + (size_t)row * a.groups_per_row + (col >> a.gs_shift));
            if constexpr (GU) {
              const size_t off2 = off + (size_t)a.swiglu_half * (size_t)M;
              wv2[r][u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(a.w + off2));
              sc2[r][u] = __builtin_nontemporal_load(
                  a.scales + (size_t)(row + a.swiglu_half) * a.groups_per_row +
                  (col >> a.gs_shift));
            }
          } else {
            wv[r][u] = i32x4{0, 0, 0, 0};
            sc[r][u] = 0.f;
            if constexpr (GU) {
              wv2[r][u] = i32x4{0, 0, 0, 0};
              sc2[r][u] = 0.f;
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (c0 + u < a.chunks) {
          const int base = ((c0 + u) * 4) << 6;
          const f32x4 x0 = xs[base + lane];
          const f32x4 x1 = xs[base + 64 + lane];
          const f32x4 x2 = xs[base + 128 + lane];
          const f32x4 x3 = xs[base + 192 + lane];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            total[r] = fmaf(dot16(wv[r][u], x0, x1, x2, x3), sc[r][u], total[r]);
            if constexpr (GU)
              total2[r] = fmaf(dot16(wv2[r][u], x0, x1, x2, x3), sc2[r][u], total2[r]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float v = wave_sum(total[r]);
      if constexpr (GU) {
        const float up = wave_sum(total2[r]);
        // swiglu_kernel_cu (cuda/swiglu_kernel.cu:4-22): gate * sigmoid(gate) * up
        v = v * (1.0f / (1.0f + expf(-v))) * up;
      }
      const int row = r0 + r;
      if (lane == 0 && row < row_end) store_row(a, row, v);
    }
  }
}

// Any M % 4 == 0 and any group size, groups may straddle rows: one wave per row, 4 elements
// per lane per step, group index computed per element as the reference does.
__global__ void gemv_w8_generic_kernel(const GemvW8Args a, int32_t group_size) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave;
  if (row >= a.K) return;
  float total = 0.f;
  const size_t base = (size_t)row * (size_t)a.M;
  for (int i = lane * 4; i < a.M; i += 256) {
    const int32_t wv = *reinterpret_cast<const int32_t*>(a.w + base + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t idx = base + (size_t)(i + e);
      const float s = a.scales[idx / (size_t)group_size];
      total = fmaf(a.x[i + e] * s, (float)(int8_t)(wv >> (8 * e)), total);
    }
  }
  total = wave_sum(total);
  if (lane == 0) store_row(a, row, total);
}

template <int R, int U, bool GU>
static int launch(const GemvW8Args& a, int threads, hipStream_t s) {
  const int out_rows = GU ? a.swiglu_half : a.K;
  const int grid = (out_rows + a.rows_per_block - 1) / a.rows_per_block;
  const size_t lds = (size_t)a.chunks * 1024 * sizeof(float);
  if (lds > 64 * 1024) {
    static bool raised = false;  // per instantiation; the attribute is sticky
    if (!raised) {
      KX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_w8_kernel<R, U, GU>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised = true;
    }
  }
  hipLaunchKernelGGL((gemv_w8_kernel<R, U, GU>), dim3(grid), dim3(threads), lds, s, a);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

static int ilog2(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

static int gemv_w8_dispatch(GemvW8Args a, int32_t group_size, hipStream_t s) {
  if (!a.x || !a.w || !a.scales || !a.y0) return KX_EINVAL;
  if (a.M <= 0 || a.K <= 0 || group_size <= 0) return KX_EINVAL;
  if (a.M % 4 != 0) return KX_EINVAL;  // matmul_kernel.cu:123
  if (((size_t)a.M * (size_t)a.K) % (size_t)group_size != 0) return KX_EINVAL;
  const bool pow2 = (group_size & (group_size - 1)) == 0;
  const bool fast = pow2 && group_size % 16 == 0 && a.M % group_size == 0 && a.M % 16 == 0 &&
                    ((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.w % 16 == 0) &&
                    (size_t)a.M * 4 <= 160 * 1024;
  if (!fast) {
    if (a.swiglu_half > 0) return KX_EUNSUPPORTED;
    const int wpb = 4;
    hipLaunchKernelGGL(gemv_w8_generic_kernel, dim3((a.K + wpb - 1) / wpb), dim3(wpb * 64), 0, s, a,
                       group_size);
    KX_LAUNCH_CHECK();
    return KX_OK;
  }
  a.chunks = (a.M + 1023) / 1024;
  a.gs_shift = ilog2(group_size);
  a.groups_per_row = a.M / group_size;
  const size_t lds = (size_t)a.chunks * 4096;
  // waves per CU are bounded by how many x copies fit in the 160 KiB LDS
  int threads = 256;
  if (lds > 32 * 1024) threads = 512;
  if (lds > 80 * 1024) threads = 1024;
  const int waves = threads / 64;
  const int out_rows = a.swiglu_half > 0 ? a.swiglu_half : a.K;
  const bool fused2 = a.swiglu_half > 0;
  // rows a wave keeps in flight together; the gate/up form already doubles the loads
  const int R = fused2 ? 1 : (a.chunks <= 2 ? 4 : 2);
  const int per_iter = waves * R;
  // aim for ~4 workgroups per CU on big matrices, one pass on small ones
  const int target_blocks = 256 * 4;
  int rpb = (out_rows + target_blocks - 1) / target_blocks;
  rpb = ((rpb + per_iter - 1) / per_iter) * per_iter;
  a.rows_per_block = rpb;
  if (fused2) {
    return a.chunks <= 2 ? launch<1, 2, true>(a, threads, s) : launch<1, 4, true>(a, threads, s);
  }
  if (a.chunks <= 2) return launch<4, 2, false>(a, threads, s);
  return launch<2, 4, false>(a, threads, s);
}

}  // namespace kx

extern "C" int kx_gemv_w8_f32_ex(const float* x, const int8_t* w, const float* scales,
                                 int32_t group_size, float* y0, float* y1, float* y2, int32_t k0,
                                 int32_t k1, const float* bias, const float* residual, int32_t M,
                                 int32_t K, void* stream) {
  kx::GemvW8Args a{};
  a.x = x; a.w = w; a.scales = scales;
  a.y0 = y0; a.y1 = y1 ? y1 : y0; a.y2 = y2 ? y2 : y0;
  if (!y1 && !y2) { k0 = K; k1 = K; }
  if (k0 < 0 || k1 < k0 || k1 > K) return KX_EINVAL;
  a.k0 = k0; a.k1 = k1;
  a.bias = bias; a.residual = residual;
  a.M = M; a.K = K;
  a.swiglu_half = 0;
  return kx::gemv_w8_dispatch(a, group_size, (hipStream_t)stream);
}

extern "C" int kx_gemv_w8_f32(const float* x, const int8_t* w, const float* scales,
                              int32_t group_size, float* y, int32_t M, int32_t K, void* stream) {
  return kx_gemv_w8_f32_ex(x, w, scales, group_size, y, nullptr, nullptr, K, K, nullptr, nullptr, M,
                           K, stream);
}

extern "C" int kx_gemv_w8_swiglu_f32(const float* x, const int8_t* w13, const float* scales13,
                                     int32_t group_size, float* out, int32_t M, int32_t H,
                                     void* stream) {
  kx::GemvW8Args a{};
  a.x = x; a.w = w13; a.scales = scales13;
  a.y0 = a.y1 = a.y2 = out;
  a.k0 = a.k1 = H;
  a.M = M; a.K = 2 * H;
  a.swiglu_half = H;
  return kx::gemv_w8_dispatch(a, group_size, (hipStream_t)stream);
}
