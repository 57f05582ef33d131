// fp32 GEMV for gfx950: y = (W[K][M] . x) * scale.
// Stands in for matmul_kernel_cu_fp32 (kuiper/source/op/kernels/cuda/matmul_kernel.cu:6-54) and
// matmul_kernel_cpu (cpu/matmul_kernel.cpp:5-41).  One wavefront per row, 128-bit loads
// (256 floats per wave instruction), x staged once per workgroup in LDS, DPP / permlane-swap reduce.
#define KX_FILE_ID 8
#include "kx_common.hpp"

namespace kx {

template <int R, int U>
__global__ void gemv_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                float* __restrict__ y, int M, int K, float scale, int chunks,
                                int rows_per_block KX_BPARAM(3)) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs = reinterpret_cast<f32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int m4 = M >> 2;
  for (int i = tid; i < chunks * 64; i += blockDim.x) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < m4) v = *KX_CKP(0, reinterpret_cast<const f32x4*>(x) + i);
    xs[i] = v;
  }
  __syncthreads();
  const int row_begin = blockIdx.x * rows_per_block;
  const int row_end = min(row_begin + rows_per_block, K);
  for (int r0 = row_begin + wave * R; r0 < row_end; r0 += nwaves * R) {
    float total[R];
#pragma unroll
    for (int r = 0; r < R; ++r) total[r] = 0.f;
    for (int c0 = 0; c0 < chunks; c0 += U) {
      f32x4 wv[R][U];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int row = min(r0 + r, row_end - 1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i4 = (c0 + u) * 64 + lane;
          wv[r][u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (i4 < m4)
            wv[r][u] = __builtin_nontemporal_load(
                KX_CKP(1, reinterpret_cast<const f32x4*>(w + (size_t)row * (size_t)M) + i4));
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (c0 + u < chunks) {
          const f32x4 xv = xs[(c0 + u) * 64 + lane];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            total[r] = fmaf(xv.x, wv[r][u].x, total[r]);
            total[r] = fmaf(xv.y, wv[r][u].y, total[r]);
            total[r] = fmaf(xv.z, wv[r][u].z, total[r]);
            total[r] = fmaf(xv.w, wv[r][u].w, total[r]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float v = wave_sum(total[r]);
      if (lane == 0 && r0 + r < row_end) *KX_CKP(2, y + (r0 + r)) = v * scale;
    }
  }
}

// Any M, any alignment: one wave per row, scalar loads.
__global__ void gemv_f32_generic_kernel(const float* x, const float* w, float* y, int M, int K,
                                        float scale KX_BPARAM(3)) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= K) return;
  float total = 0.f;
  for (int i = lane; i < M; i += 64) total = fmaf(*KX_CKP(0, x + i), *KX_CKP(1, w + ((size_t)row * M + i)), total);
  total = wave_sum(total);
  if (lane == 0) *KX_CKP(2, y + row) = total * scale;
}

}  // namespace kx

extern "C" int kx_gemv_f32(const float* x, const float* w, float* y, int32_t M, int32_t K,
                           float scale, void* stream) {
  using namespace kx;
  if (!x || !w || !y || M <= 0 || K <= 0) return KX_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool fast = M % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) &&
                    (size_t)M * 4 <= 64 * 1024;
  if (!fast) {
    hipLaunchKernelGGL(gemv_f32_generic_kernel, dim3((K + 3) / 4), dim3(256), 0, s, x, w, y, M, K,
                       scale KX_BPASS(3, x, w, y));
    KX_LAUNCH_CHECK();
    return KX_OK;
  }
  const int chunks = (M + 255) / 256;
  const size_t lds = (size_t)chunks * 1024;
  const int threads = lds > 32 * 1024 ? 512 : 256;
  const int waves = threads / 64;
  constexpr int R = 2, U = 4;
  const int per_iter = waves * R;
  int rpb = (K + 1023) / 1024;
  rpb = ((rpb + per_iter - 1) / per_iter) * per_iter;
  const int grid = (K + rpb - 1) / rpb;
  hipLaunchKernelGGL((gemv_f32_kernel<R, U>), dim3(grid), dim3(threads), lds, s, x, w, y, M, K,
                     scale, chunks, rpb KX_BPASS(3, x, w, y));
  KX_LAUNCH_CHECK();
  return KX_OK;
}
