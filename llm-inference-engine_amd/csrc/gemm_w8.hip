// W8A16 group-quant GEMM for gfx950 (prefill path):  Y[T][K] = X[T][M] . W[K][M]^T
// with the per-row formula of the decode GEMV,
//   Y[t][p] = sum_i X[t][i] * scales[(p*M+i)/gs] * (float)w[p*M+i].
//
// The reference has no batched matmul at all: its prompt phase feeds one token per launch
// (demo/main.cpp:18-41, cuda/matmul_kernel.cu:89-134 ignore a second input dim).  This kernel is
// the new capability behind the same MatmulKernelQuant registry entry.
//
// Layout / numerics
//   - 128 x 128 output tile per 256-thread workgroup (4 waves, 2 x 2, 64 x 64 each),
//     K-step = 64 = one quant group, so every K-step has exactly one scale per output column.
//   - activations are rounded to fp16 while they are staged into LDS; int8 weights are exact in
//     fp16.  v_mfma_f32_32x32x16_f16 accumulates each group's 64 products in fp32; the group sum
//     is then multiplied by its fp32 scale and added to the running fp32 total in VGPRs.
//     The only deviation from the fp32 reference formula is the fp16 rounding of X.
//   - LDS tiles are [row][64 halves] with rows padded to 144 bytes: the 16-byte fragment reads of
//     a 16-lane group then fall on distinct banks.  Two LDS buffers, one barrier per K-step, the
//     next tile's global loads are issued before the MFMAs of the current one.
#define KX_FILE_ID 3
#include "kx_common.hpp"

namespace kx {

constexpr int GM_BM = 128, GM_BN = 128, GM_BK = 64;
constexpr int GM_LDS_ROW = 72;  // halves per padded LDS row (144 bytes)

struct GemmW8Args {
  const float* X;      // [T][ldx]
  const int8_t* w;     // [K][M]
  const float* scales; // [K][M/64]
  const float* residual;  // nullable [T][ldy0-like]: added to destination 0 only
  float* y0; float* y1; float* y2;   // column ranges [0,k0) [k0,k1) [k1,K)
  int64_t ldx, ld0, ld1, ld2, ldr;
  int32_t k0, k1;
  int32_t T, M, K;
  KX_BFIELD(7)  // operands of the bounds-checking build, in the order of GemmW8Operand
};
enum GemmW8Operand { MB_X, MB_W, MB_S, MB_RES, MB_Y0, MB_Y1, MB_Y2 };

__device__ __forceinline__ f16x8 cvt_i8x8(int32_t lo, int32_t hi) {
  f16x8 r;
  r[0] = (_Float16)(float)(int8_t)(lo);
  r[1] = (_Float16)(float)(int8_t)(lo >> 8);
  r[2] = (_Float16)(float)(int8_t)(lo >> 16);
  r[3] = (_Float16)(float)(int8_t)(lo >> 24);
  r[4] = (_Float16)(float)(int8_t)(hi);
  r[5] = (_Float16)(float)(int8_t)(hi >> 8);
  r[6] = (_Float16)(float)(int8_t)(hi >> 16);
  r[7] = (_Float16)(float)(int8_t)(hi >> 24);
  return r;
}

__global__ __launch_bounds__(256, 2) void gemm_w8_kernel(const GemmW8Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // per buffer: A tile, B tile, 128 scales
  constexpr int TILE_HALVES = GM_BM * GM_LDS_ROW;
  constexpr int BUF_BYTES = 2 * TILE_HALVES * 2 + GM_BN * 4;
  auto lds_a = [&](int buf) { return reinterpret_cast<_Float16*>(smem + buf * BUF_BYTES); };
  auto lds_b = [&](int buf) { return lds_a(buf) + TILE_HALVES; };
  auto lds_s = [&](int buf) { return reinterpret_cast<float*>(lds_b(buf) + TILE_HALVES); };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // consecutive workgroups walk down the token dimension so they share one W column panel
  const int tiles_m = (a.T + GM_BM - 1) / GM_BM;
  const int tile_m = blockIdx.x % tiles_m, tile_n = blockIdx.x / tiles_m;
  const int t0 = tile_m * GM_BM, n0 = tile_n * GM_BN;
  const int groups = a.M / GM_BK;

  // ---- staging roles ----
  // X: thread -> rows (tid>>4) + 16*i, float4 column (tid&15)
  const int xr = tid >> 4, xc = (tid & 15) * 4;
  // W: thread -> rows (tid>>2) + 64*i, 16-byte column chunk (tid&3)
  const int wrow = tid >> 2, wchunk = (tid & 3) * 16;

  f32x4 xreg[8];
  i32x4 wreg[2];
  float sreg = 0.f;

  auto load_tile = [&](int g) {
    const int kk = g * GM_BK;
    // rows past the end re-read the last valid row (their outputs are never stored): unconditional
    // loads keep the whole batch in flight -- a load under a runtime condition makes the compiler
    // branch around it and drain the counter (270 -> 237 us on 256 x 4096 x 14336)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = min(t0 + xr + 16 * i, a.T - 1);
      xreg[i] = *KX_CK(a, MB_X, reinterpret_cast<const f32x4*>(a.X + (size_t)t * a.ldx + kk + xc));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = min(n0 + wrow + 64 * i, a.K - 1);
      wreg[i] = *KX_CK(a, MB_W, reinterpret_cast<const i32x4*>(a.w + (size_t)n * a.M + kk + wchunk));
    }
    if (tid < GM_BN) sreg = *KX_CK(a, MB_S, a.scales + ((size_t)min(n0 + tid, a.K - 1) * groups + g));
  };
  auto store_tile = [&](int buf) {
    _Float16* A = lds_a(buf);
    _Float16* B = lds_b(buf);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f16x4 h;
      h[0] = (_Float16)xreg[i].x;
      h[1] = (_Float16)xreg[i].y;
      h[2] = (_Float16)xreg[i].z;
      h[3] = (_Float16)xreg[i].w;
      *reinterpret_cast<f16x4*>(A + (xr + 16 * i) * GM_LDS_ROW + xc) = h;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      _Float16* dst = B + (wrow + 64 * i) * GM_LDS_ROW + wchunk;
      *reinterpret_cast<f16x8*>(dst) = cvt_i8x8(wreg[i].x, wreg[i].y);
      *reinterpret_cast<f16x8*>(dst + 8) = cvt_i8x8(wreg[i].z, wreg[i].w);
    }
    if (tid < GM_BN) lds_s(buf)[tid] = sreg;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int frow = lane & 31, fk = (lane >> 5) * 8;
  for (int g = 0; g < groups; ++g) {
    const int buf = g & 1;
    if (g + 1 < groups) load_tile(g + 1);

    const _Float16* A = lds_a(buf) + (wr * 64 + frow) * GM_LDS_ROW + fk;
    const _Float16* B = lds_b(buf) + (wc * 64 + frow) * GM_LDS_ROW + fk;
    f32x16 grp[2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f16x8 af[2], bf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        af[mi] = *reinterpret_cast<const f16x8*>(A + mi * 32 * GM_LDS_ROW + ks * 16);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        bf[ni] = *reinterpret_cast<const f16x8*>(B + ni * 32 * GM_LDS_ROW + ks * 16);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if (ks == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            grp[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi], bf[ni], z, 0, 0, 0);
          } else {
            grp[mi][ni] =
                __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi], bf[ni], grp[mi][ni], 0, 0, 0);
          }
        }
    }
    // group scale: the C layout keeps the output column on the lane
    const float s0 = lds_s(buf)[wc * 64 + frow];
    const float s1 = lds_s(buf)[wc * 64 + 32 + frow];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[mi][0][r] = fmaf(grp[mi][0][r], s0, acc[mi][0][r]);
        acc[mi][1][r] = fmaf(grp[mi][1][r], s1, acc[mi][1][r]);
      }

    if (g + 1 < groups) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wc * 64 + ni * 32 + frow;
    if (n >= a.K) continue;
    float* dst;
    int64_t ld;
    int col;
    [[maybe_unused]] int op;
    bool add_res = false;
    if (n < a.k0) {
      dst = a.y0; ld = a.ld0; col = n; add_res = a.residual != nullptr; op = MB_Y0;
    } else if (n < a.k1) {
      dst = a.y1; ld = a.ld1; col = n - a.k0; op = MB_Y1;
    } else {
      dst = a.y2; ld = a.ld2; col = n - a.k1; op = MB_Y2;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = t0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (t < a.T) {
          float v = acc[mi][ni][r];
          if (add_res) v += *KX_CK(a, MB_RES, a.residual + ((size_t)t * a.ldr + col));
          *KX_CK(a, op, dst + ((size_t)t * ld + col)) = v;
        }
      }
  }
}

int gemm_w8_launch(const GemmW8Args& a, hipStream_t s) {
  const int tiles_m = (a.T + GM_BM - 1) / GM_BM, tiles_n = (a.K + GM_BN - 1) / GM_BN;
  constexpr size_t lds = 2 * (2 * GM_BM * GM_LDS_ROW * 2 + GM_BN * 4);
  static OncePerDevice raised;
  if (int rc = raised.raise(reinterpret_cast<const void*>(&gemm_w8_kernel), (int)lds)) return rc;
  hipLaunchKernelGGL(gemm_w8_kernel, dim3(tiles_m * tiles_n), dim3(256), lds, s, a);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

}  // namespace kx

extern "C" int kx_gemm_w8_f32_ex(const float* X, int64_t ldx, const int8_t* w, const float* scales,
                                 int32_t group_size, float* y0, int64_t ld0, float* y1,
                                 int64_t ld1, float* y2, int64_t ld2, int32_t k0, int32_t k1,
                                 const float* residual, int64_t ldr, int32_t T, int32_t M,
                                 int32_t K, void* stream) {
  using namespace kx;
  if (!X || !w || !scales || !y0 || T <= 0 || M <= 0 || K <= 0 || group_size <= 0) return KX_EINVAL;
  if (!y1 && !y2) { k0 = K; k1 = K; }
  if (k0 < 0 || k1 < k0 || k1 > K) return KX_EINVAL;
  if ((k0 < K && !y1 && k1 > k0) || (k1 < K && !y2)) return KX_EINVAL;
  const bool mfma_ok = group_size == GM_BK && M % GM_BK == 0 && ldx % 4 == 0 &&
                       ((uintptr_t)X % 16 == 0) && ((uintptr_t)w % 16 == 0);
  if (!mfma_ok || T < 4) {
    // shapes the tiled kernel is not built for: one GEMV per row, same results as the decode path
    for (int32_t t = 0; t < T; ++t) {
      int rc = kx_gemv_w8_f32_ex(X + (size_t)t * ldx, w, scales, group_size, y0 + (size_t)t * ld0,
                                 y1 ? y1 + (size_t)t * ld1 : nullptr,
                                 y2 ? y2 + (size_t)t * ld2 : nullptr, k0, k1, nullptr,
                                 residual ? residual + (size_t)t * ldr : nullptr, M, K, stream);
      if (rc) return rc;
    }
    return KX_OK;
  }
  GemmW8Args a{};
  a.X = X; a.w = w; a.scales = scales; a.residual = residual;
  a.y0 = y0; a.y1 = y1 ? y1 : y0; a.y2 = y2 ? y2 : y0;
  a.ldx = ldx; a.ld0 = ld0; a.ld1 = ld1; a.ld2 = ld2; a.ldr = ldr;
  a.k0 = k0; a.k1 = k1; a.T = T; a.M = M; a.K = K;
  KX_BSET(a, MB_X, a.X); KX_BSET(a, MB_W, a.w); KX_BSET(a, MB_S, a.scales); KX_BSET(a, MB_RES, a.residual);
  KX_BSET(a, MB_Y0, a.y0); KX_BSET(a, MB_Y1, a.y1); KX_BSET(a, MB_Y2, a.y2);
  return gemm_w8_launch(a, (hipStream_t)stream);
}

extern "C" int kx_gemm_w8_f32(const float* X, const int8_t* w, const float* scales,
                              int32_t group_size, float* Y, int32_t T, int32_t M, int32_t K,
                              void* stream) {
  return kx_gemm_w8_f32_ex(X, M, w, scales, group_size, Y, K, nullptr, 0, nullptr, 0, K, K, nullptr,
                           0, T, M, K, stream);
}
