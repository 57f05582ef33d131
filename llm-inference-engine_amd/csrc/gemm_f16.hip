// Two-pass W8A16 GEMM for long prompts, gfx950:  Y[T][K] = X[T][M] . W[K][M]^T  with
//   Y[t][p] = sum_i X[t][i] * scales[(p*M+i)/gs] * (float)w[p*M+i]
// (the per-row formula of matmul_kernel_cu_fp32int8, kuiper/source/op/kernels/cuda/
// matmul_kernel.cu:56-87, applied to T rows; the reference itself has no batched matmul).
//
// The single-pass kernel in gemm_w8.hip converts int8 -> fp16 and applies the group scales inside
// its main loop, once per 128-row tile of X: with thousands of rows that vector work, not the
// matrix pipe, bounds it.  Here the conversions are done ONCE per call into a caller-provided
// workspace, and the main loop is a plain fp16 MFMA GEMM:
//   pass 1a  X fp32 -> fp16
//   pass 1b  W int8 * group scale / f[p] -> fp16, where f[p] is a power of two per output row p
//            that brings the row's largest scaled weight to <= 1 (keeps small scales out of the
//            fp16 subnormal range); f is kept in fp32 and multiplied back in the epilogue
//   pass 2   Y = (X16 . W16^T) * f, fp32 accumulation on v_mfma_f32_32x32x16_f16, tiles staged by
//            LDS-DMA, with the routing / residual epilogue of the single-pass kernel
// Deviation from the fp32 formula: one fp16 rounding of X and one of (scale * w), each 2^-11
// relative, independent per element.
#include <math.h>
#include <stdlib.h>

#define KX_FILE_ID 4
#include "kx_common.hpp"

namespace kx {

constexpr int GF_BK = 64;

struct GemmF16Args {
  const _Float16* X;   // [T][M]
  const _Float16* W;   // [K][M]
  const float* colf;   // [K] power-of-two row factors
  const float* residual;
  _Float16* h16;       // SwiGLU form: fp16 [T][K/2] output
  float* y0; float* y1; float* y2;
  int64_t ld0, ld1, ld2, ldr;
  int32_t k0, k1;
  int32_t T, M, K;
  // pipelined kernel only: tiles [full_tiles, tiles) have their reduction cut into `split` parts,
  // one workgroup each, combined through fp32 slabs and one arrival ticket per tile
  float* slabs;
  unsigned* tickets;
  int32_t full_tiles, split;
  // the parts only write their slabs; gemm_split_combine_kernel, launched behind this one, adds
  // them and writes the tiles (plain / residual form, reductions cut four ways or more)
  int32_t combine_later;
  int32_t wide_rows;  // every destination row is 16-byte aligned: the epilogue may use 16-byte accesses
  // Wq|Wk|Wv form (pipelined kernel, wide rows): bias (all K columns, or null), then the rotary
  // embedding of the q columns [0, k0) and the k columns [k0, k1) at position pos0 + row
  const float* bias;
  const float* sin_cache;  // [position][head_size], the entry of a pair at its even index
  const float* cos_cache;
  int32_t rope_variant, head_size, pos0;  // rope_variant < 0: bias only
  KX_BFIELD(13)  // operands of the bounds-checking build, in the order of GemmF16Operand
};
enum GemmF16Operand { FB_X, FB_W, FB_COLF, FB_RES, FB_H16, FB_Y0, FB_Y1, FB_Y2, FB_SLABS, FB_TICKETS, FB_BIAS, FB_SIN, FB_COS };

// X fp32 [T][ldx] -> fp16 [T][ldy], M columns
__global__ void f32_to_f16_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                       _Float16* __restrict__ y, int64_t ldy, int M, int64_t n4 KX_BPARAM(2)) {
  const int m4 = M / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / m4;
    const int c = (int)(i - t * m4) * 4;
    const f32x4 v = *KX_CKP(0, reinterpret_cast<const f32x4*>(x + t * ldx + c));
    f16x4 h;
    h[0] = sat_f16(v.x); h[1] = sat_f16(v.y); h[2] = sat_f16(v.z); h[3] = sat_f16(v.w);
    *KX_CKP(1, reinterpret_cast<f16x4*>(y + t * ldy + c)) = h;
  }
}

// One wave per weight row: W16[p][i] = fp16(w[p][i] * s[p][i/gs] / f[p]).
__global__ __launch_bounds__(256) void w8_dequant_rows_kernel(const int8_t* __restrict__ w,
                                                              const float* __restrict__ scales,
                                                              _Float16* __restrict__ w16,
                                                              float* __restrict__ colf, int M, int K,
                                                              int gs_shift, int swiglu_half KX_BPARAM(4)) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= K) return;
  // SwiGLU form (W = [gate rows | up rows], swiglu_half rows each): blocks of 32 gate rows and
  // the matching 32 up rows are stored next to each other, so that one wave of the GEMM holds
  // gate and up of the same hidden units in the same lane and register
  int q = p;
  if (swiglu_half > 0) {
    const int j = p < swiglu_half ? p : p - swiglu_half;
    q = (j >> 5) * 64 + (p < swiglu_half ? 0 : 32) + (j & 31);
  }
  const int groups = M >> gs_shift;
  const float* sp = scales + (size_t)p * groups;
  float mx = 0.f;
  for (int g = lane; g < groups; g += 64) mx = fmaxf(mx, fabsf(*KX_CKP(1, sp + g)));
  mx = wave_max(mx) * 127.0f;
  // smallest power of two >= the row's largest scaled weight (1 for an all-zero row)
  const float f = mx > 0.f ? exp2f(ceilf(log2f(mx))) : 1.0f;
  const float inv = 1.0f / f;
  if (lane == 0) *KX_CKP(3, colf + q) = f;
  const int8_t* wp = w + (size_t)p * M;
  _Float16* op = w16 + (size_t)q * M;
  for (int c = lane * 16; c < M; c += 1024) {
    const i32x4 v = __builtin_nontemporal_load(KX_CKP(0, reinterpret_cast<const i32x4*>(wp + c)));
    const float s = *KX_CKP(1, sp + (c >> gs_shift)) * inv;
    const int32_t words[4] = {v.x, v.y, v.z, v.w};
    f16x8 lo, hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo[e] = (_Float16)((float)(int8_t)(words[0] >> (8 * e)) * s);
      lo[4 + e] = (_Float16)((float)(int8_t)(words[1] >> (8 * e)) * s);
      hi[e] = (_Float16)((float)(int8_t)(words[2] >> (8 * e)) * s);
      hi[4 + e] = (_Float16)((float)(int8_t)(words[3] >> (8 * e)) * s);
    }
    *KX_CKP(2, reinterpret_cast<f16x8*>(op + c)) = lo;
    *KX_CKP(2, reinterpret_cast<f16x8*>(op + c + 8)) = hi;
  }
}

// silu(g) * u for an fp16 result (swiglu_kernel_cu, cuda/swiglu_kernel.cu:4-22): hardware exp2 and
// reciprocal, each good to about 2^-22 relative -- far inside the 2^-11 rounding of the fp16 store;
// the library expf and the IEEE division they replace took 8 of the 11 us a round of W1|W3 tiles
// spent in its epilogue at 4096 rows.
__device__ __forceinline__ float silu_mul_fast(float g, float u) {
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * g);  // inf for very negative g: rcp -> 0
  return g * __builtin_amdgcn_rcpf(1.0f + e) * u;
}

// Epilogue of one wave: acc[mi][ni] is the 32 x 32 block at rows trow0 + 32 mi, columns ncol0 + 32 ni;
// C layout of v_mfma_f32_32x32x16_f16: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
// Multiplies the power-of-two row factor back in and either routes columns to up to three
// destinations (+ residual on the first), or (SWIGLU) writes silu(gate) * up as fp16.
template <int MI, int NI, bool SWIGLU>
__device__ __forceinline__ void gemm_f16_epilogue(const GemmF16Args& a, const f32x16 (&acc)[MI][NI],
                                                  int trow0, int ncol0, int lane) {
  const int frow = lane & 31;
  if constexpr (SWIGLU) {
    static_assert(NI % 2 == 0, "gate/up blocks come in pairs");
    const int half_k = a.K / 2;
#pragma unroll
    for (int ni = 0; ni < NI; ni += 2) {
      const int n = ncol0 + ni * 32 + frow;  // gate row in the interleaved order
      if (n >= a.K) continue;
      const int j = (n >> 6) * 32 + frow;                // hidden unit
      const float fg = *KX_CK(a, FB_COLF, a.colf + n), fu = *KX_CK(a, FB_COLF, a.colf + (n + 32));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = trow0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (t < a.T) {
            const float g = acc[mi][ni][r] * fg, u = acc[mi][ni + 1][r] * fu;
            *KX_CK(a, FB_H16, a.h16 + ((size_t)t * half_k + j)) = sat_f16(silu_mul_fast(g, u));
          }
        }
    }
    return;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = ncol0 + ni * 32 + frow;
    if (n >= a.K) continue;
    const float f = *KX_CK(a, FB_COLF, a.colf + n);
    float* dst;
    int64_t ld;
    int col;
    [[maybe_unused]] int op;
    bool add_res = false;
    if (n < a.k0) {
      dst = a.y0; ld = a.ld0; col = n; add_res = a.residual != nullptr; op = FB_Y0;
    } else if (n < a.k1) {
      dst = a.y1; ld = a.ld1; col = n - a.k0; op = FB_Y1;
    } else {
      dst = a.y2; ld = a.ld2; col = n - a.k1; op = FB_Y2;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = trow0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (t < a.T) {
          float v = acc[mi][ni][r] * f;
          if (add_res) v += *KX_CK(a, FB_RES, a.residual + ((size_t)t * a.ldr + col));
          *KX_CK(a, op, dst + ((size_t)t * ld + col)) = v;
        }
      }
  }
}

// The same epilogue for accumulators of v_mfma_f32_16x16x32_f16: acc[mi][ni] is the 16 x 16 block at
// rows trow0 + 16 mi, columns ncol0 + 16 ni; col = lane & 15, row = 4 (lane >> 4) + r.  SWIGLU pairs
// block ni (gate) with block ni + 2 (up): 32 gate columns are followed by their 32 up columns.
template <int MI, int NI, bool SWIGLU>
__device__ __forceinline__ void gemm_f16_epilogue16(const GemmF16Args& a, const f32x4 (&acc)[MI][NI],
                                                    int trow0, int ncol0, int lane) {
  const int fcol = lane & 15, frow4 = (lane >> 4) * 4;
  if constexpr (SWIGLU) {
    static_assert(NI % 4 == 0, "gate/up blocks come in groups of 2 + 2");
    const int half_k = a.K / 2;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if ((ni & 2) != 0) continue;
      const int n = ncol0 + ni * 16 + fcol;  // gate row in the interleaved order
      if (n >= a.K) continue;
      const int j = (n >> 6) * 32 + (n & 31);  // hidden unit
      const float fg = *KX_CK(a, FB_COLF, a.colf + n), fu = *KX_CK(a, FB_COLF, a.colf + (n + 32));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = trow0 + mi * 16 + frow4 + r;
          if (t < a.T) {
            const float g = acc[mi][ni][r] * fg, u = acc[mi][ni + 2][r] * fu;
            *KX_CK(a, FB_H16, a.h16 + ((size_t)t * half_k + j)) = sat_f16(silu_mul_fast(g, u));
          }
        }
    }
    return;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = ncol0 + ni * 16 + fcol;
    if (n >= a.K) continue;
    const float f = *KX_CK(a, FB_COLF, a.colf + n);
    float* dst;
    int64_t ld;
    int col;
    [[maybe_unused]] int op;
    bool add_res = false;
    if (n < a.k0) {
      dst = a.y0; ld = a.ld0; col = n; add_res = a.residual != nullptr; op = FB_Y0;
    } else if (n < a.k1) {
      dst = a.y1; ld = a.ld1; col = n - a.k0; op = FB_Y1;
    } else {
      dst = a.y2; ld = a.ld2; col = n - a.k1; op = FB_Y2;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = trow0 + mi * 16 + frow4 + r;
        if (t < a.T) {
          float v = acc[mi][ni][r] * f;
          if (add_res) v += *KX_CK(a, FB_RES, a.residual + ((size_t)t * a.ldr + col));
          *KX_CK(a, op, dst + ((size_t)t * ld + col)) = v;
        }
      }
  }
}

// (64 MI) x (64 NI) output tile per 256-thread workgroup (4 waves as 2 x 2, each MI x NI
// accumulator tiles of 32 x 32), K-step 64, operands brought in by the LDS-DMA path
// (global_load_lds, 16 bytes per lane): no staging registers, no ds_write pass.  One buffer, two
// barriers per K-step; several
// workgroups per CU overlap each other's load and MFMA phases.  A wave instruction fills 1 KiB
// of LDS linearly (8 rows of 128 bytes), so rows cannot be padded: the 16-byte chunk c of row r
// is kept at chunk position c ^ ((r >> 1) & 7) instead (applied to the SOURCE address when
// filling and to the LDS address when reading), which spreads any 16 consecutive rows of one
// fragment column over all 64 banks.
// SWIGLU: W16 holds gate/up rows interleaved in blocks of 32 (see the dequant kernel); the
// epilogue writes silu(gate) * up as fp16 (swiglu_kernel_cu, cuda/swiglu_kernel.cu:4-22).
template <int MI, int NI, bool SWIGLU>
__device__ __forceinline__ void gemm_f16_body(const GemmF16Args& a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  using lds_ptr_t = __attribute__((address_space(3))) void*;
  constexpr int BM = 64 * MI, BN = 64 * NI;
  constexpr int A_BYTES = BM * GF_BK * 2;
  constexpr int AI = BM / 32, BI = BN / 32;  // 1 KiB fill instructions per wave and operand
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // consecutive workgroups walk down the token dimension so they share one W column panel
  const int tiles_m = (a.T + BM - 1) / BM;
  const int tile_m = blockIdx.x % tiles_m, tile_n = blockIdx.x / tiles_m;
  const int t0 = tile_m * BM, n0 = tile_n * BN;
  const int ksteps = a.M / GF_BK;

  // fill roles: instruction i of wave w covers tile rows 8*(4i + w) .. +8; lane -> row l>>3,
  // chunk position l&7, source chunk (l&7) ^ swizzle(row)
  const _Float16* asrc[AI];
  const _Float16* bsrc[BI];
  {
    const int r8 = lane >> 3;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = 8 * (4 * i + wave) + r8;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      // rows past the end re-read the last valid row; their outputs are never stored
      asrc[i] = a.X + (size_t)min(t0 + row, a.T - 1) * a.M + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = 8 * (4 * i + wave) + r8;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      bsrc[i] = a.W + (size_t)min(n0 + row, a.K - 1) * a.M + chunk * 8;
    }
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int frow = lane & 31, khalf = lane >> 5;
  // byte offset of this lane's first fragment row inside each operand tile; fragment rows of one
  // lane are 32 rows apart, which leaves the swizzle key (row >> 1) & 7 unchanged
  const int ra = wr * 32 * MI + frow, rb = wc * 32 * NI + frow;
  const int aoff = ra * 128, boff = A_BYTES + rb * 128;
  const int akey = (ra >> 1) & 7, bkey = (rb >> 1) & 7;

  for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
    for (int i = 0; i < AI; ++i)
      __builtin_amdgcn_global_load_lds(KX_CKN(a, FB_X, asrc[i] + ks * GF_BK, 16),
                                       (lds_ptr_t)(smem + (4 * i + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < BI; ++i)
      __builtin_amdgcn_global_load_lds(KX_CKN(a, FB_W, bsrc[i] + ks * GF_BK, 16),
                                       (lds_ptr_t)(smem + A_BYTES + (4 * i + wave) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      const int c = kq * 2 + khalf;
      f16x8 af[MI], bf[NI];
#pragma unroll
      for (int m = 0; m < MI; ++m)
        af[m] = *reinterpret_cast<const f16x8*>(smem + aoff + m * 32 * 128 + ((c ^ akey) << 4));
#pragma unroll
      for (int n = 0; n < NI; ++n)
        bf[n] = *reinterpret_cast<const f16x8*>(smem + boff + n * 32 * 128 + ((c ^ bkey) << 4));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

  gemm_f16_epilogue<MI, NI, SWIGLU>(a, acc, t0 + wr * 32 * MI, n0 + wc * 32 * NI, lane);
}

// the second launch bound is the number of 256-thread workgroups per CU the registers must allow
__global__ __launch_bounds__(256, 4) void gemm_f16_kernel_22(const GemmF16Args a) { gemm_f16_body<2, 2, false>(a); }
__global__ __launch_bounds__(256, 4) void gemm_f16_swiglu_kernel_22(const GemmF16Args a) { gemm_f16_body<2, 2, true>(a); }

// ---------------------------------------------------------------------------------------------
// 256 x 256 output tile, 512 threads = 8 waves as 2 (rows) x 4 (columns), wave tile 128 x 64 as
// 8 x 4 blocks of v_mfma_f32_16x16x32_f16 (on random data the chip holds a higher clock on this
// shape than on 32x32x16: 975 -> 855 us for 4096 x 28672 x 4096 with nothing else changed),
// K-tiles of 64, one workgroup per CU, the whole 128 KiB of two K-tile buffers in one LDS array.
// Unlike the kernels above, the LDS-DMA loads stay in flight ACROSS barriers: a K-tile is brought
// in as four 16 KiB pieces, one piece per phase, each issued six to three phases before its
// first read, into the buffer region whose previous contents were last read three phases
// earlier.  Pieces (rows of the operand tile; every wave reads 1/2 of one A piece pair and 1/4 of
// one B piece pair):
//   0  B rows with bit 5 clear  (columns 0-31 of every wave)    read in phase 3 of the previous tile
//   1  A rows with bit 6 clear  (rows 0-63 of every wave)       read in phase 0
//   2  B rows with bit 5 set    (columns 32-63)                 read in phase 1
//   3  A rows with bit 6 set    (rows 64-127)                   read in phase 2
// Phase g = 4 t + p issues piece (g + 6) % 4 of tile (g + 6) / 4, then waits `vmcnt(8)`: every
// wave has two loads per piece, so all but the newest four pieces have landed -- the piece issued
// in phase g is complete, for the issuing wave, at the wait of phase g + 4, and readable by all
// waves from phase g + 5 (after a barrier the issuing wave passed later).  A phase is
//   [LDS fragment reads | issue | wait]  barrier  [16 MFMAs]  barrier
// and waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave's reads overlap the
// other's MFMAs.  Each phase multiplies a 64 x 32 quadrant of the wave tile by the full K-tile;
// the two B register sets swap roles every tile.  Tiles past the end re-load the last K-tile
// into dead regions, which keeps the counts uniform.
constexpr int GP_BUF = 65536, GP_BOFF = 32768;
constexpr int GP_GROUP_N = 4;
using gp_lds_ptr_t = __attribute__((address_space(3))) void*;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GemmPipeCtx {
  const _Float16* src[4][2];  // per piece: this lane's source of its two fill instructions
  int dst[4][2];              // and their (wave-uniform) byte offsets inside a buffer
  int abase, bbase;           // byte offset of this lane's fragment row in the A / B region
  int coff[2];                // swizzled 16-byte chunk offset per k-step of 32
  int nt;
  bool live_lo, live_hi;      // rows 0-63 / 64-127 of this wave's tile hold real rows (short prompts)
};

template <int PIECE, int BUFSEL>
__device__ __forceinline__ void gp_issue(const GemmF16Args& a, char* smem, const GemmPipeCtx& c, int tile) {
  const int koff = min(tile, c.nt - 1) * GF_BK;
#pragma unroll
  for (int j = 0; j < 2; ++j)
    __builtin_amdgcn_global_load_lds(KX_CKN(a, (PIECE & 1) ? FB_X : FB_W, c.src[PIECE][j] + koff, 16),
                                     (gp_lds_ptr_t)(smem + BUFSEL * GP_BUF + c.dst[PIECE][j]), 16, 0, 0);
}

// fragments of v_mfma_f32_16x16x32_f16: BLOCKS consecutive 16-row blocks x two k-steps of 32
template <int BUFSEL, int BLOCK, int BLOCKS>
__device__ __forceinline__ void gp_read16(const char* smem, int base, const int (&coff)[2],
                                          f16x8 (&f)[BLOCKS][2]) {
#pragma unroll
  for (int b = 0; b < BLOCKS; ++b)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      f[b][kk] = *reinterpret_cast<const f16x8*>(smem + BUFSEL * GP_BUF + (BLOCK + b) * 2048 + base + coff[kk]);
}

// One phase: issue a piece, wait until all but the newest four pieces have landed, barrier, the
// quadrant's 16 MFMAs (row blocks MI0 .. MI0 + 3 x column blocks NI0, NI0 + 1), barrier.  The
// priority pair keeps the compiler from moving MFMAs across the barriers in among the loads.
#define GP_PHASE(PIECE, BUFSEL, TILE, MI0, NI0, BREG, LIVE)                                             \
  gp_issue<PIECE, BUFSEL>(a, smem, c, TILE);                                                            \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                    \
  __builtin_amdgcn_s_barrier();                                                                         \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                    \
  __builtin_amdgcn_s_setprio(1);                                                                        \
  if (LIVE) {                                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                    \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                     \
        _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                   \
          acc[MI0 + m][NI0 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][kk], BREG[n][kk],        \
                                                                         acc[MI0 + m][NI0 + n], 0, 0, 0); \
  }                                                                                                     \
  __builtin_amdgcn_s_setprio(0);                                                                        \
  __builtin_amdgcn_sched_barrier(0);                                                                    \
  __builtin_amdgcn_s_barrier();                                                                         \
  __builtin_amdgcn_sched_barrier(0);

// One K-tile (four phases).  PAR = t & 1 selects the buffer; bq0 holds columns 0-31 of this tile on
// entry, bq1 receives columns 32-63 in phase 1 and columns 0-31 of tile t + 1 in phase 3.
// SHORT (prompts of at most 128 rows: one row tile, most of it padding): waves whose 64-row
// halves hold no real row skip the fragment reads and the MFMAs of those halves -- the matrix pipe
// stops setting the pace of a K-tile step, the weight stream does.
template <int PAR, bool SHORT>
__device__ __forceinline__ void gp_tile(const GemmF16Args& a, char* smem, const GemmPipeCtx& c, int t, f32x4 (&acc)[8][4],
                                        f16x8 (&af)[4][2], f16x8 (&bq0)[2][2], f16x8 (&bq1)[2][2]) {
  const bool lo = SHORT ? c.live_lo : true, hi = SHORT ? c.live_hi : true;
  // phase 0: rows 0-63 of the wave tile x columns 0-31
  if (lo) gp_read16<PAR, 0, 4>(smem, c.abase, c.coff, af);
  GP_PHASE(2, PAR ^ 1, t + 1, 0, 0, bq0, lo);
  // phase 1: rows 0-63 x columns 32-63
  gp_read16<PAR, 2, 2>(smem, c.bbase, c.coff, bq1);
  GP_PHASE(3, PAR ^ 1, t + 1, 0, 2, bq1, lo);
  // phase 2: rows 64-127 x columns 32-63
  if (hi) gp_read16<PAR, 4, 4>(smem, c.abase, c.coff, af);
  GP_PHASE(0, PAR, t + 2, 4, 2, bq1, hi);
  // phase 3: rows 64-127 x columns 0-31; columns 0-31 of the next tile replace columns 32-63
  gp_read16<PAR ^ 1, 0, 2>(smem, c.bbase, c.coff, bq1);
  GP_PHASE(1, PAR, t + 2, 4, 0, bq0, hi);
}

// Epilogue of the pipelined kernel through LDS (free once the main loop is over): a wave turns
// its 128 x 64 accumulator tile, 64 rows at a time, from "lane = column, 4 rows per register
// quad" into "16 lanes = one row of 64 columns", so that results leave as 16-byte stores in
// 256-byte row segments (and the residual arrives the same way) instead of 128 dword stores per
// lane in 64-byte pieces -- measured on 4096 rows: the dword form took 135 of the 317 us of the
// Wq|Wk|Wv product and 57 of the 162 us of Wo.  Each wave uses its own 16 KiB of LDS, so no
// workgroup barrier is needed between its writes and reads.  The 16-column blocks of a row are
// stored XOR-ed with the row's quad index, which spreads a write instruction over all banks.
// Needs 16-byte aligned rows in every destination; the caller falls back to the dword form.
template <bool SWIGLU, bool ROPE>
__device__ __forceinline__ void gemm_pipe_epilogue_lds(const GemmF16Args& a, const f32x4 (&acc)[8][4],
                                                       char* smem, int trow0, int ncol0, int wave, int lane) {
  const int c16 = lane & 15, g = lane >> 4;
  if constexpr (SWIGLU) {
    // blocks 0,1 = gate, 2,3 = up of 32 hidden units; the workgroup's 256 rows x 128 units go to
    // LDS as fp16 (64 KiB; 16-byte chunk c of row r kept at chunk c ^ (r & 15)) and leave as
    // 256-byte row segments
    _Float16* wb = reinterpret_cast<_Float16*>(smem);
    const int half_k = a.K / 2;
    const int wr = wave >> 2, wc = wave & 3;
    const int tile_row0 = trow0 - wr * 128, tile_j0 = ((ncol0 - wc * 64) >> 6) * 32;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = ncol0 + ni * 16 + c16;
      const float fg = n < a.K ? *KX_CK(a, FB_COLF, a.colf + n) : 0.f, fu = n < a.K ? *KX_CK(a, FB_COLF, a.colf + (n + 32)) : 0.f;
      const int unit = wc * 32 + ni * 16 + c16;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wr * 128 + mi * 16 + 4 * g + r;
          const float gv = acc[mi][ni][r] * fg, uv = acc[mi][ni + 2][r] * fu;
          wb[row * 128 + ((((unit >> 3) ^ (row & 15)) << 3) | (unit & 7))] =
              sat_f16(silu_mul_fast(gv, uv));
        }
    }
    __syncthreads();
    // 16 lanes = one row of 128 units, 4 rows per instruction, 32 rows per wave
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = wave * 32 + it * 4 + g, t = tile_row0 + row, j = tile_j0 + c16 * 8;
      const f16x8 v = *reinterpret_cast<const f16x8*>(wb + row * 128 + ((c16 ^ (row & 15)) << 3));
      if (t < a.T && j < half_k) *KX_CK(a, FB_H16, reinterpret_cast<f16x8*>(a.h16 + (size_t)t * half_k + j)) = v;
    }
    return;
  } else {
    float* wb = reinterpret_cast<float*>(smem) + wave * 4096;
    float f[4], bias[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = ncol0 + ni * 16 + c16;
      f[ni] = n < a.K ? *KX_CK(a, FB_COLF, a.colf + n) : 0.f;
      bias[ni] = ROPE && a.bias && n < a.K ? *KX_CK(a, FB_BIAS, a.bias + n) : 0.f;
    }
    // this lane's four columns when reading back, and where they go
    const int n4 = ncol0 + 4 * c16;
    float* dst;
    int64_t ld;
    int col;
    [[maybe_unused]] int op;
    bool add_res = false;
    if (n4 < a.k0) {
      dst = a.y0; ld = a.ld0; col = n4; add_res = a.residual != nullptr; op = FB_Y0;
    } else if (n4 < a.k1) {
      dst = a.y1; ld = a.ld1; col = n4 - a.k0; op = FB_Y1;
    } else {
      dst = a.y2; ld = a.ld2; col = n4 - a.k1; op = FB_Y2;
    }
    // rotary embedding (ROPE form): the columns of q and k turn with their pair partner.  Pairs
    // are (2i, 2i+1) of a head (interleaved, rope_kernel.cu:98-128) or (i, i + hs/2) (half-split,
    // :5-36); the half-split partner sits 32 columns away inside this wave's block for 64-wide
    // heads and at the same place in the neighbouring wave's block for 128-wide heads.
    const bool turn = ROPE && a.rope_variant >= 0 && n4 < a.k1;
    const int d = col % a.head_size;                    // first of this lane's columns inside its head
    const bool split = a.rope_variant == KX_ROPE_HALF_SPLIT;
    const int hhalf = a.head_size >> 1;
    const bool low = d < hhalf;
    const int tab = split ? 2 * (low ? d : d - hhalf) : d;  // table index of the first pair
    const float* pb = a.head_size == 128 ? reinterpret_cast<float*>(smem) + (wave ^ 1) * 4096 : wb;
    const int pcol = a.head_size == 128 ? 4 * c16 : (4 * c16) ^ 32;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (ROPE && half) __syncthreads();  // the neighbour's reads of the first half are over
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            wb[(m4 * 16 + 4 * g + r) * 64 + ((ni * 16 + c16) ^ (16 * g))] =
                acc[half * 4 + m4][ni][r] * f[ni] + bias[ni];
      if (ROPE) __syncthreads();
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + g, t = trow0 + half * 64 + row;
        const int sw = 16 * ((row >> 2) & 3);
        f32x4 v = *reinterpret_cast<const f32x4*>(wb + row * 64 + ((4 * c16) ^ sw));
        if (turn) {
          const size_t trow = (size_t)(a.pos0 + min(t, a.T - 1)) * a.head_size + tab;
          if (split) {
            const f32x4 pv = *reinterpret_cast<const f32x4*>(pb + row * 64 + (pcol ^ sw));
            const f32x4 s0 = *KX_CK(a, FB_SIN, reinterpret_cast<const f32x4*>(a.sin_cache + trow));
            const f32x4 s1 = *KX_CK(a, FB_SIN, reinterpret_cast<const f32x4*>(a.sin_cache + trow + 4));
            const f32x4 c0 = *KX_CK(a, FB_COS, reinterpret_cast<const f32x4*>(a.cos_cache + trow));
            const f32x4 c1 = *KX_CK(a, FB_COS, reinterpret_cast<const f32x4*>(a.cos_cache + trow + 4));
            const f32x4 sn = {s0.x, s0.z, s1.x, s1.z}, cs = {c0.x, c0.z, c1.x, c1.z};
            // x_lo' = x_lo cos - x_hi sin ; x_hi' = x_lo sin + x_hi cos
            v = low ? v * cs - pv * sn : pv * sn + v * cs;
          } else {
            const f32x4 s0 = *KX_CK(a, FB_SIN, reinterpret_cast<const f32x4*>(a.sin_cache + trow));
            const f32x4 c0 = *KX_CK(a, FB_COS, reinterpret_cast<const f32x4*>(a.cos_cache + trow));
            v = f32x4{v.x * c0.x - v.y * s0.x, v.x * s0.x + v.y * c0.x,
                      v.z * c0.z - v.w * s0.z, v.z * s0.z + v.w * c0.z};
          }
        }
        if (t < a.T && n4 < a.K) {
          if (add_res) {
            const f32x4 rv = *KX_CK(a, FB_RES, reinterpret_cast<const f32x4*>(a.residual + (size_t)t * a.ldr + col));
            v += rv;
          }
          *KX_CK(a, op, reinterpret_cast<f32x4*>(dst + (size_t)t * ld + col)) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next half overwrites
    }
  }
}

// acc = part 0 + part 1 + ... in index order (three parts or more: the sum must not depend on which
// part arrives last), every part (this workgroup's own too) read back from its slab with
// device-scope loads; the S loads of one 16 x 16 block are in flight together and
// the unrolled loop lets the next blocks' loads go out under the sums.  Row blocks past the last
// row (`live` blocks of 16 rows in this wave's tile) were never stored and are skipped.
template <int S>
__device__ __forceinline__ void gp_combine(const GemmF16Args& a, const char* slab_base, f32x4 (&acc)[8][4],
                                           __amdgpu_buffer_rsrc_t rs, unsigned lane_off, int live) {
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    if (mi >= live) break;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      u32x4 v[S];
#pragma unroll
      for (int pp = 0; pp < S; ++pp) {
        const unsigned off = (unsigned)pp * 262144u + lane_off + (mi * 4 + ni) * 1024;
        KX_TOUCH(a, FB_SLABS, slab_base + off, 16);
        v[pp] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
      }
      f32x4 t = {__uint_as_float(v[0].x), __uint_as_float(v[0].y), __uint_as_float(v[0].z),
                 __uint_as_float(v[0].w)};
#pragma unroll
      for (int pp = 1; pp < S; ++pp) {
        t[0] += __uint_as_float(v[pp].x); t[1] += __uint_as_float(v[pp].y);
        t[2] += __uint_as_float(v[pp].z); t[3] += __uint_as_float(v[pp].w);
      }
      acc[mi][ni] = t;
    }
  }
}

// first row and column of tile `wgid` in the order the workgroups walk the tiles
__device__ __forceinline__ void gp_tile_origin(const GemmF16Args& a, int wgid, int& t0, int& n0) {
  const int tiles_m = (a.T + 255) / 256, tiles_n = (a.K + 255) / 256;
  const int panel = wgid / (tiles_m * GP_GROUP_N), local = wgid - panel * (tiles_m * GP_GROUP_N);
  const int gn = min(GP_GROUP_N, tiles_n - panel * GP_GROUP_N);
  t0 = (local / gn) * 256;
  n0 = (panel * GP_GROUP_N + local % gn) * 256;
}

template <bool SWIGLU, bool ROPE = false, bool SHORT = false>
__device__ __forceinline__ void gemm_f16_pipe_body(const GemmF16Args& a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // workgroups that share an XCD (ids equal mod 8) take consecutive tiles, walking down the token
  // dimension first: they share W column panels through that XCD's L2
  // two classes of workgroups, whole tiles first: each class is dealt over the XCDs on its own,
  // so every XCD gets the same share of both
  const bool whole = (int)blockIdx.x < a.full_tiles;
  const int bid = whole ? blockIdx.x : blockIdx.x - a.full_tiles;
  const int nwg = whole ? a.full_tiles : (int)gridDim.x - a.full_tiles;
  const int q8 = nwg >> 3, r8n = nwg & 7, xcd = bid & 7;
  const int unit = (xcd < r8n ? xcd * (q8 + 1) : r8n * (q8 + 1) + (xcd - r8n) * q8) + (bid >> 3);
  const int nsplit = whole ? 1 : a.split;
  const int wgid = whole ? unit : a.full_tiles + unit / nsplit;
  const int part = whole ? 0 : unit % nsplit;
  // inside that sequence, panels of GP_GROUP_N column tiles are walked column-first: the ~32
  // workgroups an XCD runs at a time then cover about 8 x 4 tiles and share 12 operand panels
  // instead of 18
  int t0, n0;
  gp_tile_origin(a, wgid, t0, n0);

  GemmPipeCtx c;
  // this workgroup's share of the reduction: K-tiles [kt0, kt0 + c.nt)
  int kt0 = 0;
  {
    const int nt_all = a.M / GF_BK, base = nt_all / nsplit, rem = nt_all % nsplit;
    kt0 = part * base + min(part, rem);
    c.nt = base + (part < rem ? 1 : 0);
  }
  {
    const int r8 = lane >> 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gi = 2 * wave + j;
#pragma unroll
      for (int piece = 0; piece < 4; ++piece) {
        const bool is_a = piece & 1;
        const int row = is_a ? (gi >> 3) * 128 + (gi & 7) * 8 + (piece == 3 ? 64 : 0)
                             : (gi >> 2) * 64 + (gi & 3) * 8 + (piece == 2 ? 32 : 0);
        const int rl = row + r8;
        const int chunk = (lane & 7) ^ ((rl >> 1) & 7);
        // rows past the end re-read the last valid row; their outputs are never stored
        c.src[piece][j] = (is_a ? a.X + (size_t)min(t0 + rl, a.T - 1) * a.M + chunk * 8
                                : a.W + (size_t)min(n0 + rl, a.K - 1) * a.M + chunk * 8) + kt0 * GF_BK;
        c.dst[piece][j] = (is_a ? 0 : GP_BOFF) + row * 128;
      }
    }
    c.live_lo = t0 + wr * 128 < a.T;
    c.live_hi = t0 + wr * 128 + 64 < a.T;
    const int frow = lane & 15, kq = lane >> 4, key = (frow >> 1) & 7;
    c.abase = (wr * 128 + frow) * 128;
    c.bbase = GP_BOFF + (wc * 64 + frow) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) c.coff[kk] = ((kk * 4 + kq) ^ key) << 4;
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;
  f16x8 af[4][2], bx[2][2], by[2][2];

  if (wr == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one barrier behind
  // prologue: phases -6 .. -2 issue five pieces, phase -1 reads column block 0 of tile 0
  gp_issue<0, 0>(a, smem, c, 0);
  gp_issue<1, 0>(a, smem, c, 0);
  gp_issue<2, 0>(a, smem, c, 0);
  gp_issue<3, 0>(a, smem, c, 0);
  gp_issue<0, 1>(a, smem, c, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_s_barrier();
  gp_read16<0, 0, 2>(smem, c.bbase, c.coff, bx);
  gp_issue<1, 1>(a, smem, c, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_s_barrier();

  for (int t = 0; t < c.nt; t += 2) {
    gp_tile<0, SHORT>(a, smem, c, t, acc, af, bx, by);
    if (t + 1 < c.nt) gp_tile<1, SHORT>(a, smem, c, t + 1, acc, af, by, bx);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail loads target this LDS
  if (nsplit > 1) {
    // Partial sums of one tile meet in fp32 slabs ([part][wave][block][lane] x 16 bytes: whole
    // 1 KiB lines per wave instruction).  Write-through stores, drained by every wave, then one
    // arrival ticket per tile (it wraps back to zero); the workgroup that arrives last adds the
    // parts in index order -- the same sum whichever arrives last -- with device-scope loads
    // and writes the tile.  No workgroup ever waits for another.
    char* const slab_base = reinterpret_cast<char*>(a.slabs + (size_t)(wgid - a.full_tiles) * nsplit * 65536);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab_base, 0, (unsigned)nsplit * 262144u, 0x00020000);
    const unsigned mine = (unsigned)part * 262144u + (unsigned)(wave * 32 * 1024 + lane * 16);
    // 16-row blocks of this wave's tile that hold real rows (a short prompt fills few of them)
    const int live = min(8, max(0, (a.T - (t0 + wr * 128) + 15) / 16));
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      if (mi >= live) break;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        u32x4 v;
        v.x = __float_as_uint(acc[mi][ni][0]); v.y = __float_as_uint(acc[mi][ni][1]);
        v.z = __float_as_uint(acc[mi][ni][2]); v.w = __float_as_uint(acc[mi][ni][3]);
        KX_TOUCH(a, FB_SLABS, slab_base + (mine + (mi * 4 + ni) * 1024), 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, mine + (mi * 4 + ni) * 1024, 0, 16);
      }
    }
    if (a.combine_later) return;  // the kernel boundary publishes the slabs; the combine launch does the rest
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // also: every wave is past its last read of the K-tile buffers
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) *flag = atomicInc(KX_CK(a, FB_TICKETS, a.tickets + (wgid - a.full_tiles)), (unsigned)nsplit - 1u) == (unsigned)nsplit - 1u;
    __syncthreads();
    const int last = *flag;
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler order only: sc1 loads below
    const unsigned lane_off = (unsigned)(wave * 32 * 1024 + lane * 16);
    switch (nsplit) {
      case 2:  // a + b = b + a: the other part's slab onto this part's registers
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          if (mi >= live) break;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const unsigned off = (unsigned)(part ^ 1) * 262144u + lane_off + (mi * 4 + ni) * 1024;
            KX_TOUCH(a, FB_SLABS, slab_base + off, 16);
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
            acc[mi][ni][0] += __uint_as_float(v.x); acc[mi][ni][1] += __uint_as_float(v.y);
            acc[mi][ni][2] += __uint_as_float(v.z); acc[mi][ni][3] += __uint_as_float(v.w);
          }
        }
        break;
      case 4: gp_combine<4>(a, slab_base, acc, rs, lane_off, live); break;
      case 8: gp_combine<8>(a, slab_base, acc, rs, lane_off, live); break;
      default: gp_combine<16>(a, slab_base, acc, rs, lane_off, live); break;
    }
  }
  if (a.wide_rows) {
    __syncthreads();  // every wave's last tile reads and LDS-DMA landings are over: LDS is free
    gemm_pipe_epilogue_lds<SWIGLU, ROPE>(a, acc, smem, t0 + wr * 128, n0 + wc * 64, wave, lane);
  } else {
    gemm_f16_epilogue16<8, 4, SWIGLU>(a, acc, t0 + wr * 128, n0 + wc * 64, lane);
  }
}

// The sum of a cut reduction spread over the chip.  With four parts or more the last arriver of a
// tile read S x 256 KiB of slabs alone -- one workgroup fetching 1-4 MiB at the 60-70 GB/s a
// workgroup gets from other CUs' fresh stores: 20-40 us behind a main loop of 10-15 (Wo of
// Llama-3-8B at 256 rows: 59 us, of which the matrix work is a fifth).  Here one WAVE takes one
// 16-row block of one wave tile (16 x 64 outputs): 1024 waves for 16 tiles, each adding S x 4 KiB
// in part order (the same sum whatever finished first), turning its block round in 4 KiB of LDS
// into whole 256-byte row segments, and writing it with the plain epilogue's row factor, routing
// and residual -- or (ROPE) the Wq|Wk|Wv form's bias, rotary embedding and K/V row placement: the
// four waves of a workgroup hold four neighbouring 64-column blocks of one row block, so the
// half-split partner of a 128-wide head is the neighbouring wave's image, as in the GEMM's own
// epilogue.  Launched behind the GEMM whose parts only wrote their slabs.
template <int S, bool ROPE>
__global__ __launch_bounds__(256) void gemm_split_combine_kernel(const GemmF16Args a) {
  __shared__ __attribute__((aligned(16))) float lds[4][1024];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform, and hipcc knows
  const int unit = blockIdx.x * 4 + wv;              // (tile of the partial round, row block, wave position)
  const int r = unit >> 6, mi = (unit >> 3) & 7, wpos = unit & 7;
  const int wgid = a.full_tiles + r;
  int t0, n0;
  gp_tile_origin(a, wgid, t0, n0);
  const int wr = wpos >> 2, wc = wpos & 3;
  const int trow0 = t0 + wr * 128 + mi * 16, ncol0 = n0 + wc * 64;
  if (trow0 >= a.T) return;                          // a block of padding rows: never stored
  const int c16 = lane & 15, g = lane >> 4;
  const char* slab_base = reinterpret_cast<const char*>(a.slabs + (size_t)r * S * 65536);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(slab_base), 0, (unsigned)S * 262144u, 0x00020000);
  const unsigned lane_off = (unsigned)(wpos * 32 * 1024 + mi * 4 * 1024 + lane * 16);
  float* wb = lds[wv];
  constexpr int NIB = S <= 8 ? 4 : 2;  // column blocks whose loads are in flight together (32 loads per lane)
#pragma unroll
  for (int nb = 0; nb < 4; nb += NIB) {
    u32x4 v[NIB][S];
#pragma unroll
    for (int j = 0; j < NIB; ++j)
#pragma unroll
      for (int pp = 0; pp < S; ++pp) {
        const unsigned off = (unsigned)pp * 262144u + lane_off + (nb + j) * 1024;
        KX_TOUCH(a, FB_SLABS, slab_base + off, 16);
        v[j][pp] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      }
#pragma unroll
    for (int j = 0; j < NIB; ++j) {
      const int ni = nb + j;
      f32x4 t = {__uint_as_float(v[j][0].x), __uint_as_float(v[j][0].y), __uint_as_float(v[j][0].z),
                 __uint_as_float(v[j][0].w)};
#pragma unroll
      for (int pp = 1; pp < S; ++pp) {
        t[0] += __uint_as_float(v[j][pp].x); t[1] += __uint_as_float(v[j][pp].y);
        t[2] += __uint_as_float(v[j][pp].z); t[3] += __uint_as_float(v[j][pp].w);
      }
      const int n = ncol0 + ni * 16 + c16;
      const float f = n < a.K ? *KX_CK(a, FB_COLF, a.colf + n) : 0.f;
      const float bias = ROPE && a.bias && n < a.K ? *KX_CK(a, FB_BIAS, a.bias + n) : 0.f;
      // lane (column c16 of block ni, row group g) holds rows 4g .. 4g+3: the epilogue's LDS image
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) wb[(4 * g + rr) * 64 + ((ni * 16 + c16) ^ (16 * g))] = t[rr] * f + bias;
    }
  }
  if (ROPE) __syncthreads();  // the neighbouring wave's image is read below (uniform: one row block per workgroup)
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave reads back what its own lanes wrote
  const int n4 = ncol0 + 4 * c16;
  float* dst;
  int64_t ld;
  int col;
  [[maybe_unused]] int op;
  bool add_res = false;
  if (n4 < a.k0) {
    dst = a.y0; ld = a.ld0; col = n4; add_res = a.residual != nullptr; op = FB_Y0;
  } else if (n4 < a.k1) {
    dst = a.y1; ld = a.ld1; col = n4 - a.k0; op = FB_Y1;
  } else {
    dst = a.y2; ld = a.ld2; col = n4 - a.k1; op = FB_Y2;
  }
  // rotary embedding as in gemm_pipe_epilogue_lds (rope_kernel.cu:5-36 half split, :98-128 interleaved)
  const bool turn = ROPE && a.rope_variant >= 0 && n4 < a.k1;
  const int hsz = ROPE ? a.head_size : 1;
  const int d = col % hsz;
  const bool hsplit = a.rope_variant == KX_ROPE_HALF_SPLIT;
  const int hhalf = hsz >> 1;
  const bool low = d < hhalf;
  const int tab = hsplit ? 2 * (low ? d : d - hhalf) : d;
  const float* pb = hsz == 128 ? lds[wv ^ 1] : wb;
  const int pcol = hsz == 128 ? 4 * c16 : (4 * c16) ^ 32;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 4 + g, t = trow0 + row;
    const int sw = 16 * ((row >> 2) & 3);
    f32x4 o = *reinterpret_cast<const f32x4*>(wb + row * 64 + ((4 * c16) ^ sw));
    if (turn) {
      const size_t trow = (size_t)(a.pos0 + min(t, a.T - 1)) * a.head_size + tab;
      if (hsplit) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(pb + row * 64 + (pcol ^ sw));
        const f32x4 s0 = *KX_CK(a, FB_SIN, reinterpret_cast<const f32x4*>(a.sin_cache + trow));
        const f32x4 s1 = *KX_CK(a, FB_SIN, reinterpret_cast<const f32x4*>(a.sin_cache + trow + 4));
        const f32x4 c0 = *KX_CK(a, FB_COS, reinterpret_cast<const f32x4*>(a.cos_cache + trow));
        const f32x4 c1 = *KX_CK(a, FB_COS, reinterpret_cast<const f32x4*>(a.cos_cache + trow + 4));
        const f32x4 sn = {s0.x, s0.z, s1.x, s1.z}, cs = {c0.x, c0.z, c1.x, c1.z};
        o = low ? o * cs - pv * sn : pv * sn + o * cs;
      } else {
        const f32x4 s0 = *KX_CK(a, FB_SIN, reinterpret_cast<const f32x4*>(a.sin_cache + trow));
        const f32x4 c0 = *KX_CK(a, FB_COS, reinterpret_cast<const f32x4*>(a.cos_cache + trow));
        o = f32x4{o.x * c0.x - o.y * s0.x, o.x * s0.x + o.y * c0.x, o.z * c0.z - o.w * s0.z, o.z * s0.z + o.w * c0.z};
      }
    }
    if (t < a.T && n4 < a.K) {
      if (add_res) o += *KX_CK(a, FB_RES, reinterpret_cast<const f32x4*>(a.residual + (size_t)t * a.ldr + col));
      *KX_CK(a, op, reinterpret_cast<f32x4*>(dst + (size_t)t * ld + col)) = o;
    }
  }
}

__global__ __launch_bounds__(512, 1) void gemm_f16_pipe_kernel(const GemmF16Args a) { gemm_f16_pipe_body<false>(a); }
__global__ __launch_bounds__(512, 1) void gemm_f16_pipe_swiglu_kernel(const GemmF16Args a) { gemm_f16_pipe_body<true>(a); }
__global__ __launch_bounds__(512, 1) void gemm_f16_pipe_qkv_rope_kernel(const GemmF16Args a) { gemm_f16_pipe_body<false, true>(a); }
__global__ __launch_bounds__(512, 1) void gemm_f16_pipe_short_kernel(const GemmF16Args a) { gemm_f16_pipe_body<false, false, true>(a); }
__global__ __launch_bounds__(512, 1) void gemm_f16_pipe_swiglu_short_kernel(const GemmF16Args a) { gemm_f16_pipe_body<true, false, true>(a); }
__global__ __launch_bounds__(512, 1) void gemm_f16_pipe_qkv_rope_short_kernel(const GemmF16Args a) { gemm_f16_pipe_body<false, true, true>(a); }

// Split workspace: kSplitTicketBytes of arrival tickets (one per tile of the last, partial round),
// then 256 KiB slabs.
constexpr size_t kSplitTicketBytes = 4096;
constexpr size_t kSplitSlabBytes = 256 * 256 * 4;
constexpr int kPipeCUs = 256;

// How many parts (1, 2, 4, 8 or 16) the reduction of the tiles of the last, partial round is cut
// into.  Model, in microseconds (fitted on 8B-sized matrices at 64..4096 rows): a K-tile step takes
// ~1.45, a part costs ~4 on top of its steps (pipeline fill, epilogue); a round of R tiles cut S
// ways takes ceil(R S / 256) rounds of nt / S steps, plus the slabs' way through memory -- every
// part writes its live rows (256 KiB for a full tile) and the last arriver reads them all back,
// ~0.16 per full slab with the whole chip doing it at once.
// later: the sum of four parts or more runs as a launch of its own over the whole chip
// (gemm_split_combine_kernel): about 0.125 per slab plus a launch, fitted on Wo / Wq|Wk|Wv / W2 of
// Llama-3-8B at 256-512 rows.
static int pipe_split_choice(int tiles, int nt, int T, size_t split_ws_bytes, bool later = false) {
  const int R = tiles % kPipeCUs;
  if (R == 0 || split_ws_bytes <= kSplitTicketBytes) return 1;
  const size_t slabs_fit = (split_ws_bytes - kSplitTicketBytes) / kSplitSlabBytes;
  const double live = T >= 256 ? 1.0 : (double)((T + 15) / 16 * 16) / 256.0;  // share of a slab in use
  int best = 1;
  double best_cost = 1.45 * nt + 4.0;
  for (int S = 2; S <= 16; S *= 2) {
    if (nt < 8 * S || (size_t)R * S > slabs_fit || R > (int)(kSplitTicketBytes / 4)) continue;
    const double sum_cost = later && S >= 4 ? 0.125 * live * R * S + 1.0 : 0.16 * live * R * S;
    const double cost = (double)((R * S + kPipeCUs - 1) / kPipeCUs) * (1.45 * nt / S + 4.0) + sum_cost;
    if (cost < 0.95 * best_cost) {
      best_cost = cost;
      best = S;
    }
  }
  return best;
}

// plain: the kernel has the plain / residual epilogue; rope: the Wq|Wk|Wv epilogue -- the two forms
// the separate combine launch knows (SwiGLU keeps its sum with the last arriver)
template <void (*KERN)(GemmF16Args)>
static int launch_gemm_f16_pipe(GemmF16Args a, void* split_ws, size_t split_ws_bytes, hipStream_t s,
                                bool need_wide = false, bool plain = false, bool rope = false) {
  const int tiles_m = (a.T + 255) / 256, tiles_n = (a.K + 255) / 256;
  const int tiles = tiles_m * tiles_n;
  // 128 KiB of dynamic LDS needs the opt-in, once per kernel and device
  static OncePerDevice opted_in;  // one instance per kernel
  if (int rc = opted_in.raise(reinterpret_cast<const void*>(KERN), 2 * GP_BUF)) return rc;
  {
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool swiglu = a.h16 != nullptr;
    a.wide_rows = swiglu ? (al16(a.h16) && (a.K / 2) % 8 == 0)
                         : (al16(a.y0) && al16(a.y1) && al16(a.y2) && a.ld0 % 4 == 0 && a.ld1 % 4 == 0 &&
                            a.ld2 % 4 == 0 && a.k0 % 4 == 0 && a.k1 % 4 == 0 && a.K % 4 == 0 &&
                            (!a.residual || (al16(a.residual) && a.ldr % 4 == 0)));
    if (getenv("KX_GEMM_NARROW_EPILOGUE") && !need_wide) a.wide_rows = 0;  // A/B runs and tests
    if (need_wide && !a.wide_rows) return KX_EUNSUPPORTED;
  }
  // four parts or more of a plain or Wq|Wk|Wv product: the sum runs as a launch of its own, spread
  // over the chip (KX_GEMM_COMBINE=0 keeps it with the last arriver, for A/B runs and tests)
  const char* combine_env = getenv("KX_GEMM_COMBINE");
  const bool later = (plain || rope) && a.wide_rows && !(combine_env && combine_env[0] == '0');
  int split = split_ws ? pipe_split_choice(tiles, a.M / GF_BK, a.T, split_ws_bytes, later) : 1;
  if (const char* e = getenv("KX_GEMM_SPLIT")) {  // A/B runs and tests: 1, 2, 4, 8 or 16
    const int want = atoi(e);
    const size_t need = kSplitTicketBytes + (size_t)(tiles % kPipeCUs) * want * kSplitSlabBytes;
    if (want == 1 || ((want == 2 || want == 4 || want == 8 || want == 16) && split_ws && split_ws_bytes >= need &&
                      a.M / GF_BK >= 2 * want && tiles % kPipeCUs && tiles % kPipeCUs <= 1024))
      split = want;
  }
  a.split = split;
  a.full_tiles = split > 1 ? tiles - tiles % kPipeCUs : tiles;
  a.tickets = static_cast<unsigned*>(split_ws);
  a.slabs = reinterpret_cast<float*>(static_cast<char*>(split_ws) + kSplitTicketBytes);
  const int grid = a.full_tiles + (tiles - a.full_tiles) * split;
  a.combine_later = later && split >= 4;
  KX_BSET(a, FB_SLABS, split_ws ? a.slabs : nullptr); KX_BSET(a, FB_TICKETS, a.tickets);
  hipLaunchKernelGGL(KERN, dim3(grid), dim3(512), 2 * GP_BUF, s, a);
  KX_LAUNCH_CHECK();
  if (a.combine_later) {
    const int units = (tiles - a.full_tiles) * 64;  // (tile, wave position, 16-row block), one wave each
    const dim3 cgrid((units + 3) / 4), cblock(256);
#define KX_COMBINE(S_)                                                                             \
  do {                                                                                              \
    if (rope) hipLaunchKernelGGL((gemm_split_combine_kernel<S_, true>), cgrid, cblock, 0, s, a);     \
    else hipLaunchKernelGGL((gemm_split_combine_kernel<S_, false>), cgrid, cblock, 0, s, a);         \
  } while (0)
    switch (split) {
      case 4: KX_COMBINE(4); break;
      case 8: KX_COMBINE(8); break;
      default: KX_COMBINE(16); break;
    }
#undef KX_COMBINE
    KX_LAUNCH_CHECK();
  }
  return KX_OK;
}

template <int MI, int NI, typename Kern>
static int launch_gemm_f16(Kern kern, const GemmF16Args& a, hipStream_t s) {
  constexpr int BM = 64 * MI, BN = 64 * NI;
  constexpr size_t lds = (size_t)(BM + BN) * GF_BK * 2;
  const int tiles_m = (a.T + BM - 1) / BM, tiles_n = (a.K + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, s, a);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace kx

extern "C" size_t kx_gemm_w8_workspace_bytes(int32_t T, int32_t M, int32_t K) {
  if (T <= 0 || M <= 0 || K <= 0) return 0;
  return kx::align256((size_t)T * M * 2) + kx::align256((size_t)K * M * 2) + kx::align256((size_t)K * 4);
}

extern "C" int kx_f32_to_f16_rows(const float* src, int64_t ld_src, void* dst16, int64_t ld_dst,
                                  int32_t cols, int32_t rows, void* stream) {
  if (!src || !dst16 || cols <= 0 || rows <= 0 || ld_src < cols || ld_dst < cols) return KX_EINVAL;
  if (cols % 4 || ld_src % 4 || ld_dst % 4 || (uintptr_t)src % 16 || (uintptr_t)dst16 % 8) return KX_EINVAL;
  const int64_t n4 = (int64_t)rows * (cols / 4);
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(kx::f32_to_f16_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                     ld_src, static_cast<_Float16*>(dst16), ld_dst, cols, n4 KX_BPASS(2, src, dst16));
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" size_t kx_gemm_w8_f16act_workspace_bytes(int32_t M, int32_t K) {
  if (M <= 0 || K <= 0) return 0;
  return kx::align256((size_t)K * M * 2) + kx::align256((size_t)K * 4);
}

static bool gemm_f16_shape_ok(int32_t group_size, int32_t M, const void* w, const void* workspace) {
  const bool pow2 = (group_size & (group_size - 1)) == 0;
  return pow2 && group_size >= 16 && M % group_size == 0 && M % kx::GF_BK == 0 &&
         ((uintptr_t)w % 16 == 0) && ((uintptr_t)workspace % 256 == 0);
}

// Dequantises W into the workspace and runs the fp16 GEMM on fp16 activations.
static int gemm_f16act_run(const _Float16* x16, const int8_t* w, const float* scales,
                           int32_t group_size, int32_t mode, _Float16* h16, float* y0, int64_t ld0,
                           float* y1, int64_t ld1, float* y2, int64_t ld2, int32_t k0, int32_t k1,
                           const float* residual, int64_t ldr, int32_t T, int32_t M, int32_t K,
                           char* ws, hipStream_t s, bool weights_ready = false,
                           void* split_ws = nullptr, size_t split_ws_bytes = 0,
                           const kx::GemmF16Args* qkv_form = nullptr) {
  using namespace kx;
  _Float16* w16 = reinterpret_cast<_Float16*>(ws);
  float* colf = reinterpret_cast<float*>(ws + align256((size_t)K * M * 2));
  int gs_shift = 0;
  while ((1 << gs_shift) < group_size) ++gs_shift;
  const bool swiglu = mode == KX_GEMM_SWIGLU_F16;
  if (!weights_ready) {
    hipLaunchKernelGGL(w8_dequant_rows_kernel, dim3((K + 3) / 4), dim3(256), 0, s, w, scales, w16, colf,
                       M, K, gs_shift, swiglu ? K / 2 : 0 KX_BPASS(4, w, scales, w16, colf));
    KX_LAUNCH_CHECK();
  }
  GemmF16Args a{};
  a.X = x16; a.W = w16; a.colf = colf; a.residual = residual; a.h16 = h16;
  a.y0 = y0; a.y1 = y1 ? y1 : y0; a.y2 = y2 ? y2 : y0;
  a.ld0 = ld0; a.ld1 = ld1; a.ld2 = ld2; a.ldr = ldr;
  a.k0 = k0; a.k1 = k1; a.T = T; a.M = M; a.K = K;
  a.rope_variant = -1;
  if (qkv_form) {
    a.bias = qkv_form->bias; a.sin_cache = qkv_form->sin_cache; a.cos_cache = qkv_form->cos_cache;
    a.rope_variant = qkv_form->rope_variant; a.head_size = qkv_form->head_size; a.pos0 = qkv_form->pos0;
  }
  KX_BSET(a, FB_X, a.X); KX_BSET(a, FB_W, a.W); KX_BSET(a, FB_COLF, a.colf); KX_BSET(a, FB_RES, a.residual);
  KX_BSET(a, FB_H16, a.h16); KX_BSET(a, FB_Y0, a.y0); KX_BSET(a, FB_Y1, a.y1); KX_BSET(a, FB_Y2, a.y2);
  KX_BSET(a, FB_SLABS, nullptr); KX_BSET(a, FB_TICKETS, nullptr);
  KX_BSET(a, FB_BIAS, a.bias); KX_BSET(a, FB_SIN, a.sin_cache); KX_BSET(a, FB_COS, a.cos_cache);
  // Kernel choice (KX_GEMM_F16_KERNEL = 1 / 3 forces one, for A/B runs and tests):
  //   1  128 x 128 tiles, four workgroups per CU      3  pipelined 256 x 256, one per CU
  // The pipelined kernel wins from 512 rows up on every layer matrix of Llama-3-8B, with or
  // without cut reductions (per-layer GEMM time at 512 / 1024 / 2048 / 4096 rows: 645 / 815 / 1267 /
  // 2454 us with the small tiles, 331 / 564 / 936 / 1448 with the large ones); the small tiles
  // keep the reductions too short to fill its pipeline.  A 256 x 128 variant (two per CU) was never
  // the fastest of the three and is gone.
  int choice = 0;
  if (const char* e = getenv("KX_GEMM_F16_KERNEL")) choice = atoi(e);  // a few ns; read per call so tests can switch
  if (choice != 1 && choice != 3) choice = M >= 512 ? 3 : 1;
  // at most 128 rows: the variant that skips the matrix work of all-padding wave halves
  const bool short_rows = T <= 128 && !getenv("KX_GEMM_NO_SHORT");
  if (qkv_form)  // only the pipelined kernel has this epilogue; the caller checked the shape
    return short_rows
               ? launch_gemm_f16_pipe<gemm_f16_pipe_qkv_rope_short_kernel>(a, split_ws, split_ws_bytes, s, true, false, true)
               : launch_gemm_f16_pipe<gemm_f16_pipe_qkv_rope_kernel>(a, split_ws, split_ws_bytes, s, true, false, true);
  if (choice == 3 && short_rows)
    return swiglu ? launch_gemm_f16_pipe<gemm_f16_pipe_swiglu_short_kernel>(a, split_ws, split_ws_bytes, s)
                  : launch_gemm_f16_pipe<gemm_f16_pipe_short_kernel>(a, split_ws, split_ws_bytes, s, false, true);
  if (choice == 3)
    return swiglu ? launch_gemm_f16_pipe<gemm_f16_pipe_swiglu_kernel>(a, split_ws, split_ws_bytes, s)
                  : launch_gemm_f16_pipe<gemm_f16_pipe_kernel>(a, split_ws, split_ws_bytes, s, false, true);
  return swiglu ? launch_gemm_f16<2, 2>(gemm_f16_swiglu_kernel_22, a, s)
                : launch_gemm_f16<2, 2>(gemm_f16_kernel_22, a, s);
}

extern "C" size_t kx_gemm_split_workspace_bytes(void) {
  // tickets + 768 slabs: four parts for up to 192 tiles, two parts for any partial round
  return kx::kSplitTicketBytes + 768 * kx::kSplitSlabBytes;
}

extern "C" int kx_gemm_w8_f16act_split_ws(const void* X16, const int8_t* w, const float* scales,
                                          int32_t group_size, int32_t mode, void* h16, float* y0,
                                          int64_t ld0, float* y1, int64_t ld1, float* y2, int64_t ld2,
                                          int32_t k0, int32_t k1, const float* residual, int64_t ldr,
                                          int32_t T, int32_t M, int32_t K, void* workspace,
                                          size_t workspace_bytes, void* split_workspace,
                                          size_t split_workspace_bytes, void* stream) {
  if (!X16 || !w || !scales || T <= 0 || M <= 0 || K <= 0 || group_size <= 0) return KX_EINVAL;
  const bool weights_ready = (mode & KX_GEMM_WEIGHTS_READY) != 0;
  mode &= ~KX_GEMM_WEIGHTS_READY;
  if (!workspace || workspace_bytes < kx_gemm_w8_f16act_workspace_bytes(M, K)) return KX_EWORKSPACE;
  if (!gemm_f16_shape_ok(group_size, M, w, workspace) || (uintptr_t)X16 % 16) return KX_EUNSUPPORTED;
  if (split_workspace && (uintptr_t)split_workspace % 256) return KX_EINVAL;
  if (mode == KX_GEMM_SWIGLU_F16) {
    // gate rows [0, K/2), up rows [K/2, K); whole blocks of 32 hidden units
    if (!h16 || K % 64 || residual) return KX_EINVAL;
    k0 = k1 = K;
  } else if (mode == KX_GEMM_ROUTE_F32) {
    if (!y0) return KX_EINVAL;
    if (!y1 && !y2) { k0 = K; k1 = K; }
    if (k0 < 0 || k1 < k0 || k1 > K) return KX_EINVAL;
    if ((k0 < K && !y1 && k1 > k0) || (k1 < K && !y2)) return KX_EINVAL;
  } else {
    return KX_EINVAL;
  }
  return gemm_f16act_run(static_cast<const _Float16*>(X16), w, scales, group_size, mode,
                         static_cast<_Float16*>(h16), y0, ld0, y1, ld1, y2, ld2, k0, k1, residual, ldr,
                         T, M, K, static_cast<char*>(workspace), (hipStream_t)stream, weights_ready,
                         split_workspace, split_workspace_bytes);
}

extern "C" int kx_gemm_w8_f16act_qkv_rope(const void* X16, const int8_t* w, const float* scales,
                                          int32_t group_size, int32_t weights_ready, float* q,
                                          int64_t ld_q, float* k, int64_t ld_k, float* v, int64_t ld_v,
                                          int32_t q_dim, int32_t kv_dim, int32_t head_size,
                                          int32_t rope_variant, const float* sin_cache,
                                          const float* cos_cache, int32_t pos0, const float* bias,
                                          int32_t T, int32_t M, void* workspace, size_t workspace_bytes,
                                          void* split_workspace, size_t split_workspace_bytes,
                                          void* stream) {
  if (!X16 || !w || !scales || !q || !k || !v || !sin_cache || !cos_cache) return KX_EINVAL;
  if (T <= 0 || M <= 0 || q_dim <= 0 || kv_dim <= 0 || group_size <= 0 || pos0 < 0) return KX_EINVAL;
  if (rope_variant != KX_ROPE_INTERLEAVED && rope_variant != KX_ROPE_HALF_SPLIT) return KX_EINVAL;
  const int32_t K = q_dim + 2 * kv_dim;
  if (!workspace || workspace_bytes < kx_gemm_w8_f16act_workspace_bytes(M, K)) return KX_EWORKSPACE;
  if (!gemm_f16_shape_ok(group_size, M, w, workspace) || (uintptr_t)X16 % 16) return KX_EUNSUPPORTED;
  // whole heads per 64-column wave block (64) or per pair of them (128); tables read 16 bytes at a time
  if ((head_size != 64 && head_size != 128) || q_dim % head_size || kv_dim % head_size ||
      ((uintptr_t)sin_cache | (uintptr_t)cos_cache) % 16 || (bias && (uintptr_t)bias % 4))
    return KX_EUNSUPPORTED;
  if (split_workspace && (uintptr_t)split_workspace % 256) return KX_EINVAL;
  kx::GemmF16Args form{};
  form.bias = bias; form.sin_cache = sin_cache; form.cos_cache = cos_cache;
  form.rope_variant = rope_variant; form.head_size = head_size; form.pos0 = pos0;
  return gemm_f16act_run(static_cast<const _Float16*>(X16), w, scales, group_size, KX_GEMM_ROUTE_F32,
                         nullptr, q, ld_q, k, ld_k, v, ld_v, q_dim, q_dim + kv_dim, nullptr, 0, T, M, K,
                         static_cast<char*>(workspace), (hipStream_t)stream, weights_ready != 0,
                         split_workspace, split_workspace_bytes, &form);
}

extern "C" int kx_gemm_w8_f16act_ws(const void* X16, const int8_t* w, const float* scales,
                                    int32_t group_size, int32_t mode, void* h16, float* y0,
                                    int64_t ld0, float* y1, int64_t ld1, float* y2, int64_t ld2,
                                    int32_t k0, int32_t k1, const float* residual, int64_t ldr,
                                    int32_t T, int32_t M, int32_t K, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return kx_gemm_w8_f16act_split_ws(X16, w, scales, group_size, mode, h16, y0, ld0, y1, ld1, y2, ld2, k0,
                                    k1, residual, ldr, T, M, K, workspace, workspace_bytes, nullptr, 0,
                                    stream);
}

extern "C" int kx_gemm_w8_f32_ws(const float* X, int64_t ldx, const int8_t* w, const float* scales,
                                 int32_t group_size, float* y0, int64_t ld0, float* y1,
                                 int64_t ld1, float* y2, int64_t ld2, int32_t k0, int32_t k1,
                                 const float* residual, int64_t ldr, int32_t T, int32_t M,
                                 int32_t K, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace kx;
  if (!X || !w || !scales || !y0 || T <= 0 || M <= 0 || K <= 0 || group_size <= 0) return KX_EINVAL;
  const bool two_pass = workspace && workspace_bytes >= kx_gemm_w8_workspace_bytes(T, M, K) &&
                        T >= 512 && gemm_f16_shape_ok(group_size, M, w, workspace) &&
                        ldx % 4 == 0 && ((uintptr_t)X % 16 == 0);
  if (!two_pass)  // short prompts (the two forms tie at 512..640 rows): converting the whole matrix costs more than it saves
    return kx_gemm_w8_f32_ex(X, ldx, w, scales, group_size, y0, ld0, y1, ld1, y2, ld2, k0, k1,
                             residual, ldr, T, M, K, stream);
  if (!y1 && !y2) { k0 = K; k1 = K; }
  if (k0 < 0 || k1 < k0 || k1 > K) return KX_EINVAL;
  if ((k0 < K && !y1 && k1 > k0) || (k1 < K && !y2)) return KX_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  char* ws = static_cast<char*>(workspace);
  _Float16* x16 = reinterpret_cast<_Float16*>(ws);
  const int64_t n4 = (int64_t)T * (M / 4);
  hipLaunchKernelGGL(f32_to_f16_rows_kernel, dim3(2048), dim3(256), 0, s, X, ldx, x16, (int64_t)M, M, n4 KX_BPASS(2, X, x16));
  KX_LAUNCH_CHECK();
  return gemm_f16act_run(x16, w, scales, group_size, KX_GEMM_ROUTE_F32, nullptr, y0, ld0, y1, ld1, y2,
                         ld2, k0, k1, residual, ldr, T, M, K, ws + align256((size_t)T * M * 2), s);
}
