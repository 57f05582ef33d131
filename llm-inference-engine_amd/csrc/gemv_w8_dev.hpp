// Device-side building blocks of the W8 GEMV (weight batches, x staging in LDS, row epilogues),
// shared by the GEMV kernels (gemv_w8.hip) and by kernels that run a GEMV as one of their phases
// (attn_decode.hip: the attention output projection fused behind the attention itself).
#pragma once
#include "gemv_w8.hpp"

namespace kx {

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

// Virtual row v -> matrix row.  With half-split RoPE the two rows of a rotation pair are hs/2
// apart; the virtual order puts them next to each other so one wave finishes both.
template <int MODE>
__device__ __forceinline__ int map_row(const GemvW8Args& a, int v) {
  if constexpr (MODE == GEMV_QKV_ROPE) {
    if (a.rope_variant == KX_ROPE_HALF_SPLIT && v < a.k1) {
      const int hs = a.head_size;
      const int head = v / hs, i = v - head * hs;
      return head * hs + (i >> 1) + (i & 1) * (hs >> 1);
    }
  }
  return v;
}

__device__ __forceinline__ void route(const GemvW8Args& a, int row, int pos, float*& dst, int& r) {
  if (row < a.k0) {
    dst = a.y0;
    r = row;
  } else if (row < a.k1) {
    dst = a.y1 + (size_t)pos * a.kv_pos_stride;
    r = row - a.k0;
  } else {
    dst = a.y2 + (size_t)pos * a.kv_pos_stride;
    r = row - a.k1;
  }
}

__device__ __forceinline__ void store_out(const GemvW8Args& a, float* p, float v) {
  if (a.coherent_out) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// Row of the stacked q|k|v product -> its destination; k and v rows go to the position's cache row,
// rounded to fp16 when the cache holds fp16.
__device__ __forceinline__ void store_routed(const GemvW8Args& a, int row, int pos, float v) {
  if (a.kv_f16 && row >= a.k0) {
    const bool is_k = row < a.k1;
    _Float16* dst = reinterpret_cast<_Float16*>(is_k ? a.y1 : a.y2) + (size_t)pos * a.kv_pos_stride;
    dst[row - (is_k ? a.k0 : a.k1)] = (_Float16)v;
    return;
  }
  float* dst;
  int rr;
  route(a, row, pos, dst, rr);
  store_out(a, dst + rr, v);
}

// Issues the 128-bit weight loads (and the one scale per load) of U chunks of R rows.
template <int R, int U, int R2, int U2, int MODE>
__device__ __forceinline__ void load_batch(const GemvW8Args& a, int r0, int c0, int row_end,
                                           int col_lane, i32x4 (&wv)[R][U], float (&sc)[R][U],
                                           i32x4 (&wv2)[R2][U2], float (&sc2)[R2][U2]) {
  constexpr bool GU = MODE == GEMV_GATE_UP;
  int rows[R];
#pragma unroll
  for (int r = 0; r < R; ++r) rows[r] = map_row<MODE>(a, min(r0 + r, row_end - 1));
  // issue order = consumption order (chunk-major), so the first products can start while the
  // later loads are still in flight
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // Columns past M re-read the row's last 16 weights: x is staged as zero there, so the products
    // vanish.  Unconditional loads matter: under a per-lane column test the compiler branches
    // around every load and waits for the whole batch before the first product.
    const int col = min((c0 + u) * 1024 + col_lane, a.M - 16);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = rows[r];
      const size_t off = (size_t)row * (size_t)a.M + (size_t)col;
      wv[r][u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(a.w + off));
      sc[r][u] = __builtin_nontemporal_load(a.scales + (size_t)row * a.groups_per_row +
                                            (col >> a.gs_shift));
      if constexpr (GU) {
        const size_t off2 = off + (size_t)a.swiglu_half * (size_t)a.M;
        wv2[r][u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(a.w + off2));
        sc2[r][u] = __builtin_nontemporal_load(
            a.scales + (size_t)(row + a.swiglu_half) * a.groups_per_row + (col >> a.gs_shift));
      }
    }
  }
}

// Staging of x in LDS: slot (c*4+p)*64+l <- x[c*1024 + l*16 + p*4 .. +4), zero past M.  With a
// norm weight g the staged vector is g*x and commit_x returns rsqrt(mean(x^2)+eps), which the
// caller applies to the finished row sums: norm_w * (rs * x) of cpu/rmsnorm_kernel.cpp:30-32 with
// the scalar factored out of the dot product.
//
// Two steps, because loads return in order: request_x asks for a thread's first four vectors
// BEFORE the first weight batch is requested, commit_x (after it) waits for them only -- the
// weights stay in flight across the LDS writes, the barrier and the norm reduction.  Requested
// behind the weights, x could not be touched before every weight of the batch had arrived.
constexpr int kXPer = 4;  // loads per thread and pass, all requested before the first is used
template <int XP>
struct GemvXRegsT {
  f32x4 v[XP];
  f32x4 g[XP];
};
using GemvXRegs = GemvXRegsT<kXPer>;

// Unconditional loads with a clamped index: a load under a test would be waited for on its own.
template <int XP>
__device__ __forceinline__ void request_x(const GemvW8Args& a, int i0, GemvXRegsT<XP>& X) {
  const f32x4* xg = reinterpret_cast<const f32x4*>(a.x);
  const f32x4* ng = reinterpret_cast<const f32x4*>(a.norm_w ? a.norm_w : a.x);
  const int m4 = a.M >> 2, bd = blockDim.x;
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int idx = min(i0 + j * bd, m4 - 1);
    X.v[j] = xg[idx];
    if (a.norm_w) X.g[j] = ng[idx];
  }
}

// One pass of staging: the vectors of X (requested for pass start i0) go to LDS; returns
// their sum of squares.
template <int XP>
__device__ __forceinline__ float stage_pass(const GemvW8Args& a, f32x4* xs, int i0, bool norm,
                                            const GemvXRegsT<XP>& X) {
  const int n4 = a.chunks * 256, m4 = a.M >> 2, bd = blockDim.x;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int i = i0 + j * bd;
    if (i < n4) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (i < m4) {
        o = X.v[j];
        ss += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
        if (norm) o = f32x4{X.g[j].x * o.x, X.g[j].y * o.y, X.g[j].z * o.z, X.g[j].w * o.w};
      }
      const int c = i >> 8, l = (i >> 2) & 63, p = i & 3;
      xs[((c * 4 + p) << 6) + l] = o;
    }
  }
  return ss;
}

// Ends with one workgroup barrier.  The first pass (requested ahead of the weights) is straight-
// line code of its own: sharing a loop body with the later passes, whose loads are the youngest
// in flight, it would inherit their wait and sit until the weights had arrived too.
template <int XP>
__device__ __forceinline__ float commit_x(const GemvW8Args& a, f32x4* xs, float* red,
                                          GemvXRegsT<XP>& X) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwaves = blockDim.x >> 6, bd = blockDim.x;
  const int n4 = a.chunks * 256;
  const bool norm = a.norm_w != nullptr;
  float ss = stage_pass(a, xs, tid, norm, X);
  for (int i0 = tid + XP * bd; i0 < n4; i0 += XP * bd) {
    request_x(a, i0, X);
    ss += stage_pass(a, xs, i0, norm, X);
  }
  float rs = 1.0f;
  if (norm) {
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
  }
  __syncthreads();
  if (norm) {
    float tot = 0.f;
    for (int w = 0; w < nwaves; ++w) tot += red[w];
    rs = 1.0f / sqrtf(tot / (float)a.M + a.norm_eps);
  }
  return rs;
}

// One batch of weights in registers: R rows x U chunks (and the paired "up" rows for gate/up).
template <int R, int U, bool GU>
struct GemvBatch {
  i32x4 wv[R][U];
  float sc[R][U];
  i32x4 wv2[GU ? R : 1][GU ? U : 1];
  float sc2[GU ? R : 1][GU ? U : 1];
};

// Row totals and the epilogue operands of the R rows a wave is working on.
template <int R, bool GU>
struct GemvRows {
  float total[R];
  float total2[GU ? R : 1];
  float bias[R];                       // 0 without a bias vector
  float residual[R];                   // 0 without a residual vector
  float rope_s[R / 2 > 0 ? R / 2 : 1];  // sin / cos of each rotation pair
  float rope_c[R / 2 > 0 ? R / 2 : 1];
};

template <int R, int MODE>
__device__ __forceinline__ void begin_rows(GemvRows<R, MODE == GEMV_GATE_UP>& S) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    S.total[r] = 0.f;
    if constexpr (MODE == GEMV_GATE_UP) S.total2[r] = 0.f;
  }
}

// Requests the epilogue operands of a row set ahead of its weights, so no dependent load is left
// for the tail.  The values are only assigned here, never used: a use under the pointer test would
// put a full wait inside the branch, one memory round trip per operand before the first weight is
// requested.  (Requesting them behind the last weight batch instead measured the same on the
// token loop and 3-6 % slower on the large plain shapes.)
template <int R, int MODE>
__device__ __forceinline__ void request_epilogue(const GemvW8Args& a, int r0, int row_end, int pos,
                                                 GemvRows<R, MODE == GEMV_GATE_UP>& S) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = map_row<MODE>(a, min(r0 + r, row_end - 1));
    float b = 0.f, res = 0.f;
    if (a.bias) b = a.bias[row];
    if constexpr (MODE != GEMV_QKV_ROPE) {
      if (a.residual) res = a.residual[row];
    } else if ((r & 1) == 0) {
      // rope_kernel_cu (cuda/rope_kernel.cu:17-34 half split, :111-127 interleaved)
      const int in_head = row % a.head_size;
      const int tab = a.rope_variant == KX_ROPE_HALF_SPLIT ? in_head * 2 : in_head;
      S.rope_s[r / 2] = a.sin_cache[(size_t)pos * a.head_size + tab];
      S.rope_c[r / 2] = a.cos_cache[(size_t)pos * a.head_size + tab];
    }
    S.bias[r] = b;
    S.residual[r] = res;
  }
}

// Products of one batch against the staged x.
template <int R, int U, int MODE>
__device__ __forceinline__ void consume(const GemvW8Args& a, const f32x4* xs, int lane, int c0,
                                        const GemvBatch<R, U, MODE == GEMV_GATE_UP>& B,
                                        GemvRows<R, MODE == GEMV_GATE_UP>& S) {
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
        S.total[r] = fmaf(dot16(B.wv[r][u], x0, x1, x2, x3), B.sc[r][u], S.total[r]);
        if constexpr (MODE == GEMV_GATE_UP)
          S.total2[r] = fmaf(dot16(B.wv2[r][u], x0, x1, x2, x3), B.sc2[r][u], S.total2[r]);
      }
    }
  }
}

// Wave reduction and epilogue of R finished rows.
template <int R, int MODE>
__device__ __forceinline__ void finish_rows(const GemvW8Args& a, int r0, int row_end, int pos,
                                            float rs, int lane,
                                            GemvRows<R, MODE == GEMV_GATE_UP>& S) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    S.total[r] = wave_sum(S.total[r]) * rs;
    if constexpr (MODE == GEMV_GATE_UP) {
      const float up = wave_sum(S.total2[r]) * rs;
      // swiglu_kernel_cu (cuda/swiglu_kernel.cu:4-22): gate * sigmoid(gate) * up
      S.total[r] = S.total[r] * (1.0f / (1.0f + expf(-S.total[r]))) * up;
    }
  }
  if (lane != 0) return;
  if constexpr (MODE == GEMV_QKV_ROPE) {
#pragma unroll
    for (int r = 0; r < R; r += 2) {
      const int v = r0 + r;
      if (v < row_end) {  // row counts are even, so v+1 is valid too
        const int row0 = map_row<MODE>(a, v), row1 = map_row<MODE>(a, v + 1);
        float o0 = S.total[r] + S.bias[r], o1 = S.total[r + 1] + S.bias[r + 1];
        if (v < a.k1) {
          const float fci = S.rope_s[r / 2], fcr = S.rope_c[r / 2];
          const float t0 = o0 * fcr - o1 * fci;
          const float t1 = o0 * fci + o1 * fcr;
          o0 = t0;
          o1 = t1;
        }
        store_routed(a, row0, pos, o0);
        store_routed(a, row1, pos, o1);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = r0 + r;
      if (row < row_end) {
        float* dst;
        int rr;
        route(a, row, 0, dst, rr);
        store_out(a, dst + rr, S.total[r] + (S.bias[r] + S.residual[r]));
      }
    }
  }
}


}  // namespace kx
