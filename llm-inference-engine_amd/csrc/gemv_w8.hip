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
// reads per weight load conflict-free, and the row sum is finished with DPP + permlane swaps.
// The first batch of weight loads is issued before x is staged so the HBM round trip and the
// L2 round trip for x overlap; bias, residual and rotation operands are requested before the
// chunk loop so no dependent load is left for the tail.
//
// Optional fusions used by the token loop (each replaces a separate launch of the reference):
//   - RMSNorm of x while staging it           (rmsnorm_kernel.cu:62-108, llama3.cpp:637-646):
//     g*x is staged and the scalar rsqrt(mean(x^2)+eps) multiplies the finished row sums
//   - bias / residual add in the epilogue      (matmul.cpp:76-79, llama3.cpp:751-752,787)
//   - [Wq;Wk;Wv] in one launch with RoPE applied to the finished q/k pairs and k, v written
//     straight into their KV-cache row        (llama3.cpp:648-677, rope_kernel.cu)
//   - [W1;W3] in one launch with SwiGLU         (llama3.cpp:761-776, swiglu_kernel.cu:4-22)
//
// Rounding order differs from the reference only in where the scale is applied: the 16
// products x*w of a lane are summed first and the group scale multiplies that partial sum.
#include <algorithm>
#include "gemv_w8.hpp"
#include "kx_step.hpp"

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

__device__ __forceinline__ void route(const GemvW8Args& a, int row, int pos, float*& dst, int& r, int& operand) {
  if (row < a.k0) {
    dst = a.y0;
    r = row;
    operand = GB_Y0;
  } else if (row < a.k1) {
    dst = a.y1 + (size_t)pos * a.kv_pos_stride;
    r = row - a.k0;
    operand = GB_Y1;
  } else {
    dst = a.y2 + (size_t)pos * a.kv_pos_stride;
    r = row - a.k1;
    operand = GB_Y2;
  }
}

// Row of the stacked q|k|v product -> its destination; k and v rows go to the position's cache row,
// rounded to fp16 when the cache holds fp16.
__device__ __forceinline__ void store_routed(const GemvW8Args& a, int row, int pos, float v) {
  if (a.kv_f16 && row >= a.k0) {
    const bool is_k = row < a.k1;
    _Float16* dst = reinterpret_cast<_Float16*>(is_k ? a.y1 : a.y2) + (size_t)pos * a.kv_pos_stride;
    *KX_CK(a, is_k ? GB_Y1 : GB_Y2, dst + (row - (is_k ? a.k0 : a.k1))) = (_Float16)v;
    return;
  }
  float* dst;
  int rr;
  [[maybe_unused]] int op;
  route(a, row, pos, dst, rr, op);
  *KX_CK(a, op, dst + rr) = v;
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
      wv[r][u] = __builtin_nontemporal_load(KX_CK(a, GB_W, reinterpret_cast<const i32x4*>(a.w + off)));
      sc[r][u] = __builtin_nontemporal_load(KX_CK(a, GB_S, a.scales + (size_t)row * a.groups_per_row +
                                                               (col >> a.gs_shift)));
      if constexpr (GU) {
        const size_t off2 = off + (size_t)a.swiglu_half * (size_t)a.M;
        wv2[r][u] = __builtin_nontemporal_load(KX_CK(a, GB_W, reinterpret_cast<const i32x4*>(a.w + off2)));
        sc2[r][u] = __builtin_nontemporal_load(KX_CK(
            a, GB_S, a.scales + (size_t)(row + a.swiglu_half) * a.groups_per_row + (col >> a.gs_shift)));
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
struct GemvXRegs {
  f32x4 v[kXPer];
  f32x4 g[kXPer];
};

// Unconditional loads with a clamped index: a load under a test would be waited for on its own.
__device__ __forceinline__ void request_x(const GemvW8Args& a, int i0, GemvXRegs& X) {
  const f32x4* xg = reinterpret_cast<const f32x4*>(a.x);
  const f32x4* ng = reinterpret_cast<const f32x4*>(a.norm_w ? a.norm_w : a.x);
  const int m4 = a.M >> 2, bd = blockDim.x;
#pragma unroll
  for (int j = 0; j < kXPer; ++j) {
    const int idx = min(i0 + j * bd, m4 - 1);
    X.v[j] = *KX_CK(a, GB_X, xg + idx);
    if (a.norm_w) X.g[j] = *KX_CK(a, GB_NORM, ng + idx);
  }
}

// One pass of staging: the kXPer vectors of X (requested for pass start i0) go to LDS; returns
// their sum of squares.
__device__ __forceinline__ float stage_pass(const GemvW8Args& a, f32x4* xs, int i0, bool norm,
                                            const GemvXRegs& X) {
  const int n4 = a.chunks * 256, m4 = a.M >> 2, bd = blockDim.x;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < kXPer; ++j) {
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
__device__ __forceinline__ float commit_x(const GemvW8Args& a, f32x4* xs, float* red, GemvXRegs& X) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwaves = blockDim.x >> 6, bd = blockDim.x;
  const int n4 = a.chunks * 256;
  const bool norm = a.norm_w != nullptr;
  float ss = stage_pass(a, xs, tid, norm, X);
  for (int i0 = tid + kXPer * bd; i0 < n4; i0 += kXPer * bd) {
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

// The position where an epilogue uses it: the empty statement keeps hipcc from working out
// everything that depends on it ahead of the loop (which would wait for it there).
__device__ __forceinline__ int late_pos(int pos) {
  asm volatile("" : "+v"(pos));
  return pos;
}

// Requests the epilogue operands of a row set (the loop forms do it behind the set's last weight
// batch, the written-out first set ahead of its weights).  The values are only assigned here, never
// used: a use under the pointer test would put a full wait inside the branch, one memory round trip
// per operand.
template <int R, int MODE>
__device__ __forceinline__ void request_epilogue(const GemvW8Args& a, int r0, int row_end, int pos,
                                                 GemvRows<R, MODE == GEMV_GATE_UP>& S) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = map_row<MODE>(a, min(r0 + r, row_end - 1));
    float b = 0.f, res = 0.f;
    if (a.bias) b = *KX_CK(a, GB_BIAS, a.bias + row);
    if constexpr (MODE != GEMV_QKV_ROPE) {
      if (a.residual) res = *KX_CK(a, GB_RES, a.residual + row);
    } else if ((r & 1) == 0) {
      // rope_kernel_cu (cuda/rope_kernel.cu:17-34 half split, :111-127 interleaved).  Only q and k
      // rows are rotated.  A v row keeps its own index inside the head (map_row leaves it alone), so
      // twice that index can reach past the table row -- and past the table at the last position:
      // such rows read entry 0, which nothing uses.
      const int in_head = row < a.k1 ? row % a.head_size : 0;
      const int tab = a.rope_variant == KX_ROPE_HALF_SPLIT ? in_head * 2 : in_head;
      const int p = late_pos(pos);
      S.rope_s[r / 2] = *KX_CK(a, GB_SIN, a.sin_cache + ((size_t)p * a.head_size + tab));
      S.rope_c[r / 2] = *KX_CK(a, GB_COS, a.cos_cache + ((size_t)p * a.head_size + tab));
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
// PICK: lane 0 also keeps the largest finished value and its row (first maximum wins)
template <int R, int MODE, bool PICK = false>
__device__ __forceinline__ void finish_rows(const GemvW8Args& a, int r0, int row_end, int pos,
                                            float rs, int lane,
                                            GemvRows<R, MODE == GEMV_GATE_UP>& S, float& best, int& bi) {
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
    pos = late_pos(pos);
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
        [[maybe_unused]] int op;
        route(a, row, 0, dst, rr, op);
        const float v = S.total[r] + (S.bias[r] + S.residual[r]);
        *KX_CK(a, op, dst + rr) = v;
        if constexpr (PICK) argmax_merge(best, bi, v, row);
      }
    }
  }
}

// R rows per wave at a time (R even for GEMV_QKV_ROPE), U chunk loads per row in flight.
// For shapes where a wave gets a single batch (or a few).
template <int R, int U, int MODE, bool PICK = false>
__device__ __forceinline__ void gemv_w8_body(const GemvW8Args& a, float& best, int& bi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs = reinterpret_cast<f32x4*>(smem);
  __shared__ float red[16];
  constexpr bool GU = MODE == GEMV_GATE_UP;
  const int lane = threadIdx.x & 63;
  // readfirstlane: the wave index is uniform, and hipcc must know it -- with a lane-dependent
  // index the "this wave has no rows" test below is a divergent branch, both sides of which are laid
  // out one behind the other under exec masks; the loads of the idle side then count as pending on
  // the working side and every request there is preceded by a full wait (x, then the bias, then the
  // residual, then the first weights: round trip after round trip instead of all at once)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int out_rows = GU ? a.swiglu_half : a.K;
  const int row_begin = blockIdx.x * a.rows_per_block;
  const int row_end = min(row_begin + a.rows_per_block, out_rows);
  const int col_lane = lane * 16;
  const int r_first = row_begin + wave * R;

  // A wave without rows only helps staging x.  The others request x, then their first batch of
  // weights, and only then stage x (inside the loop below, first trip), so the HBM round trip of
  // the weights overlaps the staging and the optional normalisation of x.
  GemvXRegs X;
  request_x(a, threadIdx.x, X);
  if (r_first >= row_end) {
    commit_x(a, xs, red, X);
    return;
  }
  float rs = 1.0f;
  int pos = 0;
  if constexpr (MODE == GEMV_QKV_ROPE) {
    // The position is fetched with a VECTOR load (the lane-dependent zero hides the uniform
    // address): as a scalar load hipcc waits for it together with the kernel arguments and works
    // out the table and cache-row addresses ahead of the loop -- a memory round trip in front of
    // the first weight request.  Behind x and ahead of the weights in the vector queue it costs
    // nothing: it has long arrived when the epilogue asks for it (see late_pos).
    pos = a.pos_val;
    if (a.pos_dev) {
      int lane_zero;
      asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
      pos = *KX_CK(a, GB_POS, a.pos_dev + lane_zero);
    }
  }

  GemvBatch<R, U, GU> B;
  GemvRows<R, GU> S;
  // Narrow matrices (at most two chunks, the R = 4 / U = 2 forms): many row sets per wave, and the
  // registers holding the requested x must not stay live across the loop over them (it costs
  // a quarter of the rate on a 128256 x 2048 matrix), so the first row set is written out on its
  // own.  Wide matrices: a single loop with a staged flag is the faster form (measured both ways).
  constexpr bool kPeelFirst = GU ? U == 2 : R == 4;
  int r0 = r_first;
  if constexpr (kPeelFirst) {
    begin_rows<R, MODE>(S);
    request_epilogue<R, MODE>(a, r0, row_end, pos, S);
    load_batch<R, U, GU ? R : 1, GU ? U : 1, MODE>(a, r0, 0, row_end, col_lane, B.wv, B.sc, B.wv2, B.sc2);
    rs = commit_x(a, xs, red, X);
    consume<R, U, MODE>(a, xs, lane, 0, B, S);
    for (int c0 = U; c0 < a.chunks; c0 += U) {
      load_batch<R, U, GU ? R : 1, GU ? U : 1, MODE>(a, r0, c0, row_end, col_lane, B.wv, B.sc, B.wv2,
                                                    B.sc2);
      consume<R, U, MODE>(a, xs, lane, c0, B, S);
    }
    finish_rows<R, MODE, PICK>(a, r0, row_end, pos, rs, lane, S, best, bi);
    r0 += nwaves * R;
  }
  bool staged = kPeelFirst;
  // Epilogue operands (bias, residual; the rotary sin/cos of a table row that is new every token)
  // are asked for behind the row set's last weight batch: requested in front of the weights, hipcc
  // waits for them before the first weight request of the loop (decode 496.0 -> 497.3 tok/s, the
  // 4096 x 4096 launch 5.44 -> 5.19 us on one box).
  constexpr bool kEpilogueLate = true;
  for (; r0 < row_end; r0 += nwaves * R) {
    begin_rows<R, MODE>(S);
    if constexpr (!kEpilogueLate) request_epilogue<R, MODE>(a, r0, row_end, pos, S);
    for (int c0 = 0; c0 < a.chunks; c0 += U) {
      load_batch<R, U, GU ? R : 1, GU ? U : 1, MODE>(a, r0, c0, row_end, col_lane, B.wv, B.sc, B.wv2,
                                                    B.sc2);
      if constexpr (kEpilogueLate) {
        if (c0 + U >= a.chunks) request_epilogue<R, MODE>(a, r0, row_end, pos, S);
      }
      if constexpr (!kPeelFirst) {
        if (!staged) {
          rs = commit_x(a, xs, red, X);
          staged = true;
        }
      }
      consume<R, U, MODE>(a, xs, lane, c0, B, S);
    }
    finish_rows<R, MODE, PICK>(a, r0, row_end, pos, rs, lane, S, best, bi);
  }
}

template <int R, int U, int MODE>
__global__ void gemv_w8_kernel(const GemvW8Args a) {
  float best = -INFINITY;
  int bi = 0x7fffffff;
  gemv_w8_body<R, U, MODE>(a, best, bi);
}

// The classifier GEMV that also ends the decode step (step_pick_kernel's work without its launch
// and without reading the logits back): lane 0 of every wave keeps the best of the rows it
// finished, the workgroup's best pair goes out with write-through stores, an arrival ticket follows,
// and the workgroup that draws the last ticket merges all pairs, does the step's bookkeeping and
// leaves the next step's embedding row -- which may be this GEMV's own input vector: every other
// workgroup has drawn its ticket, i.e. is done with x, by then.
template <int R, int U>
__global__ void gemv_w8_pick_kernel(const GemvW8Args a, const StepArgs st) {
  __shared__ float sv[16];
  __shared__ int si[16];
  __shared__ int sm_last, sm_tok;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  gemv_w8_body<R, U, GEMV_PLAIN, true>(a, best, bi);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) {
    sv[wave] = best;
    si[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w) argmax_merge(best, bi, sv[w], si[w]);
    __hip_atomic_store(KX_CK(st, SB_PV, st.pv + blockIdx.x), best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(KX_CK(st, SB_PI, st.pi + blockIdx.x), bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // both stores (and this thread's logits) acknowledged before the ticket
    sm_last = atomicInc(KX_CK(st, SB_TICKET, st.ticket), gridDim.x - 1u) == gridDim.x - 1u;
  }
  __syncthreads();
  if (!sm_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler order only: the loads below are device-scope
  best = -INFINITY;
  bi = 0x7fffffff;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x)
    argmax_merge(best, bi, __hip_atomic_load(KX_CK(st, SB_PV, st.pv + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                 __hip_atomic_load(KX_CK(st, SB_PI, st.pi + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __syncthreads();  // sv / si are reused
  block_argmax(best, bi, sv, si);
  if (threadIdx.x == 0) step_bookkeeping(st, bi, &sm_tok);
  __syncthreads();
  step_embed(st, sm_tok);
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
    const int32_t wv = *KX_CK(a, GB_W, reinterpret_cast<const int32_t*>(a.w + base + i));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t idx = base + (size_t)(i + e);
      const float s = *KX_CK(a, GB_S, a.scales + idx / (size_t)group_size);
      total = fmaf(*KX_CK(a, GB_X, a.x + (i + e)) * s, (float)(int8_t)(wv >> (8 * e)), total);
    }
  }
  total = wave_sum(total);
  if (lane == 0) {
    if (a.bias) total += *KX_CK(a, GB_BIAS, a.bias + row);
    if (a.residual) total += *KX_CK(a, GB_RES, a.residual + row);
    float* dst;
    int rr;
    [[maybe_unused]] int op;
    route(a, row, 0, dst, rr, op);
    *KX_CK(a, op, dst + rr) = total;
  }
}

template <int R, int U, int MODE>
static int launch(const GemvW8Args& a, int threads, hipStream_t s) {
  const int out_rows = MODE == GEMV_GATE_UP ? a.swiglu_half : a.K;
  const int grid = (out_rows + a.rows_per_block - 1) / a.rows_per_block;
  const size_t lds = (size_t)a.chunks * 1024 * sizeof(float);
  if (lds > 64 * 1024) {
    static OncePerDevice raised;  // per instantiation; the attribute is sticky
    if (int rc = raised.raise(reinterpret_cast<const void*>(&gemv_w8_kernel<R, U, MODE>), 160 * 1024 - 64)) return rc;
  }
  hipLaunchKernelGGL((gemv_w8_kernel<R, U, MODE>), dim3(grid), dim3(threads), lds, s, a);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

// Workgroup shape of the fast path for a matrix of out_rows x M (out_rows: gate/up PAIRS in the
// gate/up form): threads per workgroup, rows a wave keeps in flight together (R), and the rows a
// workgroup owns -- workgroup b reads rows [b * rows_per_block, (b + 1) * rows_per_block).
struct GemvGeometry {
  int threads, R, rows_per_block;
  bool tall_x;
};
static GemvGeometry gemv_geometry(int mode, int M, int out_rows) {
  const int chunks = (M + 1023) / 1024;
  const size_t lds = (size_t)chunks * 4096;
  // waves per CU are bounded by how many x copies fit in the 160 KiB LDS
  int threads = 256;
  if (lds > 32 * 1024) threads = 512;
  if (lds > 80 * 1024) threads = 1024;
  // Few rows of many chunks (W2: 4096 x 14336, one block per CU): 16 waves of one row each
  // instead of 8 waves of two -- x is staged in one pass per thread and twice the waves hide each
  // other's round trips (13.0 -> 12.2 us) -- as long as all blocks are resident at once.
  bool tall_x = false;
  if (mode == GEMV_PLAIN && chunks > 8) {
    const int rpb1 = std::max((out_rows + 1023) / 1024, 16);
    const int blocks1 = (out_rows + rpb1 - 1) / rpb1;
    const int fit = std::max<int>(1, (int)((160 * 1024 - 64) / lds));
    tall_x = blocks1 <= 256 * fit;
    if (tall_x) threads = 1024;
  }
  // Few rows in all (fewer than 8 waves per CU at the usual rows per wave; Llama-3.2-1B: Wo 2048 x 2048,
  // W2 2048 x 8192): one row per wave as well, in smaller workgroups -- twice to four times the
  // waves hide each other's latency (3.9 -> 3.2 us and 6.8 -> 5.5 us on those two).
  if (mode == GEMV_PLAIN && !tall_x && out_rows / (chunks <= 2 ? 4 : 2) < 2048) {
    tall_x = true;
    threads = chunks >= 2 ? 512 : 256;
  }
  // rows a wave keeps in flight together; the gate/up form already doubles the loads.  The fused
  // QKV form needs both rows of a rotation pair in one wave and is best at exactly two (3072 x 2048:
  // 5.6 us with four rows per wave, 4.3 us with two).
  const int R = tall_x ? 1 : mode == GEMV_GATE_UP ? 1 : mode == GEMV_QKV_ROPE ? 2 : (chunks <= 2 ? 4 : 2);
  const int per_iter = (threads / 64) * R;
  // aim for ~4 workgroups per CU on big matrices, one pass on small ones
  const int target_blocks = 256 * 4;
  int rpb = (out_rows + target_blocks - 1) / target_blocks;
  // Every wave gets work (at least one row set per wave); beyond that only whole row sets, so the
  // grid stays a multiple of the CU count where the row count allows it: 14336 gate/up pairs are
  // 1024 blocks of 14 (waves take 4, 4, 3, 3), not 896 blocks of 16 that leave half of the CUs
  // with a fourth block to finish.
  if (rpb < per_iter) rpb = per_iter;
  rpb = ((rpb + R - 1) / R) * R;
  return GemvGeometry{threads, R, rpb, tall_x};
}

template <int R, int U>
static int launch_pick(const GemvW8Args& a, const StepArgs& st, int threads, hipStream_t s) {
  const int grid = (a.K + a.rows_per_block - 1) / a.rows_per_block;
  if (grid > kStepSlots) return KX_EUNSUPPORTED;  // one pair slot per workgroup
  const size_t lds = (size_t)a.chunks * 1024 * sizeof(float);
  if (lds > 64 * 1024) {
    static OncePerDevice raised;  // per instantiation; the attribute is sticky
    if (int rc = raised.raise(reinterpret_cast<const void*>(&gemv_w8_pick_kernel<R, U>), 160 * 1024 - 64)) return rc;
  }
  hipLaunchKernelGGL((gemv_w8_pick_kernel<R, U>), dim3(grid), dim3(threads), lds, s, a, st);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

static int ilog2(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

int gemv_w8_dispatch(GemvW8Args a, int32_t group_size, int mode, hipStream_t s, const StepArgs* pick) {
  if (!a.x || !a.w || !a.scales || !a.y0) return KX_EINVAL;
  if (a.M <= 0 || a.K <= 0 || group_size <= 0) return KX_EINVAL;
  if (a.M % 4 != 0) return KX_EINVAL;  // matmul_kernel.cu:123
  if (((size_t)a.M * (size_t)a.K) % (size_t)group_size != 0) return KX_EINVAL;
  KX_BSET(a, GB_X, a.x); KX_BSET(a, GB_W, a.w); KX_BSET(a, GB_S, a.scales);
  KX_BSET(a, GB_Y0, a.y0); KX_BSET(a, GB_Y1, a.y1); KX_BSET(a, GB_Y2, a.y2);
  KX_BSET(a, GB_BIAS, a.bias); KX_BSET(a, GB_RES, a.residual); KX_BSET(a, GB_NORM, a.norm_w);
  KX_BSET(a, GB_SIN, a.sin_cache); KX_BSET(a, GB_COS, a.cos_cache); KX_BSET(a, GB_POS, a.pos_dev);
  const bool pow2 = (group_size & (group_size - 1)) == 0;
  const bool fast = pow2 && group_size % 16 == 0 && a.M % group_size == 0 && a.M % 16 == 0 &&
                    ((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.w % 16 == 0) &&
                    (!a.norm_w || (uintptr_t)a.norm_w % 16 == 0) &&
                    (size_t)((a.M + 1023) / 1024) * 4096 <= 160 * 1024 - 64;
  if (pick && (!fast || mode != GEMV_PLAIN || a.k0 != a.K)) return KX_EUNSUPPORTED;
  if (!fast) {
    if (mode != GEMV_PLAIN || a.norm_w) return KX_EUNSUPPORTED;
    const int wpb = 4;
    hipLaunchKernelGGL(gemv_w8_generic_kernel, dim3((a.K + wpb - 1) / wpb), dim3(wpb * 64), 0, s, a,
                       group_size);
    KX_LAUNCH_CHECK();
    return KX_OK;
  }
  a.chunks = (a.M + 1023) / 1024;
  a.gs_shift = ilog2(group_size);
  a.groups_per_row = a.M / group_size;
  const int out_rows = mode == GEMV_GATE_UP ? a.swiglu_half : a.K;
  const GemvGeometry geo = gemv_geometry(mode, a.M, out_rows);
  const int threads = geo.threads;
  const bool tall_x = geo.tall_x;
  a.rows_per_block = geo.rows_per_block;
  // (R, U) per mode from a sweep on MI355X: more loads per wave than these lowers occupancy and
  // loses; two register sets per wave (software pipelining) lost 30 % against plain occupancy
  switch (mode) {
    case GEMV_GATE_UP:
      return a.chunks <= 2 ? launch<1, 2, GEMV_GATE_UP>(a, threads, s)
                           : launch<1, 4, GEMV_GATE_UP>(a, threads, s);
    case GEMV_QKV_ROPE:
      if (a.K % 2 || a.k0 % 2 || a.k1 % 2 || a.head_size % 2) return KX_EINVAL;
      return launch<2, 2, GEMV_QKV_ROPE>(a, threads, s);
    default:
      if (pick) {
        if (tall_x) return launch_pick<1, 4>(a, *pick, threads, s);
        return a.chunks <= 2 ? launch_pick<4, 2>(a, *pick, threads, s) : launch_pick<2, 4>(a, *pick, threads, s);
      }
      if (tall_x) return launch<1, 4, GEMV_PLAIN>(a, threads, s);
      return a.chunks <= 2 ? launch<4, 2, GEMV_PLAIN>(a, threads, s)
                           : launch<2, 4, GEMV_PLAIN>(a, threads, s);
  }
}

}  // namespace kx

extern "C" int32_t kx_gemv_w8_block_rows(int32_t M, int32_t K, int32_t group_size) {
  if (M <= 0 || K <= 0 || group_size <= 0 || M % 16 || M % group_size || group_size % 16 ||
      (group_size & (group_size - 1)) || (size_t)((M + 1023) / 1024) * 4096 > 160 * 1024 - 64)
    return 0;
  return kx::gemv_geometry(kx::GEMV_PLAIN, M, K).rows_per_block;
}

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
  return kx::gemv_w8_dispatch(a, group_size, kx::GEMV_PLAIN, (hipStream_t)stream);
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
  return kx::gemv_w8_dispatch(a, group_size, kx::GEMV_GATE_UP, (hipStream_t)stream);
}

extern "C" int kx_gemv_w8_fused_f32(const kx_gemv_w8_desc* d, void* stream) {
  if (!d) return KX_EINVAL;
  kx::GemvW8Args a{};
  a.x = d->x; a.w = d->w; a.scales = d->scales;
  a.norm_w = d->norm_w; a.norm_eps = d->norm_eps;
  a.M = d->M; a.K = d->K;
  a.bias = d->bias;
  int32_t k0 = d->k0, k1 = d->k1;
  a.y0 = d->y0; a.y1 = d->y1 ? d->y1 : d->y0; a.y2 = d->y2 ? d->y2 : d->y0;
  if (!d->y1 && !d->y2) { k0 = d->K; k1 = d->K; }
  switch (d->mode) {
    case KX_GEMV_PLAIN:
      a.residual = d->residual;
      break;
    case KX_GEMV_GATE_UP:
      if (d->K % 2 || d->residual) return KX_EINVAL;
      a.swiglu_half = d->K / 2;
      k0 = k1 = a.swiglu_half;
      a.y1 = a.y2 = a.y0;
      break;
    case KX_GEMV_QKV_ROPE:
      if (!d->sin_cache || !d->cos_cache || d->head_size <= 0 || d->residual) return KX_EINVAL;
      if (k0 % d->head_size || (k1 - k0) % d->head_size) return KX_EINVAL;
      if (!d->pos_dev && d->pos < 0) return KX_EINVAL;
      a.sin_cache = d->sin_cache; a.cos_cache = d->cos_cache;
      a.pos_dev = d->pos_dev; a.pos_val = d->pos;
      a.rope_variant = d->rope_variant; a.head_size = d->head_size;
      a.kv_pos_stride = d->kv_pos_stride;
      a.kv_f16 = d->kv_f16;
      break;
    default:
      return KX_EINVAL;
  }
  if (k0 < 0 || k1 < k0 || k1 > d->K) return KX_EINVAL;
  a.k0 = k0; a.k1 = k1;
  return kx::gemv_w8_dispatch(a, d->group_size, d->mode, (hipStream_t)stream);
}

extern "C" int kx_gemv_w8_classify_pick_f32(const kx_gemv_w8_desc* d, int32_t* token_dev, int32_t* pos_dev,
                                            int32_t* history_dev, int32_t history_len,
                                            int32_t write_history, const float* table, int32_t dim,
                                            int32_t vocab_size, float* x_out, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  if (!d || !token_dev || !pos_dev) return KX_EINVAL;
  if (d->mode != KX_GEMV_PLAIN || d->residual || d->y1 || d->y2) return KX_EINVAL;
  if (table && (!x_out || !history_dev || dim <= 0 || vocab_size <= 0)) return KX_EINVAL;
  if (!workspace || workspace_bytes < kx::kStepWorkspaceBytes) return KX_EWORKSPACE;
  kx::GemvW8Args a{};
  a.x = d->x; a.w = d->w; a.scales = d->scales;
  a.norm_w = d->norm_w; a.norm_eps = d->norm_eps;
  a.M = d->M; a.K = d->K;
  a.bias = d->bias;
  a.y0 = a.y1 = a.y2 = d->y0;
  a.k0 = a.k1 = d->K;
  kx::StepArgs st{};
  st.logits = d->y0; st.n = d->K;
  st.pv = static_cast<float*>(workspace);
  st.pi = reinterpret_cast<int32_t*>(st.pv + kx::kStepSlots);
  st.ticket = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + kx::kStepTicketOff);
  st.token = token_dev; st.pos = pos_dev; st.history = history_dev; st.history_len = history_len;
  st.write_history = write_history != 0;
  st.table = table; st.dim = dim; st.vocab = vocab_size; st.x_out = x_out;
  using namespace kx;
  KX_BSET(st, SB_LOGITS, st.logits); KX_BSET(st, SB_PV, st.pv); KX_BSET(st, SB_PI, st.pi); KX_BSET(st, SB_TICKET, st.ticket);
  KX_BSET(st, SB_CAND, nullptr); KX_BSET(st, SB_TOKEN, st.token); KX_BSET(st, SB_POS, st.pos);
  KX_BSET(st, SB_HISTORY, st.history); KX_BSET(st, SB_TABLE, st.table); KX_BSET(st, SB_XOUT, st.x_out);
  return kx::gemv_w8_dispatch(a, d->group_size, kx::GEMV_PLAIN, (hipStream_t)stream, &st);
}
