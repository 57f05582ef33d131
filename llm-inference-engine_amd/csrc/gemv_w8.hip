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
#include <cstdlib>
#include "gemv_w8.hpp"

#include "gemv_w8_dev.hpp"

namespace kx {

// R rows per wave at a time (R even for GEMV_QKV_ROPE), U chunk loads per row in flight.
// For shapes where a wave gets a single batch (or a few).
template <int R, int U, int MODE>
__global__ void gemv_w8_kernel(const GemvW8Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs = reinterpret_cast<f32x4*>(smem);
  __shared__ float red[16];
  constexpr bool GU = MODE == GEMV_GATE_UP;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
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
  if constexpr (MODE == GEMV_QKV_ROPE) pos = a.pos_dev ? *a.pos_dev : a.pos_val;

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
    finish_rows<R, MODE>(a, r0, row_end, pos, rs, lane, S);
    r0 += nwaves * R;
  }
  bool staged = kPeelFirst;
  // The rotary sin/cos come from a table row that is new every token (an HBM miss behind a
  // dependent load of the position): requested in front of the weights they would be waited for
  // first, so that form asks for them behind the row set's last weight batch.
  constexpr bool kEpilogueLate = MODE == GEMV_QKV_ROPE;
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
    finish_rows<R, MODE>(a, r0, row_end, pos, rs, lane, S);
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
  if (lane == 0) {
    if (a.bias) total += a.bias[row];
    if (a.residual) total += a.residual[row];
    float* dst;
    int rr;
    route(a, row, 0, dst, rr);
    dst[rr] = total;
  }
}

// ---- several dependent GEMVs in one launch -------------------------------------------------
// A decode layer is a chain of matrix-vector products, each consuming the previous one's whole
// output: as separate launches every link pays the kernel boundary and the ramp of a cold memory
// pipeline.  But a GEMV's weights do not depend on its input vector: here one grid (one 16-wave
// workgroup per CU, all resident) walks the chain, and before it waits at the grid barrier that
// guards stage n+1's input each wave has already requested its first batch of stage n+1's weights,
// so HBM keeps streaming across the seam.  Stage outputs leave as write-through device-scope
// stores; the barrier is a release/acquire pair at device scope.
//
// Barrier state (zeroed once by the owner, then reused by every launch of the SAME grid size):
// the workgroups form kChainGroups groups by blockIdx % 8 (round-robin placement puts a group on
// one XCD: a speed matter only).  Arrivals count up on the group's word; the group's last arriver
// counts up the top word, waits for all groups, and publishes the group's generation, on which
// the rest of the group polls.  Counters only grow (compared modulo 2^32), so nothing is reset.
// Every spin is bounded: on giving up a workgroup raises the error word and runs on, so the grid
// always drains.
constexpr int kChainGroups = 8;
constexpr int kChainWordStride = 64;  // 256 bytes between the words
constexpr int kChainErrWord = (2 * kChainGroups + 1) * kChainWordStride;
constexpr size_t kChainStateBytes = (2 * kChainGroups + 2) * kChainWordStride * sizeof(unsigned);
constexpr int kChainMaxStages = 4;
constexpr int kChainThreads = 1024;
constexpr unsigned kChainSpinLimit = 1u << 18;  // polls of ~1 us each: a quarter of a second

struct GemvChainArgs {
  GemvW8Args st[kChainMaxStages];
  int32_t mode[kChainMaxStages];
  int32_t n;
  unsigned* bar;
  // optional timeline (tools/bench_chain.py): 16 wall-clock stamps (100 MHz) per workgroup
  unsigned long long* trace;
};
constexpr int kChainTraceSlots = 16;

__device__ __forceinline__ void chain_stamp(unsigned long long* trace, int slot) {
  if (trace && threadIdx.x == 0) trace[(size_t)blockIdx.x * kChainTraceSlots + slot] = wall_clock64();
}

__device__ __forceinline__ unsigned chain_ld(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct ChainBarrier {
  unsigned* cnt;   // this group's arrival counter
  unsigned* genw;  // this group's published generation
  unsigned* top;
  unsigned* err;
  unsigned gen, group_size, ngroups;

  __device__ void init(unsigned* base) {
    const unsigned grid = gridDim.x, g = blockIdx.x % kChainGroups;
    ngroups = grid < (unsigned)kChainGroups ? grid : (unsigned)kChainGroups;
    group_size = (grid - g + kChainGroups - 1) / kChainGroups;
    cnt = base + g * kChainWordStride;
    genw = base + (kChainGroups + g) * kChainWordStride;
    top = base + 2 * kChainGroups * kChainWordStride;
    err = base + kChainErrWord;
    // stable for the whole group until every one of its workgroups has arrived at a barrier
    gen = chain_ld(genw);
  }

  // All workgroups of the grid meet here.  Stores issued before it (write-through) are visible
  // to every load issued after it.
  __device__ void sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have landed
    __syncthreads();
    ++gen;
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      if (old + 1u == group_size * gen) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((int)(chain_ld(top) - ngroups * gen) < 0 && ++spins < kChainSpinLimit)
          __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        __hip_atomic_store(genw, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while ((int)(chain_ld(genw) - gen) < 0 && ++spins < kChainSpinLimit)
          __builtin_amdgcn_s_sleep(1);
      }
      if (spins >= kChainSpinLimit)
        __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // One cache invalidate per workgroup: the vector cache it acts on belongs to the CU, and the
    // workgroup barrier below orders it ahead of every other wave's loads.  (Issued by all 16
    // waves of all workgroups it cost ~40 us per barrier.)
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
  }
};

// One stage on this workgroup's share of the rows.  `wait` = the stage reads what an earlier
// stage of this launch wrote: the first weight batch is requested, THEN the grid meets, then x is
// staged (epilogue operands too: a residual may be an earlier stage's output).
template <int R, int U, int MODE>
__device__ __forceinline__ void chain_stage(const GemvW8Args& a, f32x4* xs, float* red,
                                            ChainBarrier& bar, bool wait,
                                            unsigned long long* trace, int slot0) {
  constexpr bool GU = MODE == GEMV_GATE_UP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int out_rows = GU ? a.swiglu_half : a.K;
  const int row_begin = blockIdx.x * a.rows_per_block;
  const int row_end = min(row_begin + a.rows_per_block, out_rows);
  const int col_lane = lane * 16;
  int r0 = row_begin + wave * R;
  const bool have = r0 < row_end;

  GemvBatch<R, U, GU> B;
  GemvRows<R, GU> S;
  if (have) {
    begin_rows<R, MODE>(S);
    load_batch<R, U, GU ? R : 1, GU ? U : 1, MODE>(a, r0, 0, row_end, col_lane, B.wv, B.sc, B.wv2, B.sc2);
  }
  chain_stamp(trace, slot0);
  if (wait) bar.sync();
  chain_stamp(trace, slot0 + 1);
  // 16 waves stage x: one or two vectors a thread cover up to 8 chunks in a single pass
  GemvXRegsT<2> X;
  request_x(a, threadIdx.x, X);
  const float rs = commit_x(a, xs, red, X);
  chain_stamp(trace, slot0 + 2);
  if (!have) return;
  int pos = 0;
  if constexpr (MODE == GEMV_QKV_ROPE) pos = a.pos_dev ? *a.pos_dev : a.pos_val;
  request_epilogue<R, MODE>(a, r0, row_end, pos, S);
  consume<R, U, MODE>(a, xs, lane, 0, B, S);
  for (int c0 = U; c0 < a.chunks; c0 += U) {
    load_batch<R, U, GU ? R : 1, GU ? U : 1, MODE>(a, r0, c0, row_end, col_lane, B.wv, B.sc, B.wv2, B.sc2);
    consume<R, U, MODE>(a, xs, lane, c0, B, S);
  }
  finish_rows<R, MODE>(a, r0, row_end, pos, rs, lane, S);
  for (r0 += nwaves * R; r0 < row_end; r0 += nwaves * R) {
    begin_rows<R, MODE>(S);
    request_epilogue<R, MODE>(a, r0, row_end, pos, S);
    for (int c0 = 0; c0 < a.chunks; c0 += U) {
      load_batch<R, U, GU ? R : 1, GU ? U : 1, MODE>(a, r0, c0, row_end, col_lane, B.wv, B.sc, B.wv2, B.sc2);
      consume<R, U, MODE>(a, xs, lane, c0, B, S);
    }
    finish_rows<R, MODE>(a, r0, row_end, pos, rs, lane, S);
  }
}

__global__ __launch_bounds__(kChainThreads) void gemv_chain_kernel(const GemvChainArgs c) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs = reinterpret_cast<f32x4*>(smem);
  __shared__ float red[16];
  ChainBarrier bar;
  bar.init(c.bar);
#pragma unroll
  for (int i = 0; i < kChainMaxStages; ++i) {
    if (i >= c.n) break;
    const GemvW8Args& a = c.st[i];
    switch (c.mode[i]) {
      case GEMV_GATE_UP: chain_stage<1, 2, GEMV_GATE_UP>(a, xs, red, bar, i > 0, c.trace, 3 * i); break;
      case GEMV_QKV_ROPE: chain_stage<2, 2, GEMV_QKV_ROPE>(a, xs, red, bar, i > 0, c.trace, 3 * i); break;
      default: chain_stage<1, 4, GEMV_PLAIN>(a, xs, red, bar, i > 0, c.trace, 3 * i); break;
    }
  }
  chain_stamp(c.trace, 3 * kChainMaxStages);
}

template <int R, int U, int MODE>
static int launch(const GemvW8Args& a, int threads, hipStream_t s) {
  const int out_rows = MODE == GEMV_GATE_UP ? a.swiglu_half : a.K;
  const int grid = (out_rows + a.rows_per_block - 1) / a.rows_per_block;
  const size_t lds = (size_t)a.chunks * 1024 * sizeof(float);
  if (lds > 64 * 1024) {
    static bool raised = false;  // per instantiation; the attribute is sticky
    if (!raised) {
      KX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_w8_kernel<R, U, MODE>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
      raised = true;
    }
  }
  hipLaunchKernelGGL((gemv_w8_kernel<R, U, MODE>), dim3(grid), dim3(threads), lds, s, a);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

static int ilog2(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

// Argument checks shared by the single-launch and the chained form; *fast = the 16-byte-chunk
// kernels apply.
static int gemv_w8_check(const GemvW8Args& a, int32_t group_size, bool* fast) {
  if (!a.x || !a.w || !a.scales || !a.y0) return KX_EINVAL;
  if (a.M <= 0 || a.K <= 0 || group_size <= 0) return KX_EINVAL;
  if (a.M % 4 != 0) return KX_EINVAL;  // matmul_kernel.cu:123
  if (((size_t)a.M * (size_t)a.K) % (size_t)group_size != 0) return KX_EINVAL;
  const bool pow2 = (group_size & (group_size - 1)) == 0;
  *fast = pow2 && group_size % 16 == 0 && a.M % group_size == 0 && a.M % 16 == 0 &&
          ((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.w % 16 == 0) &&
          (!a.norm_w || (uintptr_t)a.norm_w % 16 == 0) &&
          (size_t)((a.M + 1023) / 1024) * 4096 <= 160 * 1024 - 64;
  return KX_OK;
}

int gemv_w8_dispatch(GemvW8Args a, int32_t group_size, int mode, hipStream_t s) {
  bool fast = false;
  if (int rc = gemv_w8_check(a, group_size, &fast)) return rc;
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
  const size_t lds = (size_t)a.chunks * 4096;
  // waves per CU are bounded by how many x copies fit in the 160 KiB LDS
  int threads = 256;
  if (lds > 32 * 1024) threads = 512;
  if (lds > 80 * 1024) threads = 1024;
  const int out_rows = mode == GEMV_GATE_UP ? a.swiglu_half : a.K;
  // Few rows of many chunks (W2: 4096 x 14336, one block per CU): 16 waves of one row each
  // instead of 8 waves of two -- x is staged in one pass per thread and twice the waves hide each
  // other's round trips (13.0 -> 12.2 us) -- as long as all blocks are resident at once.
  bool tall_x = false;
  if (mode == GEMV_PLAIN && a.chunks > 8) {
    const int rpb1 = std::max((out_rows + 1023) / 1024, 16);
    const int blocks1 = (out_rows + rpb1 - 1) / rpb1;
    const int fit = std::max<int>(1, (int)((160 * 1024 - 64) / lds));
    tall_x = blocks1 <= 256 * fit;
    if (tall_x) threads = 1024;
  }
  // rows a wave keeps in flight together; the gate/up form already doubles the loads
  const int R = tall_x ? 1 : mode == GEMV_GATE_UP ? 1 : (a.chunks <= 2 ? 4 : 2);
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
  a.rows_per_block = rpb;
  // (R, U) per mode from a sweep on MI355X: more loads per wave than these lowers occupancy and
  // loses; two register sets per wave (software pipelining) lost 30 % against plain occupancy
  switch (mode) {
    case GEMV_GATE_UP:
      return a.chunks <= 2 ? launch<1, 2, GEMV_GATE_UP>(a, threads, s)
                           : launch<1, 4, GEMV_GATE_UP>(a, threads, s);
    case GEMV_QKV_ROPE:
      if (a.K % 2 || a.k0 % 2 || a.k1 % 2 || a.head_size % 2) return KX_EINVAL;
      return a.chunks <= 2 ? launch<4, 2, GEMV_QKV_ROPE>(a, threads, s)
                           : launch<2, 2, GEMV_QKV_ROPE>(a, threads, s);
    default:
      if (tall_x) return launch<1, 4, GEMV_PLAIN>(a, threads, s);
      return a.chunks <= 2 ? launch<4, 2, GEMV_PLAIN>(a, threads, s)
                           : launch<2, 4, GEMV_PLAIN>(a, threads, s);
  }
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

static int args_from_desc(const kx_gemv_w8_desc* d, kx::GemvW8Args& a);

extern "C" int kx_gemv_w8_fused_f32(const kx_gemv_w8_desc* d, void* stream) {
  if (!d) return KX_EINVAL;
  kx::GemvW8Args a{};
  if (int rc = args_from_desc(d, a)) return rc;
  return kx::gemv_w8_dispatch(a, d->group_size, d->mode, (hipStream_t)stream);
}

extern "C" size_t kx_gemv_chain_state_bytes(void) { return kx::kChainStateBytes; }

extern "C" int kx_gemv_w8_chain_f32(const kx_gemv_w8_desc* descs, int32_t n, void* state,
                                    size_t state_bytes, void* stream) {
  using namespace kx;
  if (!descs || n < 1 || n > kChainMaxStages) return KX_EINVAL;
  if (!state || state_bytes < kChainStateBytes || (uintptr_t)state % 256) return KX_EWORKSPACE;
  // one workgroup per CU: every workgroup is resident, so the in-launch barriers complete
  static int cus_of[64];
  int dev = 0;
  KX_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return KX_EUNSUPPORTED;
  if (!cus_of[dev]) {
    int cus = 0;
    KX_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) return KX_EUNSUPPORTED;
    KX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_chain_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    int per_cu = 0;
    KX_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &per_cu, reinterpret_cast<const void*>(&gemv_chain_kernel), kChainThreads, 160 * 1024 - 64));
    if (per_cu < 1) return KX_EUNSUPPORTED;
    cus_of[dev] = cus;
  }
  const int grid = cus_of[dev];
  GemvChainArgs c{};
  c.n = n;
  c.bar = static_cast<unsigned*>(state);
  // a state buffer with room for it, and KX_CHAIN_TRACE set: per-workgroup timeline behind the words
  if (state_bytes >= kChainStateBytes + (size_t)grid * kChainTraceSlots * 8 && getenv("KX_CHAIN_TRACE"))
    c.trace = reinterpret_cast<unsigned long long*>(static_cast<char*>(state) + kChainStateBytes);
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    const kx_gemv_w8_desc* d = descs + i;
    GemvW8Args& a = c.st[i];
    if (int rc = args_from_desc(d, a)) return rc;
    bool fast = false;
    if (int rc = gemv_w8_check(a, d->group_size, &fast)) return rc;
    if (!fast) return KX_EUNSUPPORTED;
    a.chunks = (a.M + 1023) / 1024;
    a.gs_shift = ilog2(d->group_size);
    a.groups_per_row = a.M / d->group_size;
    a.coherent_out = i + 1 < n;
    c.mode[i] = d->mode;
    if (d->mode == GEMV_QKV_ROPE && (a.K % 2 || a.k0 % 2 || a.k1 % 2 || a.head_size % 2))
      return KX_EINVAL;
    const int out_rows = d->mode == GEMV_GATE_UP ? a.swiglu_half : a.K;
    const int R = d->mode == GEMV_QKV_ROPE ? 2 : 1;
    int rpb = (out_rows + grid - 1) / grid;
    rpb = ((rpb + R - 1) / R) * R;
    a.rows_per_block = rpb;
    lds = std::max(lds, (size_t)a.chunks * 4096);
  }
  hipLaunchKernelGGL(gemv_chain_kernel, dim3(grid), dim3(kChainThreads), lds, (hipStream_t)stream, c);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

static int args_from_desc(const kx_gemv_w8_desc* d, kx::GemvW8Args& a) {
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
  return KX_OK;
}
