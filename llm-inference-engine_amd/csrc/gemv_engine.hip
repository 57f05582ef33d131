// Chained W8 GEMVs of the decode step in ONE persistent launch, gfx950.
//
// The token loop is a chain of dependent GEMVs (llama3.cpp:648-788: Wq|Wk|Wv, attention, Wo, W1|W3,
// W2, each a MatmulLayer::forward over matmul_kernel_cu_fp32int8, cuda/matmul_kernel.cu:56-87).  As
// separate launches every one of them pays a kernel boundary and the ramp of a cold memory pipeline
// (2.7 us per launch on top of bytes / 6.9 TB/s).  This kernel runs TWO consecutive GEMVs of the
// chain -- [Wo -> W1|W3] and [W2 -> next layer's Wq|Wk|Wv, or the classifier] -- so that the
// weight stream never stops at the seam:
//   - one workgroup per CU, resident for the whole launch: wave 4 is the LOADER, waves 0-3 are
//     CONSUMERS.  The loader walks the CU's share of both matrices in stream order and brings it
//     into an LDS ring by LDS-DMA (global_load_lds, 1 KiB per instruction, non-temporal).  It never
//     waits for a hand-off: when the consumers stop at the seam, it keeps filling the ring with
//     the second matrix.
//   - a CU owns blocks of 8 (virtual) rows; inside a block each consumer wave owns a row pair and
//     the stream interleaves the four pairs column step by column step ([scales of the step][a0 b0
//     a1 b1 a2 b2 a3 b3], 1 KiB pieces), so all four consumers work next to each other in the ring.
//     A pair is two plain rows, a gate row and its up row (SwiGLU, llama3.cpp:761-776), or the two
//     rows of a rotation pair (RoPE, rope_kernel.cu) -- the rows that meet in the epilogue.
//   - ring flow control is two counters in LDS: landed DMA instructions (the loader waits with a
//     counted vmcnt, so up to 40 KiB per CU stay in flight) and column steps finished per consumer.
//   - the seam: the first GEMV's outputs leave as 8-byte {value, tag} granules, each written by
//     ONE write-through (sc1) store; every CU's consumers sweep a quarter of the granule array
//     each with sc1 loads until all tags match, staging the vector (times the RMSNorm weight,
//     rmsnorm_kernel.cu:62-108) in LDS.  No flag, no counter, no fence, no grid barrier: a granule
//     is its own flag.  tag = (position + 1) * 128 + chain id, so a value from an earlier token or
//     another seam never matches; the host clears the array whenever the position is set back.
//   - arithmetic: that of the one-launch GEMV (gemv_w8.hip): 16 products per lane summed, times
//     the group scale, fp32 throughout; epilogues bias / residual / RoPE + KV placement / SwiGLU.
// Every poll is bounded by wall-clock time; a poll that gives up raises a flag in the workspace
// (kx_gemv_w8_chain_status) and the launch runs to its end instead of hanging the queue.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#define KX_FILE_ID 9
#include "kx_common.hpp"

namespace kx {

constexpr int EN_MAX_NCW = 8;             // consumer waves: 4 or 8 (a row pair each)
constexpr int EN_DEPTH_THIN = 12;         // DMA instructions in flight while the loader's own CU sweeps granules
constexpr int EN_CTRL = 256;              // bytes of counters at the head of LDS
constexpr unsigned long long EN_WAIT_TICKS = 100000000ull;  // 1 s of the 100 MHz wall counter
constexpr int EN_MAX_OPS = 2;

struct EngineOp {
  const int8_t* w;
  const float* scales;
  const float* x;        // op 0: the input vector in memory; later ops: gathered from the seam
  const float* norm_w;   // RMSNorm weight applied to the input (null: none)
  float norm_eps;
  float* y0; float* y1; float* y2;
  const float* bias;
  const float* residual;
  int32_t k0, k1;
  int32_t M, K;
  int32_t mode;          // KX_GEMV_*
  int32_t ppr;           // column steps per row: M / 1024
  int32_t blocks;        // K / 8
  int32_t half;          // GATE_UP: hidden units (K / 2)
  const float* sin_cache; const float* cos_cache;
  int32_t rope_variant, head_size, hs_shift, kv_pos_stride, kv_f16;
  int32_t xs_off;        // byte offset of the op's staged input in LDS
};

struct EngineArgs {
  EngineOp op[EN_MAX_OPS];
  int32_t nops;
  unsigned long long* granules;  // max(K of op 0) entries
  int32_t* status;               // [0]: a bounded wait gave up
  const int32_t* pos_dev;
  int32_t pos_val;
  int32_t chain_id;
  int32_t variant;               // probe switches (KX_CHAIN_VARIANT)
  int32_t trace;                 // status[1..]: summed wait counts and times (KX_CHAIN_TRACE; see the host entry)
  int32_t ring_steps;            // column steps the weight ring holds
  int32_t sring_off, ring_off;   // byte offsets in LDS
  KX_BFIELD(16)
};
enum EngineOperand { EB_W0, EB_S0, EB_X0, EB_N0, EB_Y00, EB_Y01, EB_Y02, EB_W1, EB_S1, EB_N1, EB_Y10, EB_Y11, EB_Y12,
                     EB_GRAN, EB_RES, EB_MISC };

using en_lds_ptr_t = __attribute__((address_space(3))) void*;

// virtual row (the order the stream walks) -> matrix row
__device__ __forceinline__ int en_row(const EngineOp& o, int v) {
  if (o.mode == KX_GEMV_GATE_UP) return (v >> 1) + (v & 1) * o.half;
  if (o.mode == KX_GEMV_QKV_ROPE && o.rope_variant == KX_ROPE_HALF_SPLIT && v < o.k1) {
    const int head = v >> o.hs_shift, i = v & (o.head_size - 1);
    return (head << o.hs_shift) + (i >> 1) + (i & 1) * (o.head_size >> 1);
  }
  return v;
}

__device__ __forceinline__ float en_dot16(const i32x4 wv, const f32x4 x0, const f32x4 x1, const f32x4 x2,
                                          const f32x4 x3) {
  float a = 0.f;
#define KX_B(word, sh) ((float)(int8_t)((word) >> (sh)))
  a = fmaf(x0.x, KX_B(wv.x, 0), a);  a = fmaf(x0.y, KX_B(wv.x, 8), a);
  a = fmaf(x0.z, KX_B(wv.x, 16), a); a = fmaf(x0.w, KX_B(wv.x, 24), a);
  a = fmaf(x1.x, KX_B(wv.y, 0), a);  a = fmaf(x1.y, KX_B(wv.y, 8), a);
  a = fmaf(x1.z, KX_B(wv.y, 16), a); a = fmaf(x1.w, KX_B(wv.y, 24), a);
  a = fmaf(x2.x, KX_B(wv.z, 0), a);  a = fmaf(x2.y, KX_B(wv.z, 8), a);
  a = fmaf(x2.z, KX_B(wv.z, 16), a); a = fmaf(x2.w, KX_B(wv.z, 24), a);
  a = fmaf(x3.x, KX_B(wv.w, 0), a);  a = fmaf(x3.y, KX_B(wv.w, 8), a);
  a = fmaf(x3.z, KX_B(wv.w, 16), a); a = fmaf(x3.w, KX_B(wv.w, 24), a);
#undef KX_B
  return a;
}

// LDS control words (byte offsets from the start of LDS, where the control block sits).  They are
// read and written with DS instructions spelled out in asm: the compiler orders every LDS access
// it can see in a wave behind ALL of that wave's pending LDS-DMA (a vmcnt(0) before each), which
// would let the loader keep one step in flight instead of DEPTH instructions.
enum : unsigned {
  EC_FILLED = 0,                                // [4] per loader wave: its DMA instructions that have landed
  EC_CONSUMED = 16,                              // [EN_MAX_NCW] column steps (kernel-wide index) finished per consumer
  EC_SWEEPING = EC_CONSUMED + 4 * EN_MAX_NCW,   // consumers of this CU that are inside a granule sweep
  EC_X_ARRIVE = EC_SWEEPING + 4,                // [EN_MAX_OPS] consumers that have staged their share of an op's input
  EC_SS_PART = EC_X_ARRIVE + 4 * EN_MAX_OPS,    // [EN_MAX_OPS][EN_MAX_NCW] partial sums of squares (float bits)
  EC_GAVE_UP = EC_SS_PART + 4 * EN_MAX_OPS * EN_MAX_NCW,
  EC_END = EC_GAVE_UP + 4
};
static_assert(EC_END <= EN_CTRL, "control block");

__device__ __forceinline__ unsigned ec_ld(unsigned off) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
  return v;
}
__device__ __forceinline__ void ec_st(unsigned off, unsigned v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(off), "v"(v) : "memory");
}
__device__ __forceinline__ void ec_add(unsigned off, unsigned v) {
  asm volatile("ds_add_u32 %0, %1" ::"v"(off), "v"(v) : "memory");
}

// float index of the input vector -> float offset in its LDS stage: chunk c of 1024 columns,
// lane l holds columns 16 l .. 16 l + 15 of the chunk as four 16-byte quarters p, quarter p of
// all 64 lanes contiguous (the four 16-byte reads per weight load are conflict-free)
__device__ __forceinline__ int en_xs_index(int i) {
  const int c = i >> 10, within = i & 1023;
  const int l = within >> 4, p = (within >> 2) & 3, e = within & 3;
  return (((c * 4 + p) << 6) + l) * 4 + e;
}

__device__ __forceinline__ bool en_timed_out(unsigned long long t0) {
  return __builtin_amdgcn_s_memrealtime() - t0 > EN_WAIT_TICKS;
}

// NCW consumer waves (a CU owns blocks of 2 NCW rows); the loader keeps DEPTH DMA instructions in flight
// NL loader waves deal the column steps round-robin (step js belongs to loader js % NL)
template <int NCW, int DEPTH, int NL>
__global__ __launch_bounds__(64 * (NCW + NL), 1) void gemv_engine_kernel(const EngineArgs a) {
  constexpr int EN_NCW = NCW;
  constexpr int EN_STEP_PIECES = 2 * NCW;           // weight pieces (1 KiB) per column step of a block
  constexpr int EN_STEP_INSTR = EN_STEP_PIECES + 1;  // + the step's scale fetch
  constexpr int EN_STEP_BYTES = EN_STEP_PIECES * 1024;
  constexpr int EN_SSLOT = EN_STEP_PIECES * 64;      // scales of one step: pieces x 16 groups x 4 B
  constexpr int EN_DEPTH = DEPTH;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  // the control block is the head of the dynamic LDS, whose LDS address is 0 (no static LDS here)
  const unsigned ctrl = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cu = blockIdx.x, ncu = gridDim.x;
  char* const sring = smem + a.sring_off;
  char* const ring = smem + a.ring_off;
  const int ring_steps = a.ring_steps;

  if (threadIdx.x < EN_CTRL / 4) reinterpret_cast<volatile unsigned*>(smem)[threadIdx.x] = 0;
  __syncthreads();  // the only workgroup barrier: the control words start at zero

  // this CU's blocks of every op, and where the op's column steps start in the kernel-wide count
  int b0[EN_MAX_OPS], nb[EN_MAX_OPS], js0[EN_MAX_OPS];
  const int bstride = (a.variant & 1) ? ncu : 1;
  {
    int js = 0;
#pragma unroll
    for (int i = 0; i < EN_MAX_OPS; ++i) {
      b0[i] = nb[i] = 0;
      js0[i] = js;
      if (i < a.nops) {
        if (a.variant & 1) {  // blocks dealt round-robin: block bl of this CU is cu + bl * ncu
          b0[i] = cu;
          nb[i] = (a.op[i].blocks - cu + ncu - 1) / ncu;
        } else {
          b0[i] = (int)((long long)cu * a.op[i].blocks / ncu);
          nb[i] = (int)((long long)(cu + 1) * a.op[i].blocks / ncu) - b0[i];
        }
        js += nb[i] * a.op[i].ppr;
      }
    }
  }

  if (wave >= EN_NCW) {
    const int ldr = wave - EN_NCW;
    // ------------------------------ loader ------------------------------
    unsigned issued = 0, cons_min = 0, published = 0;  // published: the landed count, which never goes back
    unsigned spin_space = 0, blocked_steps = 0;
    if (a.trace == 3) {  // probe: the consumers' own pace (they multiply whatever the ring holds)
      ec_st(ctrl + EC_FILLED + 4 * ldr, 0x7fffffffu);
      return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < a.nops; ++i) {
      const EngineOp& o = a.op[i];
      const char* wbase = reinterpret_cast<const char*>(o.w);
      const char* sbase = reinterpret_cast<const char*>(o.scales);
      const int groups_b = (o.M >> 6) * 4;  // bytes of scales per row
      for (int bl = 0; bl < nb[i]; ++bl) {
        const int v0 = (b0[i] + bl * bstride) * EN_STEP_PIECES;
        // this lane's scale source: piece (lane >> 2) of a step, 16-byte quarter (lane & 3)
        const size_t srow = (size_t)en_row(o, v0 + ((lane >> 2) & (EN_STEP_PIECES - 1))) * groups_b + (lane & 3) * 16;
        size_t wrow[EN_STEP_PIECES];
#pragma unroll
        for (int p = 0; p < EN_STEP_PIECES; ++p) wrow[p] = (size_t)en_row(o, v0 + p) * (size_t)o.M;
        for (int j = 0; j < o.ppr; ++j) {
          const unsigned js = (unsigned)(js0[i] + bl * o.ppr + j);
          if ((js & (NL - 1)) != (unsigned)ldr) continue;
          // ring space: the step that held this position has been finished by every consumer
          if (js >= cons_min + (unsigned)ring_steps) ++blocked_steps;
          while (js >= cons_min + (unsigned)ring_steps) {
            ++spin_space;
            // blocked on the consumers: nothing else to do, so everything issued so far is waited
            // for and published (a consumer may be waiting for one of the newest instructions)
            if (published != issued) {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              ec_st(ctrl + EC_FILLED + 4 * ldr, published = issued);
            }
            unsigned m = ec_ld(ctrl + EC_CONSUMED);
#pragma unroll
            for (int w = 1; w < EN_NCW; ++w) m = min(m, ec_ld(ctrl + EC_CONSUMED + 4 * w));
            cons_min = m;
            if (js < cons_min + (unsigned)ring_steps) break;
            if (en_timed_out(t0)) { ec_st(ctrl + EC_GAVE_UP, 1); cons_min = js; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          const int pos = (int)(js % (unsigned)ring_steps);
          if (lane < 4 * EN_STEP_PIECES && !(a.variant & 2))
            __builtin_amdgcn_global_load_lds(
                KX_CKN(a, i ? EB_S1 : EB_S0, reinterpret_cast<const uint32_t*>(sbase + srow + (size_t)j * 64), 16),
                (en_lds_ptr_t)(sring + pos * EN_SSLOT), 16, 0, 0);
#pragma unroll
          for (int p = 0; p < EN_STEP_PIECES; ++p)
            __builtin_amdgcn_global_load_lds(
                KX_CKN(a, i ? EB_W1 : EB_W0,
                       reinterpret_cast<const uint32_t*>(wbase + wrow[p] + (size_t)j * 1024 + lane * 16), 16),
                (en_lds_ptr_t)(ring + pos * EN_STEP_BYTES + p * 1024), 16, 0, 2);  // 2 = nt: streamed once
          issued += EN_STEP_INSTR;
          // everything but the newest DEPTH instructions has landed (loads complete in order)
          unsigned landed;
          if (a.variant & 4) {  // probe: issue pace alone (nothing waited for; pair with trace 2)
            landed = issued;
          } else if (ec_ld(ctrl + EC_SWEEPING)) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EN_DEPTH_THIN) : "memory");
            landed = issued - min(issued, (unsigned)EN_DEPTH_THIN);
          } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EN_DEPTH) : "memory");
            landed = issued - min(issued, (unsigned)EN_DEPTH);
          }
          if (landed > published) ec_st(ctrl + EC_FILLED + 4 * ldr, published = landed);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ec_st(ctrl + EC_FILLED + 4 * ldr, issued);
    if (a.trace == 1 && lane == 0) {
      atomicAdd(a.status + 1, (int)spin_space);
      atomicAdd(a.status + 2, (int)blocked_steps);
      atomicAdd(a.status + 3, (int)(__builtin_amdgcn_s_memrealtime() - t0));
    }
    return;
  }

  // ------------------------------ consumers ------------------------------
  const int pos_now = a.pos_dev ? *KX_CK(a, EB_MISC, a.pos_dev) : a.pos_val;
  const unsigned tag = ((unsigned)(pos_now + 1) << 7) | (unsigned)a.chain_id;
  unsigned filled_seen[1] = {};
  unsigned spin_fill = 0, starved_steps = 0, t_stage = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < a.nops; ++i) {
    const EngineOp& o = a.op[i];
    f32x4* xs = reinterpret_cast<f32x4*>(smem + o.xs_off);
    float* xsf = reinterpret_cast<float*>(xs);
    float ss = 0.f;
    const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
    if (i == 0) {
      // the first op's input is a vector in memory (written by an earlier launch): every consumer
      // stages a quarter of it
      const int m4 = o.M >> 2;
      for (int v4 = wave * 64 + lane; v4 < m4; v4 += 64 * EN_NCW) {
        f32x4 v = *KX_CK(a, EB_X0, reinterpret_cast<const f32x4*>(o.x) + v4);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        if (o.norm_w) {
          const f32x4 g = *KX_CK(a, EB_N0, reinterpret_cast<const f32x4*>(o.norm_w) + v4);
          v = f32x4{g.x * v.x, g.y * v.y, g.z * v.z, g.w * v.w};
        }
        *reinterpret_cast<f32x4*>(xsf + en_xs_index(v4 * 4)) = v;
      }
    } else {
      // the seam: sweep this wave's quarter of the granules until every tag matches
      const int quarter = o.M / EN_NCW;  // M % 1024 == 0
      const int g0 = wave * quarter;
      if (lane == 0) ec_add(ctrl + EC_SWEEPING, 1u);
      for (int base = g0; base < g0 + quarter; base += 64 * 16) {
        unsigned long long gv[16];
        bool all_ok = false;
        while (!all_ok) {
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int idx = min(base + k * 64 + lane, g0 + quarter - 1);
            gv[k] = __hip_atomic_load(KX_CK(a, EB_GRAN, a.granules + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          bool ok = true;
#pragma unroll
          for (int k = 0; k < 16; ++k) ok &= (unsigned)(gv[k] >> 32) == tag;
          all_ok = __all(ok);
          if (!all_ok) {
            if (en_timed_out(t0)) { ec_st(ctrl + EC_GAVE_UP, 1); break; }
            __builtin_amdgcn_s_sleep(4);
          }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int idx = base + k * 64 + lane;
          if (idx < g0 + quarter) {
            float v = __uint_as_float((unsigned)gv[k]);
            ss += v * v;
            if (o.norm_w) v *= *KX_CK(a, i ? EB_N1 : EB_N0, o.norm_w + idx);
            xsf[en_xs_index(idx)] = v;
          }
        }
      }
      if (lane == 0) ec_add(ctrl + EC_SWEEPING, ~0u);
    }
    // every consumer's share is staged before any consumer multiplies
    ss = wave_sum(ss);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) {
      // LDS operations of one wave complete in order: the partial sum and the staged share are
      // in LDS before the arrival count moves
      ec_st(ctrl + EC_SS_PART + 4 * (i * EN_MAX_NCW + wave), __float_as_uint(ss));
      ec_add(ctrl + EC_X_ARRIVE + 4 * i, 1u);
    }
    while (ec_ld(ctrl + EC_X_ARRIVE + 4 * i) < (unsigned)EN_NCW) {
      if (en_timed_out(t0)) { ec_st(ctrl + EC_GAVE_UP, 1); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    float rs = 1.0f;
    if (o.norm_w) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < EN_NCW; ++w) tot += __uint_as_float(ec_ld(ctrl + EC_SS_PART + 4 * (i * EN_MAX_NCW + w)));
      rs = 1.0f / sqrtf(tot / (float)o.M + o.norm_eps);
    }

    t_stage += (unsigned)(__builtin_amdgcn_s_memrealtime() - ts0);
    const bool publish = i + 1 < a.nops;
    for (int bl = 0; bl < nb[i]; ++bl) {
      const int v = (b0[i] + bl * bstride) * EN_STEP_PIECES + wave * 2;  // this wave's pair: virtual rows v, v + 1
      const int row0 = en_row(o, v), row1 = en_row(o, v + 1);
      float ta = 0.f, tb = 0.f;
      for (int j = 0; j < o.ppr; ++j) {
        const unsigned js = (unsigned)(js0[i] + bl * o.ppr + j);
        // this pair's second piece has landed: counted in its loader's own instructions
        const unsigned need = (js / NL) * EN_STEP_INSTR + 1 + wave * 2 + 2;
        const int fl = js & (NL - 1);
        unsigned fs = NL == 1 ? filled_seen[0] : 0u;
        if (fs < need && ec_ld(ctrl + EC_FILLED + 4 * fl) < need) ++starved_steps;
        while (fs < need) {
          fs = ec_ld(ctrl + EC_FILLED + 4 * fl);
          if (fs >= need) break;
          ++spin_fill;
          if (en_timed_out(t0)) { ec_st(ctrl + EC_GAVE_UP, 1); fs = need; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (NL == 1) filled_seen[0] = fs;
        if (a.trace == 2) {  // probe: the loader's own pace (nothing read, nothing computed)
          if (lane == 0) ec_st(ctrl + EC_CONSUMED + 4 * wave, js + 1);
          continue;
        }
        const int pos = (int)(js % (unsigned)ring_steps);
        const char* wp = ring + pos * EN_STEP_BYTES + wave * 2048 + lane * 16;
        const i32x4 wa = *reinterpret_cast<const i32x4*>(wp);
        const i32x4 wb = *reinterpret_cast<const i32x4*>(wp + 1024);
        const char* sp = sring + pos * EN_SSLOT + wave * 128 + (lane >> 2) * 4;
        const float sa = *reinterpret_cast<const float*>(sp);
        const float sb = *reinterpret_cast<const float*>(sp + 64);
        const int xb = (j * 4) << 6;
        const f32x4 x0 = xs[xb + lane], x1 = xs[xb + 64 + lane], x2 = xs[xb + 128 + lane], x3 = xs[xb + 192 + lane];
        ta = fmaf(en_dot16(wa, x0, x1, x2, x3), sa, ta);
        tb = fmaf(en_dot16(wb, x0, x1, x2, x3), sb, tb);
        // this wave's reads of the step are done (the values are in registers)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) ec_st(ctrl + EC_CONSUMED + 4 * wave, js + 1);
      }
      // ---- the pair's epilogue (lane 0) ----
      ta = wave_sum(ta) * rs;
      tb = wave_sum(tb) * rs;
      if (lane != 0) continue;
      if (o.mode == KX_GEMV_GATE_UP) {
        // swiglu_kernel_cu (cuda/swiglu_kernel.cu:4-22): gate * sigmoid(gate) * up
        const float h = ta * (1.0f / (1.0f + expf(-ta))) * tb;
        const int u = v >> 1;
        *KX_CK(a, i ? EB_Y10 : EB_Y00, o.y0 + u) = h;
        if (publish) __hip_atomic_store(KX_CK(a, EB_GRAN, a.granules + u), ((unsigned long long)tag << 32) | __float_as_uint(h),
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      float o0 = ta, o1 = tb;
      if (o.bias) {
        o0 += *KX_CK(a, EB_MISC, o.bias + row0);
        o1 += *KX_CK(a, EB_MISC, o.bias + row1);
      }
      if (o.mode == KX_GEMV_QKV_ROPE) {
        if (v < o.k1) {
          // rope_kernel_cu (cuda/rope_kernel.cu:17-34 half split, :111-127 interleaved)
          const int in_head = row0 & (o.head_size - 1);
          const int tab = o.rope_variant == KX_ROPE_HALF_SPLIT ? in_head * 2 : in_head;
          const float fci = *KX_CK(a, EB_MISC, o.sin_cache + ((size_t)pos_now * o.head_size + tab));
          const float fcr = *KX_CK(a, EB_MISC, o.cos_cache + ((size_t)pos_now * o.head_size + tab));
          const float t0r = o0 * fcr - o1 * fci;
          o1 = o0 * fci + o1 * fcr;
          o0 = t0r;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int row = r ? row1 : row0;
          const float val = r ? o1 : o0;
          if (row < o.k0) {
            *KX_CK(a, i ? EB_Y10 : EB_Y00, o.y0 + row) = val;
          } else {
            const bool is_k = row < o.k1;
            const int c = row - (is_k ? o.k0 : o.k1);
            if (o.kv_f16) {
              _Float16* dst = reinterpret_cast<_Float16*>(is_k ? o.y1 : o.y2) + (size_t)pos_now * o.kv_pos_stride;
              *KX_CK(a, is_k ? (i ? EB_Y11 : EB_Y01) : (i ? EB_Y12 : EB_Y02), dst + c) = (_Float16)val;
            } else {
              float* dst = (is_k ? o.y1 : o.y2) + (size_t)pos_now * o.kv_pos_stride;
              *KX_CK(a, is_k ? (i ? EB_Y11 : EB_Y01) : (i ? EB_Y12 : EB_Y02), dst + c) = val;
            }
          }
        }
        continue;
      }
      if (o.residual) {
        o0 += *KX_CK(a, EB_RES, o.residual + row0);
        o1 += *KX_CK(a, EB_RES, o.residual + row1);
      }
      *KX_CK(a, i ? EB_Y10 : EB_Y00, o.y0 + row0) = o0;
      *KX_CK(a, i ? EB_Y10 : EB_Y00, o.y0 + row1) = o1;
      if (publish) {
        __hip_atomic_store(KX_CK(a, EB_GRAN, a.granules + row0), ((unsigned long long)tag << 32) | __float_as_uint(o0),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(KX_CK(a, EB_GRAN, a.granules + row1), ((unsigned long long)tag << 32) | __float_as_uint(o1),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (ec_ld(ctrl + EC_GAVE_UP) && lane == 0) *KX_CK(a, EB_MISC, a.status) = 1;
  if (a.trace == 1 && lane == 0 && wave == 0) {
    atomicAdd(a.status + 4, (int)spin_fill);
    atomicAdd(a.status + 5, (int)starved_steps);
    atomicAdd(a.status + 6, (int)t_stage);
    atomicAdd(a.status + 7, (int)(__builtin_amdgcn_s_memrealtime() - t0));
    atomicAdd(a.status + 8, js0[a.nops - 1] + nb[a.nops - 1] * a.op[a.nops - 1].ppr);
  }
}

static int engine_cu_count() {
  static int cus[64] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return 0;
  if (!cus[d]) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) != hipSuccess) return 0;
    cus[d] = p.multiProcessorCount;
  }
  return cus[d];
}

}  // namespace kx

extern "C" size_t kx_gemv_w8_chain_workspace_bytes(int32_t max_rows) {
  if (max_rows <= 0) return 0;
  return 256 + (size_t)max_rows * 8;  // status words, then one granule per output row of a first op
}

extern "C" int kx_gemv_w8_chain_reset(void* workspace, size_t workspace_bytes, void* stream) {
  if (!workspace || workspace_bytes < 256) return KX_EINVAL;
  // status words to zero, granules to a tag no launch produces
  KX_HIP_TRY(hipMemsetAsync(workspace, 0, 256, (hipStream_t)stream));
  if (workspace_bytes > 256)
    KX_HIP_TRY(hipMemsetAsync(static_cast<char*>(workspace) + 256, 0xFF, workspace_bytes - 256, (hipStream_t)stream));
  return KX_OK;
}

extern "C" int kx_gemv_w8_chain_status(void* workspace, int32_t* gave_up) {
  if (!workspace || !gave_up) return KX_EINVAL;
  KX_HIP_TRY(hipMemcpy(gave_up, workspace, 4, hipMemcpyDeviceToHost));
  return KX_OK;
}

extern "C" int kx_gemv_w8_chain_f32(const kx_gemv_w8_desc* ops, int32_t nops, void* workspace,
                                    size_t workspace_bytes, int32_t chain_id, const int32_t* pos_dev,
                                    int32_t pos, void* stream) {
  using namespace kx;
  if (!ops || nops < 1 || nops > EN_MAX_OPS || !workspace || chain_id < 0 || chain_id > 127) return KX_EINVAL;
  if (!pos_dev && pos < 0) return KX_EINVAL;
  if ((uintptr_t)workspace % 256) return KX_EINVAL;
  // consumer waves and DMA depth: 4 x 40 unless KX_CHAIN_FORM="waves,depth" names another
  // instantiated form (the probe in tools/bench_engine.py walks them)
  int ncw = 4, depth = 40, nl = 1;
  if (const char* e = getenv("KX_CHAIN_FORM")) {
    if (sscanf(e, "%d,%d,%d", &ncw, &depth, &nl) < 2) return KX_EINVAL;
  }
  // KX_CHAIN_TRACE=1: every launch ADDS to the workspace's status words 1..8, over all CUs: the
  // loader's polls for ring space, its steps that had to wait for space, its run time (10 ns ticks);
  // consumer 0's polls for landed data, its steps that had to wait, its time staging inputs (seams
  // included), its run time; column steps.  kx_gemv_w8_chain_reset clears them.
  const char* tr = getenv("KX_CHAIN_TRACE");
  const int a_trace = tr ? atoi(tr) : 0;
  const int step_pieces = 2 * ncw, step_bytes = step_pieces * 1024, sslot = step_pieces * 64;
  EngineArgs a{};
  a.nops = nops;
  a.trace = a_trace;
  if (const char* e = getenv("KX_CHAIN_VARIANT")) a.variant = atoi(e);
  a.status = static_cast<int32_t*>(workspace);
  a.granules = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + 256);
  a.pos_dev = pos_dev;
  a.pos_val = pos;
  a.chain_id = chain_id;
  int lds = EN_CTRL;
  for (int i = 0; i < nops; ++i) {
    const kx_gemv_w8_desc& d = ops[i];
    EngineOp& o = a.op[i];
    if (!d.w || !d.scales || !d.y0 || d.M <= 0 || d.K <= 0) return KX_EINVAL;
    if (d.group_size != 64 || d.M % 1024 || d.K % step_pieces || ((uintptr_t)d.w % 16) || ((uintptr_t)d.scales % 16))
      return KX_EUNSUPPORTED;
    if (i == 0 && (!d.x || (uintptr_t)d.x % 16 || (d.norm_w && (uintptr_t)d.norm_w % 16))) return KX_EINVAL;
    if (i > 0) {
      // the input of a later op is the output vector of the op before it
      const kx_gemv_w8_desc& p = ops[i - 1];
      const int prev_out = p.mode == KX_GEMV_GATE_UP ? p.K / 2 : p.K;
      if (p.mode == KX_GEMV_QKV_ROPE || p.y1 || p.y2 || prev_out != d.M) return KX_EINVAL;
      if (d.x && d.x != p.y0) return KX_EINVAL;
      if (workspace_bytes < kx_gemv_w8_chain_workspace_bytes(prev_out)) return KX_EWORKSPACE;
    }
    o.w = d.w; o.scales = d.scales; o.x = d.x; o.norm_w = d.norm_w; o.norm_eps = d.norm_eps;
    o.y0 = d.y0; o.y1 = d.y1 ? d.y1 : d.y0; o.y2 = d.y2 ? d.y2 : d.y0;
    o.bias = d.bias; o.residual = d.residual;
    o.M = d.M; o.K = d.K; o.mode = d.mode;
    o.ppr = d.M / 1024; o.blocks = d.K / step_pieces;
    o.k0 = d.k0; o.k1 = d.k1;
    if (!d.y1 && !d.y2) { o.k0 = d.K; o.k1 = d.K; }
    switch (d.mode) {
      case KX_GEMV_PLAIN:
        if (d.y1 || d.y2) return KX_EUNSUPPORTED;  // routed rows: the QKV form
        break;
      case KX_GEMV_GATE_UP:
        if (d.K % 16 || d.residual || d.bias) return KX_EINVAL;
        o.half = d.K / 2;
        o.k0 = o.k1 = d.K;
        break;
      case KX_GEMV_QKV_ROPE: {
        if (!d.sin_cache || !d.cos_cache || d.head_size <= 0 || d.residual) return KX_EINVAL;
        if (d.head_size & (d.head_size - 1)) return KX_EUNSUPPORTED;
        if (o.k0 < 0 || o.k1 < o.k0 || o.k1 > d.K || o.k0 % d.head_size || (o.k1 - o.k0) % d.head_size) return KX_EINVAL;
        o.sin_cache = d.sin_cache; o.cos_cache = d.cos_cache;
        o.rope_variant = d.rope_variant; o.head_size = d.head_size;
        o.hs_shift = 0;
        while ((1 << o.hs_shift) < d.head_size) ++o.hs_shift;
        o.kv_pos_stride = d.kv_pos_stride; o.kv_f16 = d.kv_f16;
        if (d.pos_dev && d.pos_dev != pos_dev) return KX_EINVAL;  // one position per launch
        break;
      }
      default:
        return KX_EINVAL;
    }
    o.xs_off = lds;
    lds += d.M * 4;
  }
  const int room = 160 * 1024 - lds;
  int steps = room / (step_bytes + sslot);
  if (steps > 64) steps = 64;
  // the loader's instructions in flight and one step per consumer must fit the ring with room to spare
  if (steps * (step_pieces + 1) < nl * depth + 3 * (step_pieces + 1)) return KX_EUNSUPPORTED;
  a.ring_steps = steps;
  a.sring_off = lds;
  lds += steps * sslot;
  lds = (lds + 1023) / 1024 * 1024;
  a.ring_off = lds;
  lds += steps * step_bytes;
  if (lds > 160 * 1024) {
    a.ring_steps = --steps;
    lds -= step_bytes;
  }
  const int ncu = engine_cu_count();
  if (ncu <= 0) return KX_EINVAL;
  KX_BSET(a, EB_W0, a.op[0].w); KX_BSET(a, EB_S0, a.op[0].scales); KX_BSET(a, EB_X0, a.op[0].x);
  KX_BSET(a, EB_N0, a.op[0].norm_w); KX_BSET(a, EB_Y00, a.op[0].y0); KX_BSET(a, EB_Y01, a.op[0].y1);
  KX_BSET(a, EB_Y02, a.op[0].y2); KX_BSET(a, EB_W1, a.op[1].w); KX_BSET(a, EB_S1, a.op[1].scales);
  KX_BSET(a, EB_N1, a.op[1].norm_w); KX_BSET(a, EB_Y10, a.op[1].y0); KX_BSET(a, EB_Y11, a.op[1].y1);
  KX_BSET(a, EB_Y12, a.op[1].y2); KX_BSET(a, EB_GRAN, a.granules); KX_BSET(a, EB_RES, nullptr);
  KX_BSET(a, EB_MISC, nullptr);
#define KX_FORM(W, D, NLD)                                                                                \
  if (ncw == W && depth == D && nl == NLD) {                                                              \
    static OncePerDevice raised;                                                                          \
    if (int rc = raised.raise(reinterpret_cast<const void*>(&gemv_engine_kernel<W, D, NLD>), 160 * 1024)) return rc; \
    hipLaunchKernelGGL((gemv_engine_kernel<W, D, NLD>), dim3(ncu), dim3(64 * (W + NLD)), lds, (hipStream_t)stream, a); \
    KX_LAUNCH_CHECK();                                                                                    \
    return KX_OK;                                                                                         \
  }
  KX_FORM(4, 40, 1) KX_FORM(4, 56, 1) KX_FORM(8, 40, 1) KX_FORM(4, 24, 1) KX_FORM(4, 40, 2) KX_FORM(4, 24, 2)
  KX_FORM(8, 24, 2) KX_FORM(4, 24, 4) KX_FORM(4, 12, 4)
#undef KX_FORM
  return KX_EUNSUPPORTED;
}
