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
// Softmax is online (running max / sum); the workgroup that finishes last for a kv head (an
// arrival ticket in the workspace) merges the split results, so one launch does it all.
#include <math.h>

#include <cstdlib>
#include <mutex>

#define KX_FILE_ID 5
#include "kx_common.hpp"

namespace kx {

// operands of the bounds-checking build
enum AttnOperand { AB_Q, AB_K, AB_V, AB_OUT, AB_WS, AB_TICKETS, AB_POS, AB_NEXT0, AB_NEXT1 };

constexpr int kAttnWaves = 4;      // waves per workgroup (the eight-wave form below: NW = 8)
constexpr int kMaxSplits = 128;
// Head of the workspace: one arrival ticket per kv head.  Zero before the first launch; every
// launch leaves the tickets at zero again (atomicInc wraps at the launch's split count).
// Tickets sit 256 bytes apart: device-scope atomics on one cache line serialise.
constexpr size_t kTicketStride = 256;
constexpr int kMaxTickets = 64;
constexpr size_t kTicketBytes = kTicketStride * kMaxTickets;

// workspace record per (head, split): O[hs], then m, l and two pad words (16-byte aligned rows)
__host__ __device__ inline size_t attn_ws_stride(int hs) { return (size_t)hs + 4; }

// Split geometry shared by host and device.  The ntiles 16-key tiles of a kv head are cut into
// nsplit contiguous runs whose lengths differ by at most one (the first `rem` runs hold one tile
// more); never more splits than ceil(ntiles / nw) for workgroups of nw waves, so no wave of a split is idle unless
// the whole context is shorter than that.
struct AttnSplit {
  int nsplit, base, rem;
  __host__ __device__ int begin(int split) const { return split * base + (split < rem ? split : rem); }
  __host__ __device__ int count(int split) const { return base + (split < rem ? 1 : 0); }
  __host__ __device__ int longest() const { return base + (rem > 0 ? 1 : 0); }
};
// kvh: kv heads (workgroups per split).  While one tile per wave fits the grid (at most two
// workgroups per CU) that is the cut; past it the splits grow longer AND fall back to one
// workgroup per CU: with waves walking several tiles anyway, 256 workgroups of 8+ tiles beat 512 of
// 4..5 (equal at 4.4k keys, 1.5 % of a token at 8k).
// (fp32 cache only: with 16-bit rows a tile is half the bytes and the finer cut stays ahead.)
__host__ __device__ inline AttnSplit attn_split(int ntiles, int grid_splits, int kvh, bool fp32_cache,
                                                int nw = kAttnWaves) {
  int ns = (ntiles + nw - 1) / nw;
  if (ns > grid_splits && !fp32_cache) ns = grid_splits;
  if (ns > grid_splits) {
    const int one_per_cu = 256 / kvh > 0 ? 256 / kvh : 1;
    ns = grid_splits < one_per_cu ? grid_splits : one_per_cu;
  }
  if (ns < 1) ns = 1;
  return AttnSplit{ns, ntiles / ns, ntiles % ns};
}

// e^x for x <= 0 (or -inf) on the hardware exponential: one v_exp_f32 instead of the library
// routine's range reduction; relative error ~2^-22, far inside the attention tolerance.
__device__ __forceinline__ float fast_exp(float x) {
  return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

// Device-scope relaxed accesses: write-through stores / cache-bypassing loads.
__device__ __forceinline__ void st_device(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_device(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16- and 8-byte device-scope loads (buffer_load ... sc1; cache-policy bit 4 of the builtin)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 ld_device4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
  return f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
__device__ __forceinline__ float2 ld_device2(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16);
  return float2{__uint_as_float(v.x), __uint_as_float(v.y)};
}

// What the launch behind this one reads first (e.g. Wo's int8 rows and their scales), as up to two
// byte ranges cut into granules of whole 128-byte lines: granule c is what workgroup c of that
// launch reads, and workgroup c runs on XCD c mod 8.  The split merge at the end of the attention
// launch leaves HBM idle for a couple of microseconds with all but one workgroup per kv head
// finished: those workgroups, on their way out, read one dword of every line of the granules that
// belong to THEIR XCD, so the next launch finds them in that XCD's L2.  Nothing is kept; a launch
// that touches nothing (no ranges, or a single split) behaves as before.
struct AttnNext {
  const char* p[2];
  unsigned lines[2];       // 128-byte lines in the range
  unsigned gran_lines[2];  // lines per granule
  unsigned take_lines[2];  // lines of a granule that are read (its first ones)
};
constexpr int kTouchRounds = 4;  // loads per thread and range at most
// Bytes of the first range read per launch: what HBM delivers while the merge runs.  More makes the
// launch end later than the next one gains (Llama-3-8B, 4.4 k keys, Wo of 16.8 MB: all of it 480.7
// tok/s against 480.5 without, 35-50 % of it 485.7-488.1; 10 % 483.8).
constexpr size_t kTouchBudgetBytes = 7u << 20;
// ... and never more than this share of a range: small matrices (Llama-3.2-1B: Wo of 4.2 MB, a
// shorter merge) lose 0.35 % with all of it and nothing with 40 %.
constexpr double kTouchShareMost = 0.4;

__device__ __forceinline__ void attn_touch_next(const AttnNext& nx, int member, int members KX_BPARAM(9)) {
  // member: this workgroup's number among the launch's live workgroups, dispatched round-robin
  // over the XCDs
  const int xcd = member & 7, j = member >> 3, per_xcd = (members + 7) >> 3;
  // All loads go out, lanes without a line masked off, and are waited for once -- inside ONE
  // assembly statement, so that from hipcc's side nothing is in flight when it ends: hipcc does not
  // track loads issued by inline assembly and would hand their destination registers to other
  // values while a load may still write them.  (Written as C++ loads, each lane-tested load is
  // branched around and waited for on its own; made unconditional on a clamped address, thousands
  // of idle lanes hammer one line.)
  const char* at[2 * kTouchRounds];
  unsigned on[2 * kTouchRounds];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const unsigned gl = nx.gran_lines[r], tl = nx.p[r] ? nx.take_lines[r] : 1u;
    const unsigned granules = nx.p[r] ? (nx.lines[r] + gl - 1) / gl : 0u;
#pragma unroll
    for (int round = 0; round < kTouchRounds; ++round) {
      const unsigned idx = threadIdx.x + round * blockDim.x;
      const unsigned c = xcd + 8u * (j + per_xcd * (idx / tl));
      const unsigned line = c * gl + idx % tl;
      const bool ok = c < granules && line < nx.lines[r];
      const int i = r * kTouchRounds + round;
      at[i] = nx.p[r] + (size_t)(ok ? line : 0u) * 128;
      on[i] = ok;
      if (ok) KX_TOUCHP(r == 0 ? AB_NEXT0 : AB_NEXT1, at[i], 4);
    }
  }
  static_assert(kTouchRounds == 4, "the statement below names eight loads");
  unsigned o0, o1, o2, o3, o4, o5, o6, o7;
  unsigned long long live;
#define KX_TOUCH_ONE(n)                        \
  "s_mov_b64 exec, %[live]\n\t"               \
  "v_cmp_ne_u32_e32 vcc, 0, %[p" #n "]\n\t"    \
  "s_and_b64 exec, %[live], vcc\n\t"          \
  "global_load_dword %[o" #n "], %[a" #n "], off\n\t"
  asm volatile("s_mov_b64 %[live], exec\n\t"
               KX_TOUCH_ONE(0) KX_TOUCH_ONE(1) KX_TOUCH_ONE(2) KX_TOUCH_ONE(3)
               KX_TOUCH_ONE(4) KX_TOUCH_ONE(5) KX_TOUCH_ONE(6) KX_TOUCH_ONE(7)
               "s_mov_b64 exec, %[live]\n\t"
               "s_waitcnt vmcnt(0)"
               : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [o4] "=&v"(o4),
                 [o5] "=&v"(o5), [o6] "=&v"(o6), [o7] "=&v"(o7), [live] "=&s"(live)
               : [a0] "v"(at[0]), [a1] "v"(at[1]), [a2] "v"(at[2]), [a3] "v"(at[3]), [a4] "v"(at[4]),
                 [a5] "v"(at[5]), [a6] "v"(at[6]), [a7] "v"(at[7]), [p0] "v"(on[0]), [p1] "v"(on[1]),
                 [p2] "v"(on[2]), [p3] "v"(on[3]), [p4] "v"(on[4]), [p5] "v"(on[5]), [p6] "v"(on[6]),
                 [p7] "v"(on[7])
               : "vcc", "memory");
#undef KX_TOUCH_ONE
}

template <int HS>
struct AttnTile {
  static constexpr int NJ = HS / 16;
  static constexpr int NH = HS >= 64 ? HS / 64 : 1;
  f32x4 k[NJ];
  f32x4 v[4][NH];
};

// four consecutive cache elements as fp32; a 16-bit cache is widened here (exactly)
__device__ __forceinline__ f32x4 load_kv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load_kv4(const _Float16* p) {
  const f16x4 h = *reinterpret_cast<const f16x4*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

template <int HS, typename KVT>
__device__ __forceinline__ void attn_load_tile(AttnTile<HS>& T, const KVT* __restrict__ kbase,
                                               const KVT* __restrict__ vbase, int key0, int last,
                                               int kv_dim, int n, int g KX_BPARAM(9)) {
  constexpr int NJ = AttnTile<HS>::NJ, NH = AttnTile<HS>::NH;
  constexpr int DLV = HS >= 64 ? 16 : HS / 4;  // dim-lanes that carry data
  // K: lane (key n, g) loads K[key][16j+4g .. +4)
  const KVT* kr = kbase + (size_t)min(key0 + n, last) * kv_dim + 4 * g;
#pragma unroll
  for (int j = 0; j < NJ; ++j) T.k[j] = load_kv4(KX_CKPN(AB_K, kr + 16 * j, 4 * sizeof(KVT)));
  // V: lane (dim-lane n, g) loads V[key0+4g+r][64h+4n .. +4)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const KVT* vr = vbase + (size_t)min(key0 + 4 * g + r, last) * kv_dim + 4 * n;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      T.v[r][h] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (n < DLV) T.v[r][h] = load_kv4(KX_CKPN(AB_V, vr + 64 * h, 4 * sizeof(KVT)));
    }
  }
}

// ---- 4x4x1 form (head sizes 128 and 64, at most 4 query heads per kv head and group) -------------
// v_mfma_f32_16x16x4_f32 spends 16 head columns on kv_mul heads; with kv_mul = 4 three quarters of
// its cycles are idle, and a 16-key tile costs its wave 64 such instructions on the critical path.
// v_mfma_f32_4x4x1_16B_f32 is the same arithmetic (an fp32 fma chain per output) as 16 independent
// 4 x 4 blocks: block b of lanes 4b..4b+3 multiplies a 4-vector (one value per lane: rows) by a
// 4-vector (one value per lane: columns).  Scores: lane l = 16 qd + key holds K[key][16j + 4qd + e]
// (rows = the 4 keys of its block) and Q[head l&3][same dims] (columns = heads); the four dim
// quarters qd are summed across lanes afterwards.  Output: the probabilities of (key, head) go
// through 64 floats of LDS per wave into "rows = heads", the V values are "columns = dims":
// block b accumulates dims 4(l&31) + t for keys of parity l>>5, t = 0..3 in four accumulators.
// With 64-wide heads a V row is 16 lanes wide: the lanes split into FOUR key phases l>>4 (key =
// 4 step + phase) instead of two parities, and K / Q take 4 instead of 8 sixteen-dim steps.
template <int HS>
struct AttnTile4 {
  static constexpr int NJ = HS / 16;       // 16-dim steps of K and Q per lane
  static constexpr int LPR = HS / 4;       // lanes across a V row
  static constexpr int PH = 64 / LPR;      // key phases of the second product (2 or 4)
  static constexpr int NS = 16 / PH;       // its steps per tile (8 or 4)
  f32x4 k[NJ];
  f32x4 v[NS];
};

template <int HS, typename KVT>
__device__ __forceinline__ void attn_load_tile4(AttnTile4<HS>& T, const KVT* __restrict__ kbase,
                                                const KVT* __restrict__ vbase, int key0, int last,
                                                int kv_dim, int lane KX_BPARAM(9)) {
  using TT = AttnTile4<HS>;
  const KVT* kr = kbase + (size_t)min(key0 + (lane & 15), last) * kv_dim + 4 * (lane >> 4);
#pragma unroll
  for (int j = 0; j < TT::NJ; ++j) T.k[j] = load_kv4(KX_CKPN(AB_K, kr + 16 * j, 4 * sizeof(KVT)));
  const int ph = lane / TT::LPR, c4 = (lane % TT::LPR) * 4;
#pragma unroll
  for (int st = 0; st < TT::NS; ++st)
    T.v[st] = load_kv4(KX_CKPN(AB_V, vbase + (size_t)min(key0 + TT::PH * st + ph, last) * kv_dim + c4, 4 * sizeof(KVT)));
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// PREFETCH: a wave with several tiles requests the next one before it consumes the current one
// (costs a second set of tile registers; pointless when every wave has a single tile).
// M4: the 4x4x1 form above (HS = 128) with M4 groups of four query heads: 1 for kv_mul <= 4, 2 for
// kv_mul <= 8 (0: the 16-column form).
// NW: waves per workgroup.  8 (one workgroup per CU; 4x4x1 forms only, whose per-wave results are
// small enough for LDS) halves the number of splits a long context is cut into at the same number
// of waves in flight: half the partials to publish, acknowledge and merge behind the stream.
// STATIC: the cut into splits and tiles is that of last_val (an upper bound of the position, passed
// by value) and the position read from the device only masks: rows above it get zero weight.  The
// position is then first needed at the softmax of a wave's first tile instead of ahead of the first
// cache row's address -- one dependent memory round trip less between the launch and the stream.
// The rows up to last_val must be readable and hold finite numbers.
template <int HS, bool PREFETCH, typename KVT, int M4 = 0, int NW = kAttnWaves, bool STATIC = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void attn_decode_kernel(
    const float* __restrict__ q, const KVT* __restrict__ kc, const KVT* __restrict__ vc,
    float* __restrict__ out, float* __restrict__ ws, unsigned* __restrict__ tickets, int last_val,
    const int32_t* __restrict__ pos_dev, int grid_kvh, int grid_splits, int kv_mul, int kv_dim, float scale,
    unsigned long long* __restrict__ trace, const AttnNext next KX_BPARAM(9)) {
  // optional timeline (tools/bench_attn.py with KX_ATTN_TRACE): wall-clock stamps (100 MHz), 8 per
  // workgroup
  // grid_kvh x grid_splits: the grid, as arguments (they come with the other arguments; hipcc fetched
  // gridDim with a vector load of its own in some instantiations)
  auto stamp = [&](int slot) {
    if (trace && threadIdx.x == 0)
      trace[((size_t)blockIdx.y * grid_kvh + blockIdx.x) * 8 + slot] = wall_clock64();
  };
  stamp(0);
  constexpr int NJ = AttnTile<HS>::NJ;  // float4 K loads per lane per tile
  constexpr int NH = AttnTile<HS>::NH;  // 64-dim halves of the output
  constexpr int kAttnWaves = NW;  // (shadows the default: everything below is per NW)
  constexpr int MAXH = M4 ? 4 * M4 : 16;  // query heads of a kv head this form can hold
  static_assert(NW == 4 || (NW == 8 && M4 > 0), "eight waves: 4x4x1 forms only");
  __shared__ float sm_m[kAttnWaves][16];
  __shared__ float sm_l[kAttnWaves][16];
  __shared__ __attribute__((aligned(16))) float sm_o[kAttnWaves][MAXH][HS];
  __shared__ float sm_sf[kAttnWaves * kMaxSplits];  // final merge: split weights, one set per wave
  __shared__ int sm_last;
  __shared__ __attribute__((aligned(16))) float sm_p[M4 ? kAttnWaves * 64 * M4 : 4];  // M4: P of one tile per wave and head group

  // (readfirstlane: hipcc must know the wave index is uniform, or every test on it is a divergent
  // branch whose two sides count each other's loads as pending)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int kvh = blockIdx.x, split = blockIdx.y;
  // The position may live in device memory so a captured launch can be replayed as it advances;
  // the grid is then sized for the longest context and surplus splits leave at once.
  // `last`: rows the cut and the loads go by; `seen`: the last row that counts.
  int last, seen;
  if constexpr (STATIC) {
    // a VECTOR load: a uniform address would be fetched through the scalar cache, and scalar loads
    // are waited for all together (lgkmcnt(0)) with the kernel arguments, ahead of everything else;
    // the lane-dependent zero hides the uniformity
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    seen = *KX_CKP(AB_POS, pos_dev + lane_zero);
    last = last_val;
  } else {
    last = pos_dev ? *KX_CKP(AB_POS, pos_dev) : last_val;
    seen = last;
  }
  const int ntiles = (last >> 4) + 1;
  const AttnSplit cut = attn_split(ntiles, grid_splits, grid_kvh, sizeof(KVT) == 4, NW);
  const int nsplit = cut.nsplit;
  if (split >= nsplit) return;
  const KVT* kbase = kc + (size_t)kvh * HS;
  const KVT* vbase = vc + (size_t)kvh * HS;

  const int t_begin = cut.begin(split);
  const int t_end = t_begin + cut.count(split);
  if constexpr (M4 > 0) {
    static_assert(M4 == 0 || HS == 128 || HS == 64, "the 4x4x1 form is laid out for 128- and 64-wide heads");
    using T4 = AttnTile4<HS>;
    constexpr int NJ4 = T4::NJ, LPR4 = T4::LPR, PH4 = T4::PH, NS4 = T4::NS;
    // G groups of four query heads (kv_mul <= 4 G): every group repeats the two products on the
    // tile the wave holds in registers, with its own Q fragment, softmax state and accumulators
    constexpr int G = M4 > 0 ? M4 : 1;
    const int qd = lane >> 4, kb = (lane >> 2) & 3, hd = lane & 3, hi = lane / LPR4;  // hi: key phase
    // Q: lane (qd, *, head hd) holds Q[head][16j + 4qd + e]; requested ahead of the tile
    f32x4 qf4[G][NJ4];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const float* qrow = q + ((size_t)kvh * kv_mul + min(4 * gi + hd, kv_mul - 1)) * HS + 4 * qd;
#pragma unroll
      for (int j = 0; j < NJ4; ++j) qf4[gi][j] = *KX_CKP(AB_Q, reinterpret_cast<const f32x4*>(qrow + 16 * j));
    }

    int t = t_begin + wave;
    AttnTile4<HS> cur;
    if (t < t_end) attn_load_tile4<HS, KVT>(cur, kbase, vbase, t << 4, last, kv_dim, lane KX_BFWD);
    // single-tile form: without the fences the scheduler sinks the tile loads into the product
    // loop and waits for them one at a time (the prefetching form schedules better without)
    if (!PREFETCH) __builtin_amdgcn_sched_barrier(0);
    f32x4 o[G][4];  // o[gi][t][r] = O[head 4gi + r][dim 4(lane&31) + t], keys of parity hi
    float m_run[G], l_run[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[gi][e] = f32x4{0.f, 0.f, 0.f, 0.f};
      m_run[gi] = -INFINITY;
      l_run[gi] = 0.f;
    }
    float* pl = sm_p + wave * 64 * G;

    for (; t < t_end; t += kAttnWaves) {
      const int key0 = t << 4;
      AttnTile4<HS> nxt;
      const bool more = t + kAttnWaves < t_end;
      if (PREFETCH && more)
        attn_load_tile4<HS, KVT>(nxt, kbase, vbase, (t + kAttnWaves) << 4, last, kv_dim, lane KX_BFWD);
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        // ---- scores: rows = keys 4kb..4kb+3, columns = heads, this lane's quarter of the dims ----
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ4; ++j) {
          s0 = mfma4(cur.k[j].x, qf4[gi][j].x, s0);
          s1 = mfma4(cur.k[j].y, qf4[gi][j].y, s1);
          s0 = mfma4(cur.k[j].z, qf4[gi][j].z, s0);
          s1 = mfma4(cur.k[j].w, qf4[gi][j].w, s1);
        }
        float sv[4] = {s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // sum of the four dim quarters
          sv[r] = xor16_sum(sv[r]);
          sv[r] = xor32_sum(sv[r]);
          sv[r] *= scale;
          if (key0 + 4 * kb + r > seen) sv[r] = -INFINITY;
        }
        // every lane (any qd, kb) of head hd ends up with the head's maximum over the 16 keys
        float mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        mx = fmaxf(mx, dpp_mov<0x124>(mx));  // row_ror:4 -> the next key block, same head
        mx = fmaxf(mx, dpp_mov<0x128>(mx));  // row_ror:8
        const float m_new = fmaxf(m_run[gi], mx);
        // STATIC: a tile wholly above the position leaves the maximum at -inf; subtracting 0
        // instead keeps every factor at exp(-inf) = 0
        const float m_sub = STATIC && m_new == -INFINITY ? 0.f : m_new;
        const float alpha = fast_exp(m_run[gi] - m_sub);  // first tile: exp(-inf) = 0
        m_run[gi] = m_new;
        float pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = fast_exp(sv[r] - m_sub);
        l_run[gi] = l_run[gi] * alpha + ((pv[0] + pv[1]) + (pv[2] + pv[3]));  // this lane's 4 keys
        // P[key][head] -> LDS as [phase][head][step], key = PH4 step + phase, so that a lane of
        // the second product reads the probabilities it needs as 16-byte words (two parities of 8
        // steps for 128-wide heads, four phases of 4 steps for 64-wide ones)
        float* plg = pl + 64 * gi;
        if (qd == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 4 * kb + r;
            plg[((key % PH4) * 4 + hd) * NS4 + key / PH4] = pv[r];
          }
        }
        // the accumulators keep the head on the register axis: alpha of head r for every lane
        const float ah[4] = {dpp_mov<0x00>(alpha), dpp_mov<0x55>(alpha), dpp_mov<0xAA>(alpha),
                             dpp_mov<0xFF>(alpha)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[gi][e].x *= ah[0]; o[gi][e].y *= ah[1]; o[gi][e].z *= ah[2]; o[gi][e].w *= ah[3];
        }
        float pa[NS4];
#pragma unroll
        for (int w4 = 0; w4 < NS4 / 4; ++w4) {
          const f32x4 pw = *reinterpret_cast<const f32x4*>(plg + (hi * 4 + hd) * NS4 + 4 * w4);
          pa[4 * w4] = pw.x; pa[4 * w4 + 1] = pw.y; pa[4 * w4 + 2] = pw.z; pa[4 * w4 + 3] = pw.w;
        }
        // ---- output: rows = heads, columns = dims 4(lane % LPR4) + t, one key per step ----
#pragma unroll
        for (int st = 0; st < NS4; ++st) {
          o[gi][0] = mfma4(pa[st], cur.v[st].x, o[gi][0]);
          o[gi][1] = mfma4(pa[st], cur.v[st].y, o[gi][1]);
          o[gi][2] = mfma4(pa[st], cur.v[st].z, o[gi][2]);
          o[gi][3] = mfma4(pa[st], cur.v[st].w, o[gi][3]);
        }
      }
      if (more) {
        if (PREFETCH) cur = nxt;
        else attn_load_tile4<HS, KVT>(cur, kbase, vbase, (t + kAttnWaves) << 4, last, kv_dim, lane KX_BFWD);
      }
      if (!PREFETCH) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- wave result -> LDS ----
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      float lr = l_run[gi];
      lr += dpp_mov<0x124>(lr);  // the four key blocks of a head
      lr += dpp_mov<0x128>(lr);
      if (lane < 4 && 4 * gi + lane < kv_mul) {
        sm_m[wave][4 * gi + lane] = m_run[gi];
        sm_l[wave][4 * gi + lane] = lr;
      }
      float oo[4][4];  // [head r][t], all key phases
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        oo[0][e] = xor32_sum(o[gi][e].x);
        oo[1][e] = xor32_sum(o[gi][e].y);
        oo[2][e] = xor32_sum(o[gi][e].z);
        oo[3][e] = xor32_sum(o[gi][e].w);
        if constexpr (PH4 == 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) oo[r][e] = xor16_sum(oo[r][e]);
        }
      }
      if (lane < LPR4) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * gi + r < kv_mul)
            *reinterpret_cast<f32x4*>(&sm_o[wave][4 * gi + r][4 * lane]) =
                f32x4{oo[r][0], oo[r][1], oo[r][2], oo[r][3]};
      }
    }
  } else {
  // Q fragment: B operand, lane (head n, group g) holds Q[head][16j+4g+e].  Lanes past the last
  // head of the group repeat it (their output columns are never stored): the loads stay
  // unconditional (a load under a lane test is branched around and waited for on its own) and go
  // out ahead of the tile loads: they hit in L2 and loads return in order.
  float qf[NJ][4];
  const float* qrow = q + ((size_t)kvh * kv_mul + min(n, kv_mul - 1)) * HS + 4 * g;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const f32x4 v = *KX_CKP(AB_Q, reinterpret_cast<const f32x4*>(qrow + 16 * j));
    qf[j][0] = v.x; qf[j][1] = v.y; qf[j][2] = v.z; qf[j][3] = v.w;
  }

  int t = t_begin + wave;
  AttnTile<HS> cur;
  if (t < t_end) attn_load_tile<HS, KVT>(cur, kbase, vbase, t << 4, last, kv_dim, n, g KX_BFWD);

  f32x4 o[NH][4];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) o[h][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (; t < t_end; t += kAttnWaves) {
    const int key0 = t << 4;
    // the wave's next tile is requested before this one is consumed
    AttnTile<HS> nxt;
    const bool more = t + kAttnWaves < t_end;
    if (PREFETCH && more)
      attn_load_tile<HS, KVT>(nxt, kbase, vbase, (t + kAttnWaves) << 4, last, kv_dim, n, g KX_BFWD);
    // ---- S^T[key][head] = K . Q^T (two independent accumulation chains) ----
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.k[j].x, qf[j][0], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.k[j + 1].x, qf[j + 1][0], s1, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.k[j].y, qf[j][1], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.k[j + 1].y, qf[j + 1][1], s1, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.k[j].z, qf[j][2], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.k[j + 1].z, qf[j + 1][2], s1, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.k[j].w, qf[j][3], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.k[j + 1].w, qf[j + 1][3], s1, 0, 0, 0);
    }
    // lane (head n, g) now holds scores of keys key0+4g+r, r=0..3
    float sv[4] = {(s0.x + s1.x) * scale, (s0.y + s1.y) * scale, (s0.z + s1.z) * scale,
                   (s0.w + s1.w) * scale};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (key0 + 4 * g + r > seen) sv[r] = -INFINITY;
    float mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
    mx = xor16_max(mx);
    mx = xor32_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float m_sub = STATIC && m_new == -INFINITY ? 0.f : m_new;  // see the 4x4x1 form
    const float alpha = fast_exp(m_run - m_sub);  // first tile: exp(-inf) = 0
    m_run = m_new;
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = fast_exp(sv[r] - m_sub);
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
        o[h][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.v[r][h].x, p[r], o[h][0], 0, 0, 0);
        o[h][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.v[r][h].y, p[r], o[h][1], 0, 0, 0);
        o[h][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.v[r][h].z, p[r], o[h][2], 0, 0, 0);
        o[h][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.v[r][h].w, p[r], o[h][3], 0, 0, 0);
      }
    if (more) {
      if (PREFETCH) cur = nxt;
      else attn_load_tile<HS, KVT>(cur, kbase, vbase, (t + kAttnWaves) << 4, last, kv_dim, n, g KX_BFWD);
    }
  }

  // ---- wave result -> LDS.  lane (head n, g'): o[h][e][r'] = O[head][64h + 4(4g'+r') + e] ----
  l_run = xor16_sum(l_run);
  l_run = xor32_sum(l_run);
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
  }
  stamp(1);  // wave 0's tiles are done
  __syncthreads();
  stamp(2);  // every wave's tiles are done

  // ---- merge the workgroup's waves; write the final row or a split partial ----
  for (int idx = threadIdx.x; idx < kv_mul * HS; idx += blockDim.x) {
    const int head = idx / HS, d = idx % HS;
    float M = sm_m[0][head];
#pragma unroll
    for (int w = 1; w < kAttnWaves; ++w) M = fmaxf(M, sm_m[w][head]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < kAttnWaves; ++w) {
      // a wave without tiles carries m = -inf, l = 0, o = 0
      const float f = sm_m[w][head] == -INFINITY ? 0.f : fast_exp(sm_m[w][head] - M);
      L += sm_l[w][head] * f;
      O += sm_o[w][head][d] * f;
    }
    const int hq = kvh * kv_mul + head;
    if (nsplit == 1) {
      *KX_CKP(AB_OUT, out + ((size_t)hq * HS + d)) = O / L;
    } else {
      // write-through (device-scope) stores: visible to the merging workgroup on any XCD without
      // a cache write-back
      float* dst = ws + ((size_t)hq * grid_splits + split) * attn_ws_stride(HS);
      if (d == 0) {
        st_device(KX_CKP(AB_WS, dst + HS), M);
        st_device(KX_CKP(AB_WS, dst + HS + 1), L);
      }
      st_device(KX_CKP(AB_WS, dst + d), O);
    }
  }
  if (nsplit == 1) return;

  // ---- the workgroup that arrives last for this kv head combines the splits ----
  // The partial is published by three things together.  (1) Every element of it is stored with a
  // device-scope atomic store, which the hardware writes through the XCD's L2, so no dirty line of
  // this kernel ever needs a cache write-back.  (2) Each storing wave waits for the
  // acknowledgement of its stores (vmcnt(0)); the asm's memory clobber keeps the compiler from
  // moving a store below it.  (3) The barrier (a workgroup-scope release/acquire) orders all of
  // that before thread 0's ticket.  A device-scope release fence in front of the ticket would
  // say the same in the memory model's own terms, but it is emitted as an L2 write-back
  // (buffer_wbl2) that every workgroup of the launch then pays for with nothing to write:
  // measured +0.3 us per launch at 256 keys, +0.9 us at 1024, +3.0 us at 4096 keys with 64
  // splits.  KX_ATTN_AGENT_RELEASE compiles it in for a build that wants it regardless.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(3);  // partial stored and acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
#ifdef KX_ATTN_AGENT_RELEASE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#endif
    const unsigned old = atomicInc(KX_CKP(AB_TICKETS, &tickets[kvh * (kTicketStride / sizeof(unsigned))]), (unsigned)nsplit - 1u);  // wraps back to 0
    sm_last = old == (unsigned)nsplit - 1u;
  }
  __syncthreads();
  stamp(4);  // ticket drawn
  if (!sm_last) {
    attn_touch_next(next, split * grid_kvh + kvh, nsplit * grid_kvh KX_BFWD);
    return;
  }
  // Every byte another workgroup wrote is read below with a device-scope (sc1) load, which goes
  // past this CU's vector cache to where the write-through stores landed; nothing else of theirs
  // is read.  That takes the place of a device-scope acquire fence, whose cache invalidate costs
  // the merging workgroup about a microsecond before its first load; the wavefront-scope fence
  // emits no instruction and only keeps the compiler from hoisting the loads above the ticket.
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const __amdgpu_buffer_rsrc_t ws_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(ws, 0, 0x7fffffff, 0x00020000);
  // out[h][d] = sum_s O_s[d] e^{m_s-M} / sum_s l_s e^{m_s-M}.  One wave per head, no workgroup
  // barriers: a row of HS floats is HS/4 lanes wide, the wave's RG row groups take the splits
  // round-robin, and every load of a pass is in flight before the first is used.
  constexpr int LPR = HS / 4;        // lanes per row
  constexpr int RG = 64 / LPR;       // rows loaded per wave instruction
  constexpr int PASS = 16;           // loads in flight per lane
  const size_t stride = attn_ws_stride(HS);
  const int c4 = (lane % LPR) * 4, rg = lane / LPR;
  float* sf = sm_sf + wave * kMaxSplits;  // this wave's split weights
  for (int head = wave; head < kv_mul; head += kAttnWaves) {
    // byte offset of this head's partials in the workspace (a few MiB at most)
    const unsigned base = (unsigned)((kvh * kv_mul + head) * grid_splits * stride) * 4u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < nsplit; s0 += PASS * RG) {
      f32x4 ov[PASS];
#pragma unroll
      for (int k = 0; k < PASS; ++k) {
        // rows past the last split re-read the last one; their weight below is zero
        const unsigned sp = min(s0 + k * RG + rg, nsplit - 1);
        KX_TOUCHP(AB_WS, reinterpret_cast<const char*>(ws) + (base + (sp * (unsigned)stride + c4) * 4u), 16);
        ov[k] = ld_device4(ws_rsrc, base + (sp * (unsigned)stride + c4) * 4u);
      }
      if (s0 == 0) {
        // split weights, one split per lane (two rounds cover kMaxSplits = 128)
        float m[kMaxSplits / 64], l[kMaxSplits / 64];
        float M = -INFINITY;
#pragma unroll
        for (int c = 0; c < kMaxSplits / 64; ++c) {
          // unconditional (clamped) loads: a load under a lane test is branched around and
          // waited for on its own, one memory round trip per round
          const int sp = c * 64 + lane;
          KX_TOUCHP(AB_WS, reinterpret_cast<const char*>(ws) + (base + ((unsigned)min(sp, nsplit - 1) * (unsigned)stride + HS) * 4u), 8);
          const float2 ml =
              ld_device2(ws_rsrc, base + ((unsigned)min(sp, nsplit - 1) * (unsigned)stride + HS) * 4u);
          m[c] = sp < nsplit ? ml.x : -INFINITY;
          l[c] = sp < nsplit ? ml.y : 0.f;
          M = fmaxf(M, m[c]);
        }
        M = wave_max(M);
        float f[kMaxSplits / 64], Lsum = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxSplits / 64; ++c) {
          f[c] = c * 64 + lane < nsplit ? fast_exp(m[c] - M) : 0.f;
          Lsum += f[c] * l[c];
        }
        const float inv_l = 1.0f / wave_sum(Lsum);
#pragma unroll
        for (int c = 0; c < kMaxSplits / 64; ++c) sf[c * 64 + lane] = f[c] * inv_l;
      }
#pragma unroll
      for (int k = 0; k < PASS; ++k) {
        const int sp = s0 + k * RG + rg;
        const float w = sp < nsplit ? sf[sp] : 0.f;
        acc.x = fmaf(ov[k].x, w, acc.x);
        acc.y = fmaf(ov[k].y, w, acc.y);
        acc.z = fmaf(ov[k].z, w, acc.z);
        acc.w = fmaf(ov[k].w, w, acc.w);
      }
    }
    // fold the row groups
#pragma unroll
    for (int off = 32; off >= LPR; off >>= 1) {
      acc.x += __shfl_xor(acc.x, off);
      acc.y += __shfl_xor(acc.y, off);
      acc.z += __shfl_xor(acc.z, off);
      acc.w += __shfl_xor(acc.w, off);
    }
    if (rg == 0) *KX_CKP(AB_OUT, reinterpret_cast<f32x4*>(out + (size_t)(kvh * kv_mul + head) * HS + c4)) = acc;
  }
  stamp(5);  // this wave's heads merged
}

// Any head size / any kv_mul: one workgroup per head, scores kept in a [head][seq_len] scratch
// (the reference's score tensor has exactly that shape, llama3.cpp:523-525).
__global__ void attn_decode_generic_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                           const float* __restrict__ vc, float* __restrict__ out,
                                           float* __restrict__ score, int last, int kv_mul,
                                           int kv_dim, int hs, int seq_len, float scale KX_BPARAM(9)) {
  __shared__ float red[16];
  const int h = blockIdx.x;
  const float* qh = q + (size_t)h * hs;
  const size_t hoff = (size_t)(h / kv_mul) * hs;
  float* sc = score + (size_t)h * seq_len;
  float mx = -INFINITY;
  for (int t = threadIdx.x; t <= last; t += blockDim.x) {
    const float* kr = kc + (size_t)t * kv_dim + hoff;
    float acc = 0.f;
    for (int d = 0; d < hs; ++d) acc = fmaf(*KX_CKP(AB_Q, qh + d), *KX_CKP(AB_K, kr + d), acc);
    acc *= scale;
    *KX_CKP(AB_WS, sc + t) = acc;
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
    const float e = expf(*KX_CKP(AB_WS, sc + t) - mx);
    *KX_CKP(AB_WS, sc + t) = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();  // also publishes sc[] to the whole workgroup
  sum = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += red[w];
  for (int d = threadIdx.x; d < hs; d += blockDim.x) {
    float acc = 0.f;
    for (int t = 0; t <= last; ++t) acc = fmaf(*KX_CKP(AB_WS, sc + t), *KX_CKP(AB_V, vc + ((size_t)t * kv_dim + hoff + d)), acc);
    *KX_CKP(AB_OUT, out + ((size_t)h * hs + d)) = acc / sum;
  }
}

}  // namespace kx

extern "C" size_t kx_attn_workspace_bytes(int32_t head_num, int32_t head_size) {
  if (head_num <= 0 || head_size <= 0) return 0;
  return kx::kTicketBytes +
         (size_t)head_num * kx::kMaxSplits * kx::attn_ws_stride(head_size) * sizeof(float);
}

template <typename KVT>
static int mha_decode_launch(const float* query, const KVT* key_cache, const KVT* value_cache,
                             float* out, int32_t pos, const int32_t* pos_dev, int32_t head_num,
                             int32_t kv_mul, int32_t head_size, int32_t kv_dim, int32_t seq_len,
                             int32_t layer_index, void* workspace, size_t workspace_bytes,
                             void* stream, const kx::AttnNext* next_or_null = nullptr,
                             bool rows_to_pos_readable = false) {
  using namespace kx;
  if (!query || !key_cache || !value_cache || !out) return KX_EINVAL;
  const AttnNext next = next_or_null ? *next_or_null : AttnNext{};
  if (pos < 0 || pos >= seq_len || head_num <= 0 || kv_mul <= 0 || head_num % kv_mul) return KX_EINVAL;
  const int kvh = head_num / kv_mul;
  if (kv_dim != kvh * head_size) return KX_EINVAL;
  const size_t layer_off = (size_t)layer_index * (size_t)seq_len * (size_t)kv_dim;
  const KVT* kc = key_cache + layer_off;
  const KVT* vc = value_cache + layer_off;
  constexpr bool kFp32Cache = sizeof(KVT) == 4;
  const float scale = 1.0f / sqrtf((float)head_size);  // mha_kernel.cu / cpu/mha_kernel.cpp:11
  // one MFMA column block holds the heads of a kv head; the tiled kernel is built for the
  // head sizes of the supported model families
  const bool tiled = kv_mul <= 16 && (head_size == 32 || head_size == 64 || head_size == 128) &&
                     (((uintptr_t)query | (uintptr_t)kc | (uintptr_t)vc) % 16 == 0) &&
                     kv_dim % 4 == 0;
  if (!tiled) {
    if (pos_dev || !kFp32Cache) return KX_EUNSUPPORTED;
    // the score rows sit behind the ticket words so the two paths can share one workspace
    if (!workspace || workspace_bytes < kTicketBytes + (size_t)head_num * seq_len * sizeof(float))
      return KX_EWORKSPACE;
    hipLaunchKernelGGL(attn_decode_generic_kernel, dim3(head_num), dim3(256), 0,
                       (hipStream_t)stream, query, reinterpret_cast<const float*>(kc),
                       reinterpret_cast<const float*>(vc), out,
                       reinterpret_cast<float*>(static_cast<char*>(workspace) + kTicketBytes), pos,
                       kv_mul, kv_dim, head_size, seq_len, scale
                       KX_BPASS(9, query, kc, vc, out, workspace, nullptr, nullptr, nullptr, nullptr));
    KX_LAUNCH_CHECK();
    return KX_OK;
  }

  // `pos` sizes the grid: the actual position when passed by value, the largest position the
  // launch may ever see when it is read from pos_dev.
  const int ntiles = (pos >> 4) + 1;
  // One tile per wave is the finest useful cut.  Beyond two workgroups per CU the splits stop
  // growing in number and grow in length instead (runs of 4..8 tiles: some waves take two tiles,
  // both requested up front by the PREFETCH form): a CU that draws a third workgroup finishes
  // late, and so does a kv head whose split count runs into a third merge pass -- 69 splits of 4
  // tiles cost 15.3 us at 4368 keys where 64 splits of 4..5 cost 14.2.
  // up to four query heads per kv head: the 4x4x1 form of the matrix instruction (a quarter of
  // the matrix-core cycles of the 16-column form, which leaves 12 head columns idle)
  // Five to eight heads run as two groups of four on the same tile registers only where that was
  // measured faster than the 16-column form: eight heads per kv head (Llama-3-70B) from 2048 keys on
  // (one kv head, 4368 keys: 13.8 vs 14.3 us; 1024 keys: 10.1 vs 9.8 us; seven heads -- Qwen2.5-7B,
  // the second group one head short -- 16.6 vs 16.1 us at 4368 keys).  KX_ATTN_M4X2=1 / =0 force it.
  // (64-wide heads: the one-group form only)
  int m4 = head_size == 64 ? (kv_mul <= 4 ? 1 : 0)
           : head_size != 128 ? 0 : kv_mul <= 4 ? 1 : (kv_mul == 8 && pos >= 2047) ? 2 : 0;
  if (head_size == 64 && getenv("KX_ATTN_NO_M4_64")) m4 = 0;  // A/B runs and tests
  if (head_size == 128 && kv_mul > 4 && kv_mul <= 8)
    if (const char* e = getenv("KX_ATTN_M4X2")) m4 = e[0] == '1' ? 2 : 0;
  // Eight waves per workgroup, one workgroup per CU (4x4x1 forms): chosen where the four-wave form
  // has run out of its finest cut (one tile per wave in two workgroups per CU) and would fall back
  // to one four-wave workgroup per CU with 8+ tiles per split.  Twice the waves in flight per split
  // there: Llama-3-8B geometry at 4368 keys 12.7 vs 13.1 us, equal at 8192; below that point the
  // four-wave form wins (4096 keys: 12.4 vs 12.6 us; 2048: 8.7 vs 10.2).  KX_ATTN_NW=4 / =8 force
  // a form (8: 4x4x1 forms only).
  const int one_per_cu = 256 / kvh > 0 ? 256 / kvh : 1;
  int nw = m4 > 0 && (ntiles + kAttnWaves - 1) / kAttnWaves > 2 * one_per_cu ? 8 : kAttnWaves;
  if (const char* e = getenv("KX_ATTN_NW")) nw = (e[0] == '8' && m4 > 0) ? 8 : kAttnWaves;
  int nsplit = (ntiles + nw - 1) / nw;
  const int cap_grid = nw == 8 ? one_per_cu : (512 / kvh > 0 ? 512 / kvh : 1);
  if (nsplit > cap_grid) nsplit = cap_grid;
  if (nsplit > kMaxSplits) nsplit = kMaxSplits;
  if (kvh > kMaxTickets) return KX_EUNSUPPORTED;
  const size_t per_split = (size_t)head_num * attn_ws_stride(head_size) * sizeof(float);
  if (nsplit > 1) {
    const size_t fit =
        workspace && workspace_bytes > kTicketBytes ? (workspace_bytes - kTicketBytes) / per_split : 0;
    if (fit < (size_t)nsplit) nsplit = fit < 1 ? 1 : (int)fit;
  }
  // The grid keeps the two-per-CU split count even where the cut at `pos` falls back to one per CU:
  // a launch that reads its position from the device may see shorter contexts, for which the
  // finer cut is the better one (surplus workgroups leave at once).
  const AttnSplit cut = attn_split(ntiles, nsplit, kvh, kFp32Cache, nw);
  // the cut of `pos` itself, with the position read from the device as a mask only
  static const bool no_fixed = getenv("KX_ATTN_NO_FIXED_CUT") != nullptr;  // A/B runs
  const bool fixed_cut = pos_dev && rows_to_pos_readable && !no_fixed;
  if (!pos_dev || fixed_cut) nsplit = cut.nsplit;
  unsigned* tickets = reinterpret_cast<unsigned*>(workspace);
  float* ws = workspace ? reinterpret_cast<float*>(static_cast<char*>(workspace) + kTicketBytes) : nullptr;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(kvh, nsplit), block(nw * 64);
  // KX_ATTN_TRACE set and a workspace with room behind the partials: per-workgroup timeline there
  unsigned long long* trace = nullptr;
  {
    const size_t used = kTicketBytes + (size_t)kMaxSplits * per_split;
    static const bool env_trace = getenv("KX_ATTN_TRACE") != nullptr;  // read once
    if (env_trace && workspace && workspace_bytes >= used + (size_t)kvh * nsplit * 64)
      trace = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + used);
  }
  const bool prefetch = cut.longest() > nw;
#define KX_ATTN_LAUNCH_ONE(HS_, PF_, M4_, NW_, ST_)                                                          \
  hipLaunchKernelGGL((attn_decode_kernel<HS_, PF_, KVT, M4_, NW_, ST_>), grid, block, 0, s, query, kc, vc, out, \
                     ws, tickets, pos, pos_dev, kvh, nsplit, kv_mul, kv_dim, scale, trace, next                  \
                     KX_BPASS(9, query, kc, vc, out, ws, tickets, pos_dev, next.p[0], next.p[1]))
#define KX_ATTN_LAUNCH_NW(HS_, M4_, NW_)                          \
  do {                                                           \
    if (prefetch && fixed_cut) KX_ATTN_LAUNCH_ONE(HS_, true, M4_, NW_, true);        \
    else if (prefetch) KX_ATTN_LAUNCH_ONE(HS_, true, M4_, NW_, false);               \
    else if (fixed_cut) KX_ATTN_LAUNCH_ONE(HS_, false, M4_, NW_, true);              \
    else KX_ATTN_LAUNCH_ONE(HS_, false, M4_, NW_, false);                            \
  } while (0)
#define KX_ATTN_LAUNCH(HS_, M4_) KX_ATTN_LAUNCH_NW(HS_, M4_, 4)
  switch (head_size) {
    case 32: KX_ATTN_LAUNCH(32, 0); break;
    case 64:
      if (m4 == 1 && nw == 8) KX_ATTN_LAUNCH_NW(64, 1, 8);
      else if (m4 == 1) KX_ATTN_LAUNCH(64, 1);
      else KX_ATTN_LAUNCH(64, 0);
      break;
    case 128:
      if (m4 == 1 && nw == 8) KX_ATTN_LAUNCH_NW(128, 1, 8);
      else if (m4 == 2 && nw == 8) KX_ATTN_LAUNCH_NW(128, 2, 8);
      else if (m4 == 1) KX_ATTN_LAUNCH(128, 1);
      else if (m4 == 2) KX_ATTN_LAUNCH(128, 2);
      else KX_ATTN_LAUNCH(128, 0);
      break;
    default: return KX_EUNSUPPORTED;
  }
#undef KX_ATTN_LAUNCH
#undef KX_ATTN_LAUNCH_NW
#undef KX_ATTN_LAUNCH_ONE
  KX_LAUNCH_CHECK();
  return KX_OK;
}

extern "C" int kx_mha_decode_f32(const float* query, const float* key_cache,
                                 const float* value_cache, float* out, int32_t pos,
                                 int32_t head_num, int32_t kv_mul, int32_t head_size, int32_t kv_dim,
                                 int32_t seq_len, int32_t layer_index, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return mha_decode_launch<float>(query, key_cache, value_cache, out, pos, nullptr, head_num, kv_mul,
                           head_size, kv_dim, seq_len, layer_index, workspace, workspace_bytes,
                           stream);
}

extern "C" int kx_mha_decode_scores_f32(const float* query, const float* key_cache,
                                        const float* value_cache, float* out, int32_t pos,
                                        int32_t head_num, int32_t kv_mul, int32_t head_size,
                                        int32_t kv_dim, int32_t seq_len, int32_t layer_index,
                                        float* score, size_t score_bytes, void* stream) {
  using namespace kx;
  const bool tiled = kv_mul > 0 && kv_mul <= 16 &&
                     (head_size == 32 || head_size == 64 || head_size == 128);
  if (!tiled) {
    // the generic kernel keeps its scores in [head_num][seq_len] rows: the score tensor itself,
    // without the ticket words that head a workspace (they would not fit behind it)
    if (!query || !key_cache || !value_cache || !out) return KX_EINVAL;
    if (pos < 0 || pos >= seq_len || head_num <= 0 || kv_mul <= 0 || head_num % kv_mul) return KX_EINVAL;
    if (kv_dim != head_num / kv_mul * head_size) return KX_EINVAL;
    if (!score || score_bytes < (size_t)head_num * seq_len * sizeof(float)) return KX_EWORKSPACE;
    const size_t layer_off = (size_t)layer_index * (size_t)seq_len * (size_t)kv_dim;
    hipLaunchKernelGGL(attn_decode_generic_kernel, dim3(head_num), dim3(256), 0, (hipStream_t)stream,
                       query, key_cache + layer_off, value_cache + layer_off, out, score, pos, kv_mul,
                       kv_dim, head_size, seq_len, 1.0f / sqrtf((float)head_size)
                       KX_BPASS(9, query, key_cache, value_cache, out, score, nullptr, nullptr, nullptr, nullptr));
    KX_LAUNCH_CHECK();
    return KX_OK;
  }
  // Tiled kernel: the tensor serves as split workspace when it holds the ticket words and at
  // least two splits; its bytes are whatever the last call left, so the tickets are cleared on
  // the stream first.  A smaller tensor means one split per kv head and no workspace at all.
  const size_t per_split = (size_t)head_num * attn_ws_stride(head_size) * sizeof(float);
  if (score && ((uintptr_t)score % 16 == 0) && score_bytes >= kTicketBytes + 2 * per_split) {
    KX_HIP_TRY(hipMemsetAsync(score, 0, kTicketBytes, (hipStream_t)stream));
    return mha_decode_launch<float>(query, key_cache, value_cache, out, pos, nullptr, head_num,
                                    kv_mul, head_size, kv_dim, seq_len, layer_index, score,
                                    score_bytes, stream);
  }
  return mha_decode_launch<float>(query, key_cache, value_cache, out, pos, nullptr, head_num, kv_mul,
                                  head_size, kv_dim, seq_len, layer_index, nullptr, 0, stream);
}

// ---- the entry point that allocates nothing ---------------------------------------------------
// Split workspace of kx_mha_decode_static_f32: the ticket words followed by the split partials, as
// in a caller-provided workspace.  One copy per GPU the library runs on, held until the process
// ends.  Sized for 64 query heads of 128 (the largest model shape this library is benchmarked on,
// unsharded): 16 KiB + 128 splits x 64 x 132 floats = 4.14 MiB.
namespace kx {
constexpr size_t kStaticPartialFloats = (size_t)kMaxSplits * 64 * (128 + 4);
constexpr size_t kStaticScratchBytes = kTicketBytes + kStaticPartialFloats * sizeof(float);
constexpr int kMaxDevices = 64;
}  // namespace kx
static __device__ __align__(256) unsigned char scratch[kx::kStaticScratchBytes];
// Device address of `scratch` on each GPU (the array's name is only a host-side handle): looked up
// once per device, with that device current.
static void* g_scratch_ptr[kx::kMaxDevices];
static std::mutex g_scratch_mutex;

// Puts a split workspace back into its initial state (arrival tickets at zero), e.g. after a
// launch that used it was aborted midway.
extern "C" int kx_attn_workspace_reset(void* workspace, size_t workspace_bytes, void* stream) {
  if (!workspace || workspace_bytes < kx::kTicketBytes) return KX_EINVAL;
  KX_HIP_TRY(hipMemsetAsync(workspace, 0, kx::kTicketBytes, (hipStream_t)stream));
  return KX_OK;
}

extern "C" size_t kx_mha_decode_static_limit_bytes(void) { return kx::kStaticScratchBytes; }

extern "C" int kx_mha_decode_static_f32(const float* query, const float* key_cache,
                                        const float* value_cache, float* out, int32_t pos,
                                        int32_t head_num, int32_t kv_mul, int32_t head_size,
                                        int32_t kv_dim, int32_t seq_len, int32_t layer_index,
                                        void* stream) {
  using namespace kx;
  if (head_num <= 0 || head_size <= 0 || seq_len <= 0) return KX_EINVAL;
  const bool tiled = kv_mul > 0 && kv_mul <= 16 &&
                     (head_size == 32 || head_size == 64 || head_size == 128);
  // what the call would need of a caller-provided workspace must fit the static array
  const size_t need = tiled ? kx_attn_workspace_bytes(head_num, head_size)
                            : kTicketBytes + (size_t)head_num * (size_t)seq_len * sizeof(float);
  if (need > kStaticScratchBytes) return KX_ELIMIT;
  int dev = -1;
  KX_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return KX_ELIMIT;
  void* ptr = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
      void* p = nullptr;
      KX_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
      if (!p) return KX_EINVAL;
      g_scratch_ptr[dev] = p;
#ifdef KX_BOUNDS
      kx::bounds_track(p, kStaticScratchBytes);  // the array is a block like any other to the checks
#endif
    }
    ptr = g_scratch_ptr[dev];
  }
  // The array keeps whatever the last call on this GPU left.  A launch that ran to completion
  // returns its tickets to zero, one that was cut short does not: they are cleared every time.
  KX_HIP_TRY(hipMemsetAsync(ptr, 0, kTicketBytes, (hipStream_t)stream));
  return mha_decode_launch<float>(query, key_cache, value_cache, out, pos, nullptr, head_num, kv_mul,
                                  head_size, kv_dim, seq_len, layer_index, ptr, kStaticScratchBytes,
                                  stream);
}

extern "C" int kx_mha_decode_dpos_f32(const float* query, const float* key_cache,
                                      const float* value_cache, float* out,
                                      const int32_t* pos_dev, int32_t max_pos, int32_t head_num,
                                      int32_t kv_mul, int32_t head_size, int32_t kv_dim,
                                      int32_t seq_len, int32_t layer_index, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (!pos_dev) return KX_EINVAL;
  return mha_decode_launch<float>(query, key_cache, value_cache, out, max_pos, pos_dev, head_num, kv_mul,
                           head_size, kv_dim, seq_len, layer_index, workspace, workspace_bytes,
                           stream);
}

namespace kx {
// ranges that are not whole 128-byte lines from an aligned start are left out (nothing depends on them)
static AttnNext attn_next_of(const void* p0, size_t bytes0, size_t gran0, const void* p1, size_t bytes1,
                             size_t gran1) {
  AttnNext nx{};
  const void* ps[2] = {p0, p1};
  const size_t bs[2] = {bytes0, bytes1}, gs[2] = {gran0, gran1};
  static const bool off = getenv("KX_ATTN_NO_TOUCH") != nullptr;  // A/B runs
  for (int r = 0; r < 2; ++r) {
    if (off || !ps[r] || !bs[r] || !gs[r] || gs[r] % 128 || bs[r] % 128 || (uintptr_t)ps[r] % 128) continue;
    if (bs[r] / 128 > 0x7fffffffu / 2) continue;
    nx.p[r] = static_cast<const char*>(ps[r]);
    nx.lines[r] = (unsigned)(bs[r] / 128);
    nx.gran_lines[r] = (unsigned)(gs[r] / 128);
  }
  // the same share of every granule, of both ranges: the first range's budget over its size
  // (KX_ATTN_TOUCH_PCT: that share in percent, for A/B runs)
  static const int pct = getenv("KX_ATTN_TOUCH_PCT") ? atoi(getenv("KX_ATTN_TOUCH_PCT")) : 0;
  const size_t ref_bytes = nx.p[0] ? (size_t)nx.lines[0] * 128 : nx.p[1] ? (size_t)nx.lines[1] * 128 : 1;
  const double share = pct > 0 ? std::min(100, pct) / 100.0 : std::min(kTouchShareMost, (double)kTouchBudgetBytes / (double)ref_bytes);
  for (int r = 0; r < 2; ++r) {
    if (!nx.p[r]) continue;
    nx.take_lines[r] = std::max(1u, (unsigned)(nx.gran_lines[r] * share + 0.5));
  }
  return nx;
}
}  // namespace kx

extern "C" int kx_mha_decode_dpos_ex_f32(const float* query, const float* key_cache,
                                            const float* value_cache, float* out,
                                            const int32_t* pos_dev, int32_t max_pos, int32_t rows_to_max_pos_readable, int32_t head_num,
                                            int32_t kv_mul, int32_t head_size, int32_t kv_dim,
                                            int32_t seq_len, int32_t layer_index, void* workspace,
                                            size_t workspace_bytes, const void* next0,
                                            size_t next0_bytes, size_t next0_granule, const void* next1,
                                            size_t next1_bytes, size_t next1_granule, void* stream) {
  if (!pos_dev) return KX_EINVAL;
  const kx::AttnNext nx = kx::attn_next_of(next0, next0_bytes, next0_granule, next1, next1_bytes, next1_granule);
  return mha_decode_launch<float>(query, key_cache, value_cache, out, max_pos, pos_dev, head_num, kv_mul,
                                  head_size, kv_dim, seq_len, layer_index, workspace, workspace_bytes,
                                  stream, &nx, rows_to_max_pos_readable != 0);
}

extern "C" int kx_mha_decode_kv16(const float* query, const void* key_cache16,
                                  const void* value_cache16, float* out, int32_t pos,
                                  const int32_t* pos_dev, int32_t head_num, int32_t kv_mul,
                                  int32_t head_size, int32_t kv_dim, int32_t seq_len,
                                  int32_t layer_index, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  return mha_decode_launch<_Float16>(query, static_cast<const _Float16*>(key_cache16),
                                     static_cast<const _Float16*>(value_cache16), out, pos, pos_dev,
                                     head_num, kv_mul, head_size, kv_dim, seq_len, layer_index,
                                     workspace, workspace_bytes, stream);
}

extern "C" int kx_mha_decode_kv16_ex(const float* query, const void* key_cache16,
                                        const void* value_cache16, float* out, int32_t pos,
                                        const int32_t* pos_dev, int32_t rows_to_pos_readable, int32_t head_num, int32_t kv_mul,
                                        int32_t head_size, int32_t kv_dim, int32_t seq_len,
                                        int32_t layer_index, void* workspace, size_t workspace_bytes,
                                        const void* next0, size_t next0_bytes, size_t next0_granule,
                                        const void* next1, size_t next1_bytes, size_t next1_granule,
                                        void* stream) {
  const kx::AttnNext nx = kx::attn_next_of(next0, next0_bytes, next0_granule, next1, next1_bytes, next1_granule);
  return mha_decode_launch<_Float16>(query, static_cast<const _Float16*>(key_cache16),
                                     static_cast<const _Float16*>(value_cache16), out, pos, pos_dev,
                                     head_num, kv_mul, head_size, kv_dim, seq_len, layer_index,
                                     workspace, workspace_bytes, stream, &nx, rows_to_pos_readable != 0);
}
