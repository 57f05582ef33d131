// W8A16 group-quant product for a FEW activation rows (2..192) on gfx950:
//   Y[t][p] = sum_i X[t][i] * scales[(p*M+i)/64] * (float)w[p*M+i],   t < T <= 192.
//
// The reference has no batched matmul (its prompt phase feeds one token per launch,
// demo/main.cpp:18-41; cuda/matmul_kernel.cu:89-134 ignore a second input dim).  With so few rows
// the product is bound by the int8 weight stream, exactly like the decode GEMV, and the tiled
// GEMMs (128- or 256-row tiles) leave most of the chip idle: a 32-row prompt on 4096 x 4096 is 32
// of their tiles.  This kernel streams the weights once and multiplies them with all T rows on
// the matrix cores (v_mfma_f32_16x16x32_f16; int8 is exact in fp16).
//
// Work split
//   - a workgroup of five waves owns 128 weight rows and a slice of the reduction, walked in
//     units of 128 elements (two quant groups).  Waves 0-3 own 32 rows each (two blocks of 16);
//     wave 4 only brings in the activation rows.
//   - everything arrives by LDS-DMA (global_load_lds, 16 bytes per lane), so loads in flight hold
//     no registers and are waited for with hand-counted vmcnt.  A compute wave keeps a private
//     ring of three unit slots (4 KiB of weights each) and requests unit u + 2 before it
//     multiplies unit u; the rows' scales come 32 bytes per row at a time (four units' worth, one
//     instruction) into a small ring of their own.  A deeper ring did not stream faster (six slots:
//     W1|W3 of Llama-3-8B at 16 rows 31.7 us against 30.1); a second workgroup on the CU, for which
//     the short ring leaves the LDS, does (26.0).  The activation units (T rows x
//     128 halves) are shared, in two to four buffers; wave 4 requests them one to three units
//     ahead and waits for them on its own counter, so that wait never drains the compute waves'
//     weight prefetch (loads complete in order within a wave).
//   - one DMA instruction covers 8 rows x 128 contiguous bytes (16 rows x 64 bytes, the
//     operand layout itself, reads a quarter slower from HBM).  Inside LDS a row's eight 16-byte
//     pieces are stored XOR-ed with a per-row key chosen so that the fragment reads
//     (ds_read_b128, lane = row + 16 * piece pair) touch every bank once.
//   - int8 -> fp16 without conversions: v_perm_b32 drops each (byte ^ 0x80) into the mantissa of
//     0x6400 (= 1024 + byte + 128, exact), one v_pk_add_f16 takes the 1152 off again.
//   - numerics as the tiled int8 kernel: activations are fp16, every group's 64 products are
//     summed in fp32 by the matrix core, times the group's fp32 scale, added to an fp32 total.
//
// Narrow matrices cut the reduction S = 2, 4, 8 or 16 ways over workgroups: each part writes its
// fp32 partial tile to a slab with write-through stores, takes an arrival ticket (which wraps back
// to zero), and the last arriver of a tile adds the parts in index order with device-scope loads
// and writes the rows -- the same sum whichever part arrives last.
#define KX_FILE_ID 2
#include "kx_common.hpp"

namespace kx {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
using lds_ptr_t = __attribute__((address_space(3))) void*;

constexpr int FR_BN = 128;     // weight rows per workgroup
constexpr int FR_KU = 128;     // reduction elements per unit
constexpr int FR_SLOT = 4096;  // one unit of a wave's 32 rows
constexpr int FR_SRING = 3;    // scale slots per compute wave: 1 KiB = the scales of four units
// weight unit slots per compute wave
template <int TB>
constexpr int rows_ring() { return 3; }
template <int TB>
constexpr int rows_wave_lds() { return rows_ring<TB>() * FR_SLOT + FR_SRING * 1024; }
constexpr size_t kRowsTicketBytes = 4096;
constexpr size_t kRowsX16Bytes = 8u << 20;  // fp16 copy of fp32 activation rows (128 x 32768, 192 x 16384)

struct RowsArgs {
  const _Float16* X;    // [T][ldx] fp16
  const int8_t* w;      // [K][M]
  const float* scales;  // [K][M/64]
  const float* residual;
  float* y0; float* y1; float* y2;  // weight-row ranges [0,k0) [k0,k1) [k1,K)
  float* slabs;
  unsigned* tickets;
  int64_t ldx, ld0, ld1, ld2, ldr;
  int32_t k0, k1;
  int32_t T, M, K;
  int32_t nsplit;
  // what follows the product (the launches these replace are named at the entry points)
  int32_t mode;                 // ROWS_PLAIN, ROWS_SWIGLU, ROWS_QKV_ROPE
  _Float16* h16; int64_t ldh;   // ROWS_SWIGLU: fp16 rows out, silu(gate) * up
  int32_t half;                 //   gate rows [0, half), up rows [half, 2 half)
  const float* sin_cache; const float* cos_cache; const float* bias;  // ROWS_QKV_ROPE
  int32_t head_size, rope_variant, pos0;
  KX_BFIELD(13)  // operands of the bounds-checking build, in the order of RowsOperand
};
enum { ROWS_PLAIN = 0, ROWS_SWIGLU = 1, ROWS_QKV_ROPE = 2 };
enum RowsOperand { RB_X, RB_W, RB_S, RB_RES, RB_Y0, RB_Y1, RB_Y2, RB_SLABS, RB_TICKETS, RB_H16, RB_SIN, RB_COS, RB_BIAS };

// Which weight rows a compute wave's ring holds: row(rb, i) = base + i * istep + rb * rbstep for row
// block rb (0, 1) and index i (0..15).  A lane ends up with the totals of rows (0, i) and (1, i) for
// its tokens, so the fused forms put the two rows that meet in the epilogue there: a gate row and
// its up row, or the two rows of a rotation pair (neighbours, or half a head apart).
struct RowsMap { int base, istep, rbstep; };
__device__ __forceinline__ RowsMap rows_map(const RowsArgs& a, int tile, int w) {
  if (a.mode == ROWS_SWIGLU) return {tile * 64 + w * 16, 1, a.half};
  const int n0 = tile * 128;
  if (a.mode == ROWS_QKV_ROPE && n0 < a.k1) {
    if (a.rope_variant == KX_ROPE_HALF_SPLIT) {
      const int P = a.head_size >> 1, c = w * 16;
      return {n0 + (c / P) * a.head_size + c % P, 1, P};
    }
    return {n0 + w * 32, 2, 1};
  }
  return {n0 + w * 32, 1, 16};
}

// four int8 -> four fp16 (two packed pairs), exact
__device__ __forceinline__ void cvt_w8x4(uint32_t d, uint32_t& lo, uint32_t& hi) {
  const uint32_t x = d ^ 0x80808080u;
  const f16x2 off = {(_Float16)1152.0f, (_Float16)1152.0f};
  const f16x2 a = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0x64646464u, x, 0x04010400u)) - off;
  const f16x2 b = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0x64646464u, x, 0x04030402u)) - off;
  lo = __builtin_bit_cast(uint32_t, a);
  hi = __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ f16x8 cvt_w8x8(uint32_t d0, uint32_t d1) {
  uint32_t a, b, c, d;
  cvt_w8x4(d0, a, b);
  cvt_w8x4(d1, c, d);
  const u32x4 r = {a, b, c, d};
  return __builtin_bit_cast(f16x8, r);
}

// waits until at most `n` of this wave's DMA instructions are still in flight
__device__ __forceinline__ void wait_dma(int n) {
#define KX_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    KX_W(0) KX_W(1) KX_W(2) KX_W(3) KX_W(4) KX_W(5) KX_W(6) KX_W(7) KX_W(8) KX_W(9) KX_W(10) KX_W(11)
    KX_W(12) KX_W(13) KX_W(14) KX_W(15) KX_W(16) KX_W(17) KX_W(18) KX_W(19) KX_W(20) KX_W(21)
    default: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
  }
#undef KX_W
}
static_assert(4 * (3 - 1) + 2 <= 22, "wait_dma covers the units in flight and their scale fetches");

// TB: token blocks of the tile, TBW: this wave's share of them, starting at block tb0
template <int S, int TB, int TBW>
__device__ __forceinline__ void rows_combine(const RowsArgs& a, const char* slab_base, f32x4 (&acc)[2][TBW],
                                             __amdgpu_buffer_rsrc_t rs, unsigned lane_off, int tb0) {
  constexpr unsigned PB = TB * 8192u;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int tb = 0; tb < TBW; ++tb) {
      u32x4 v[S];
#pragma unroll
      for (int p = 0; p < S; ++p) {
        const unsigned off = (unsigned)p * PB + lane_off + (unsigned)((rb * TB + tb0 + tb) * 1024);
        KX_TOUCH(a, RB_SLABS, slab_base + off, 16);
        v[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
      }
      f32x4 t = {__uint_as_float(v[0].x), __uint_as_float(v[0].y), __uint_as_float(v[0].z),
                 __uint_as_float(v[0].w)};
#pragma unroll
      for (int p = 1; p < S; ++p) {
        t[0] += __uint_as_float(v[p].x); t[1] += __uint_as_float(v[p].y);
        t[2] += __uint_as_float(v[p].z); t[3] += __uint_as_float(v[p].w);
      }
      acc[rb][tb] = t;
    }
}

// activation unit buffers: as many as LDS has room for beside the weight rings
template <int TB>
constexpr int rows_xbufs() { return TB == 1 ? 4 : TB == 2 || TB > 8 ? 2 : 3; }
// the activation wave counts its requests with vmcnt (6 bits): the units in flight behind the one
// it waits for must stay below 64 instructions
template <int TB>
constexpr bool rows_counts_fit() { return (rows_xbufs<TB>() - 2) * TB * 4 < 64; }
// LDS: [activation unit buffers: TB x 4 KiB each][per compute wave: weight ring, scale ring]
template <int TB>
constexpr int rows_lds_bytes() { return rows_xbufs<TB>() * TB * 4096 + 4 * rows_wave_lds<TB>(); }

// PAIR = 1: four compute waves, each with all TB token blocks of its 32 weight rows.
// PAIR = 2: eight compute waves; waves w and w + 4 share wave w's weight ring (wave w requests it)
// and take half of the token blocks each, so every SIMD holds two compute waves and one's LDS
// reads, conversions and scale updates run beside the other's matrix instructions.
template <int TB, int PAIR>
__global__ __launch_bounds__(64 * (4 * PAIR + 1), 1) void gemm_w8_rows_kernel(const RowsArgs a) {
  static_assert(PAIR == 1 || (PAIR == 2 && TB % 2 == 0), "token blocks split evenly over a pair");
  constexpr int NCW = 4 * PAIR;    // compute waves; wave NCW brings in the activation rows
  constexpr int TBW = TB / PAIR;   // token blocks per compute wave
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int XBUF = TB * 4096;  // one activation unit: 2 groups x TB token blocks x 2 halves x 1 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsplit = a.nsplit;
  const int part = blockIdx.x % nsplit, tile = blockIdx.x / nsplit;
  const RowsMap map = rows_map(a, tile, wave & 3);
  // this part's units [u0, u0 + un)
  int u0, un;
  {
    const int nu = a.M / FR_KU, base = nu / nsplit, rem = nu % nsplit;
    u0 = part * base + min(part, rem);
    un = base + (part < rem ? 1 : 0);
  }
  constexpr int NXB = rows_xbufs<TB>();
  constexpr int FR_RING = rows_ring<TB>(), FR_AHEAD = FR_RING - 1, FR_WAVE_LDS = rows_wave_lds<TB>();
  static_assert(rows_counts_fit<TB>() && rows_lds_bytes<TB>() <= 160 * 1024, "token blocks per unit");
  char* const xbuf = smem;
  char* const ring = smem + NXB * XBUF + (wave & 3) * FR_WAVE_LDS;
  char* const sring = ring + FR_RING * FR_SLOT;

  const int tb0 = (wave >> 2) * TBW;  // first token block of a compute wave
  const bool requests = wave < 4;     // the wave that requests the ring's weights
  f32x4 acc[2][TBW];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int tb = 0; tb < TBW; ++tb) acc[rb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (wave == NCW) {
    // ---- activation rows: one instruction = 8 tokens x 128 bytes (one quant group) ----
    // lane (r8, s) = (lane >> 3, lane & 7) fetches piece s ^ key of its token, key from the
    // token's index inside its block of 16 with bits 0 and 1 of (index >> 1) exchanged
    unsigned xoff[TB][2];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r8 = lane >> 3, x = h * 4 + (r8 >> 1);
        const int key = ((x & 1) << 1) | ((x >> 1) & 1) | (x & 4);
        const unsigned row = (unsigned)min(tb * 16 + h * 8 + r8, a.T - 1);
        xoff[tb][h] = (row * (unsigned)a.ldx + (unsigned)u0 * FR_KU) * 2u + (unsigned)(((lane & 7) ^ key) * 16);
      }
    const char* const xbase = reinterpret_cast<const char*>(a.X);
    auto issue_x = [&](int u) {
      char* dst = xbuf + (u % NXB) * XBUF;
#pragma unroll
      for (int gl = 0; gl < 2; ++gl)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            __builtin_amdgcn_global_load_lds(
                KX_CKN(a, RB_X,
                       reinterpret_cast<const uint32_t*>(xbase + (xoff[tb][h] + (unsigned)(u * FR_KU + gl * 64) * 2u)),
                       16),
                (lds_ptr_t)(dst + ((gl * TB + tb) * 2 + h) * 1024), 16, 0, 0);
    };
    // unit u + NXB - 1 is requested after the barrier that ends unit u - 1 (whose buffer it takes)
    // and has NXB - 1 units' time to arrive: one unit ahead, every unit would take a whole trip
    // to L2, and that trip, not the weight stream, would set the pace
#pragma unroll
    for (int u = 0; u < NXB - 1; ++u)
      if (u < un) issue_x(u);
    for (int u = 0; u < un; ++u) {
      // unit u has landed when at most the younger ones (TB * 4 instructions each) are in flight
      const int younger = min(NXB - 2, un - 1 - u);
      if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TB * 4) : "memory");
      else if constexpr (NXB > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TB * 8) : "memory");
      __builtin_amdgcn_s_barrier();  // unit u is in place; unit u - 1 has been multiplied by every wave
      if (u + NXB - 1 < un) issue_x(u + NXB - 1);
    }
  } else {
    // ---- weights: one instruction = 8 rows x 128 bytes (the whole unit of those rows) ----
    // lane (r8, s) fetches piece s ^ key of its row, key = (row's index in its block of 16) >> 1
    unsigned woff[2][2], soff;
    // A row holds M / 64 scales, a multiple of four (16 bytes) but not always of eight: the last
    // four-unit block of a row may be half a block.  Its absent second half belongs to units that do
    // not exist, so those lanes re-read the first half instead of reaching past the row -- for the
    // matrix's last row that would be past the end of the scale array.
    const unsigned srow = (unsigned)(a.M / 64) * 4u, shalf = (unsigned)(lane >> 5) * 16u;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r8 = lane >> 3, key = h * 4 + (r8 >> 1);
        // rows past the end re-read the last row; their results are never stored
        const unsigned n = (unsigned)min(map.base + (h * 8 + r8) * map.istep + rb * map.rbstep, a.K - 1);
        woff[rb][h] = n * (unsigned)a.M + (unsigned)u0 * FR_KU + (unsigned)(((lane & 7) ^ key) * 16);
      }
    {
      // lanes 0-31: the rows' first 16 bytes of a four-unit block of scales, lanes 32-63: the second
      const unsigned n = (unsigned)min(map.base + (lane & 15) * map.istep + ((lane >> 4) & 1) * map.rbstep, a.K - 1);
      soff = n * (unsigned)(a.M / 64) * 4u;
    }
    const char* const wbase = reinterpret_cast<const char*>(a.w);
    const char* const sbase = reinterpret_cast<const char*>(a.scales);
    auto issue_w = [&](int u) {
      if (!requests) return;
      char* dst = ring + (u % FR_RING) * FR_SLOT;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          __builtin_amdgcn_global_load_lds(
              KX_CKN(a, RB_W, reinterpret_cast<const uint32_t*>(wbase + (woff[rb][h] + (unsigned)u * FR_KU)), 16),
              (lds_ptr_t)(dst + (rb * 2 + h) * 1024), 16, 0, 2);  // 2 = nt: streamed once
      // the 32 bytes of scales per row that hold this and the next three units' groups, fetched
      // with the first unit of such a block (or of this part)
      const int U = u0 + u;
      if ((U & 3) == 0 || u == 0)
        __builtin_amdgcn_global_load_lds(
            KX_CKN(a, RB_S,
                   reinterpret_cast<const uint32_t*>(sbase + (soff + min((unsigned)(U >> 2) * 32u + shalf, srow - 16u))),
                   16),
            (lds_ptr_t)(sring + ((U >> 2) % FR_SRING) * 1024), 16, 0, 0);
    };
    // fragment addresses: lane (j, q) = (lane & 15, lane >> 4)
    const int j = lane & 15, q = lane >> 4;
    int rdw[2], rdx[2];
    {
      const int h = j >> 3, r8 = j & 7, keyw = j >> 1;
      const int x = j >> 1, keyx = ((x & 1) << 1) | ((x >> 1) & 1) | (x & 4);
#pragma unroll
      for (int g = 0; g < 2; ++g) rdw[g] = h * 1024 + (r8 * 8 + ((g * 4 + q) ^ keyw)) * 16;
#pragma unroll
      for (int m = 0; m < 2; ++m) rdx[m] = h * 1024 + (r8 * 8 + ((q * 2 + m) ^ keyx)) * 16;
    }

#pragma unroll
    for (int u = 0; u < FR_AHEAD; ++u)
      if (u < un) issue_w(u);
    for (int u = 0; u < un; ++u) {
      // A ring of its own a wave refills before the barrier (the slot's last reads were its own).
      // A shared ring is refilled behind the barrier: the slot of unit u + FR_AHEAD held unit u - 1,
      // which the partner wave may still be reading until it arrives here.
      if (PAIR == 1 && u + FR_AHEAD < un) issue_w(u + FR_AHEAD);
      if (requests) {
        // younger than unit u's: four instructions per unit requested since, one more per block of
        // scales that started among them
        const int k = min(PAIR == 1 ? FR_AHEAD : FR_AHEAD - 1, un - 1 - u);
        wait_dma(4 * k + (((u0 + u + k) >> 2) - ((u0 + u) >> 2)));
      }
      __builtin_amdgcn_s_barrier();
      if (PAIR == 2 && u + FR_AHEAD < un) issue_w(u + FR_AHEAD);
      __builtin_amdgcn_sched_barrier(0);
      const char* slot = ring + (u % FR_RING) * FR_SLOT;
      const char* xb = xbuf + (u % NXB) * XBUF;
      const int U = u0 + u, sel = (U & 1) * 2;
      const char* sslot = sring + ((U >> 2) % FR_SRING) * 1024 + ((U >> 1) & 1) * 512;
      f32x4 sv[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) sv[rb] = *reinterpret_cast<const f32x4*>(sslot + (rb * 16 + j) * 16);
#pragma unroll
      for (int gl = 0; gl < 2; ++gl) {
        f16x8 xa[TBW][2];
#pragma unroll
        for (int tb = 0; tb < TBW; ++tb)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            xa[tb][m] = *reinterpret_cast<const f16x8*>(xb + (gl * TB + tb0 + tb) * 2048 + rdx[m]);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const u32x4 wv = *reinterpret_cast<const u32x4*>(slot + rb * 2048 + rdw[gl]);
          const f16x8 w0 = cvt_w8x8(wv.x, wv.y);
          const f16x8 w1 = cvt_w8x8(wv.z, wv.w);
          const float sc = sel ? (gl ? sv[rb][3] : sv[rb][2]) : (gl ? sv[rb][1] : sv[rb][0]);
#pragma unroll
          for (int tb = 0; tb < TBW; ++tb) {
            f32x4 g = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[tb][0], w0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[tb][1], w1, g, 0, 0, 0);
            // Element by element on purpose (this file is built without SLP vectorisation): as
            // v_pk_fma_f32 with the scale picked by op_sel from the upper half of a register pair,
            // an earlier form of this kernel lost a group's update in lanes 48-63 of the lower
            // result on the device (timing dependent).
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rb][tb][r] = __builtin_fmaf(g[r], sc, acc[rb][tb][r]);
          }
        }
      }
      // this wave's reads of the slot and of the activation unit are done before it requests
      // anything into them again (the slot: next iteration; the unit: wave 4, after the barrier)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }

  const bool computes = wave < NCW;
  if (nsplit > 1) {
    constexpr unsigned PB = TB * 8192u;  // one part's partial tile: [wave][row block][token block][lane] x 16 B
    char* const slab_base = reinterpret_cast<char*>(a.slabs) + (size_t)tile * nsplit * PB;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab_base, 0, (unsigned)nsplit * PB, 0x00020000);
    const unsigned lane_off = (unsigned)((wave & 3) * 2 * TB * 1024 + lane * 16);
    if (computes) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int tb = 0; tb < TBW; ++tb) {
          u32x4 v;
          v.x = __float_as_uint(acc[rb][tb][0]); v.y = __float_as_uint(acc[rb][tb][1]);
          v.z = __float_as_uint(acc[rb][tb][2]); v.w = __float_as_uint(acc[rb][tb][3]);
          const unsigned off = (unsigned)part * PB + lane_off + (unsigned)((rb * TB + tb0 + tb) * 1024);
          KX_TOUCH(a, RB_SLABS, slab_base + off, 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) *flag = atomicInc(KX_CK(a, RB_TICKETS, a.tickets + tile), (unsigned)nsplit - 1u) == (unsigned)nsplit - 1u;
    __syncthreads();
    if (!*flag || !computes) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler order only: device-scope loads below
    switch (nsplit) {
      case 2: rows_combine<2, TB, TBW>(a, slab_base, acc, rs, lane_off, tb0); break;
      case 4: rows_combine<4, TB, TBW>(a, slab_base, acc, rs, lane_off, tb0); break;
      case 8: rows_combine<8, TB, TBW>(a, slab_base, acc, rs, lane_off, tb0); break;
      default: rows_combine<16, TB, TBW>(a, slab_base, acc, rs, lane_off, tb0); break;
    }
  }
  if (!computes) return;

  // ---- rows out: a lane holds tokens 4*(lane>>4)+r of the rows (rb, lane&15) of its wave ----
  const int row0 = map.base + (lane & 15) * map.istep, row1 = row0 + map.rbstep;
  if (a.mode == ROWS_SWIGLU) {
    // swiglu_kernel_cu_fp32 (cuda/swiglu_kernel.cu:4-22) on the finished gate / up totals
#pragma unroll
    for (int tb = 0; tb < TBW; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = (tb0 + tb) * 16 + 4 * (lane >> 4) + r;
        if (t < a.T) {
          const float g = acc[0][tb][r], u = acc[1][tb][r];
          *KX_CK(a, RB_H16, a.h16 + ((size_t)t * a.ldh + row0)) = sat_f16(g * (1.0f / (1.0f + expf(-g))) * u);
        }
      }
    return;
  }
  if (a.mode == ROWS_QKV_ROPE) {
    // bias ahead of the rotation (matmul.cpp:76-79), rope_kernel_cu (cuda/rope_kernel.cu:17-34 half
    // split, :111-127 interleaved) for position pos0 + t on q and k rows; v rows get their bias only
    const bool rotate = row0 < a.k1;
    float* dst;
    int64_t ld;
    int c0;
    [[maybe_unused]] int op;
    if (row0 < a.k0) { dst = a.y0; ld = a.ld0; c0 = row0; op = RB_Y0; }
    else if (row0 < a.k1) { dst = a.y1; ld = a.ld1; c0 = row0 - a.k0; op = RB_Y1; }
    else { dst = a.y2; ld = a.ld2; c0 = row0 - a.k1; op = RB_Y2; }
    const int c1 = c0 + map.rbstep;
    const float b0 = a.bias ? *KX_CK(a, RB_BIAS, a.bias + row0) : 0.f, b1 = a.bias ? *KX_CK(a, RB_BIAS, a.bias + row1) : 0.f;
    const int in_head = row0 % a.head_size;
    const int tab = a.rope_variant == KX_ROPE_HALF_SPLIT ? in_head * 2 : in_head;
#pragma unroll
    for (int tb = 0; tb < TBW; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = (tb0 + tb) * 16 + 4 * (lane >> 4) + r;
        if (t < a.T) {
          float o0 = acc[0][tb][r] + b0, o1 = acc[1][tb][r] + b1;
          if (rotate) {
            const size_t ti = (size_t)(a.pos0 + t) * a.head_size + tab;
            const float fci = *KX_CK(a, RB_SIN, a.sin_cache + ti), fcr = *KX_CK(a, RB_COS, a.cos_cache + ti);
            const float t0 = o0 * fcr - o1 * fci;
            o1 = o0 * fci + o1 * fcr;
            o0 = t0;
          }
          *KX_CK(a, op, dst + ((size_t)t * ld + c0)) = o0;
          *KX_CK(a, op, dst + ((size_t)t * ld + c1)) = o1;
        }
      }
    return;
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int n = rb ? row1 : row0;
    if (n >= a.K) continue;
    float* dst;
    int64_t ld;
    int col;
    [[maybe_unused]] int op;
    bool add_res = false;
    if (n < a.k0) {
      dst = a.y0; ld = a.ld0; col = n; add_res = a.residual != nullptr; op = RB_Y0;
    } else if (n < a.k1) {
      dst = a.y1; ld = a.ld1; col = n - a.k0; op = RB_Y1;
    } else {
      dst = a.y2; ld = a.ld2; col = n - a.k1; op = RB_Y2;
    }
#pragma unroll
    for (int tb = 0; tb < TBW; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = (tb0 + tb) * 16 + 4 * (lane >> 4) + r;
        if (t < a.T) {
          float v = acc[rb][tb][r];
          if (add_res) v += *KX_CK(a, RB_RES, a.residual + ((size_t)t * a.ldr + col));
          *KX_CK(a, op, dst + ((size_t)t * ld + col)) = v;
        }
      }
  }
}

// Parts per tile.  Up to 32 rows two workgroups fit a CU's LDS, and twice the waves keep more
// requests in flight than a deeper ring does: matrices with many tiles are cut until the grid has
// up to two workgroups per CU, narrow ones (where a part is short and the last arriver's sum of S
// slabs shows) up to one.  Measured on Llama-3-8B's matrices, 16 rows: Wq|Wk|Wv 25.7 / 15.8 / 11.1 /
// 11.1 / 15.7 us for 1 / 2 / 4 / 8 / 16 parts, W1|W3 28.8 / 24.6 / 29.2 / 34.9, Wo 8.8 at 8.
static int rows_split_choice(int tiles, int nu, int TB, size_t slab_bytes) {
  const int cap = TB <= 2 && tiles >= 128 ? 512 : 256;
  int S = 1;
  while (S < 16 && tiles * S * 2 <= cap && nu >= 4 * S && (size_t)tiles * (2 * S) * TB * 8192 <= slab_bytes) S *= 2;
  return S;
}

template <int TB, int PAIR>
static int launch_rows(const RowsArgs& a, hipStream_t s) {
  static OncePerDevice raised;
  constexpr int lds = rows_lds_bytes<TB>();
  if (int rc = raised.raise(reinterpret_cast<const void*>(&gemm_w8_rows_kernel<TB, PAIR>), lds)) return rc;
  const int tiles = a.mode == ROWS_SWIGLU ? a.half / 64 : (a.K + FR_BN - 1) / FR_BN;
  hipLaunchKernelGGL((gemm_w8_rows_kernel<TB, PAIR>), dim3(tiles * a.nsplit), dim3(64 * (4 * PAIR + 1)), lds, s,
                     a);
  KX_LAUNCH_CHECK();
  return KX_OK;
}

__global__ void rows_f32_to_f16_kernel(const float* __restrict__ x, int64_t ldx, _Float16* __restrict__ y,
                                       int32_t M, int64_t n4 KX_BPARAM(2)) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / (M / 4), c = (i % (M / 4)) * 4;
    const f32x4 v = *KX_CKP(0, reinterpret_cast<const f32x4*>(x + t * ldx + c));
    f16x4 h;
    h[0] = sat_f16(v[0]); h[1] = sat_f16(v[1]); h[2] = sat_f16(v[2]); h[3] = sat_f16(v[3]);
    *KX_CKP(1, reinterpret_cast<f16x4*>(y + t * M + c)) = h;
  }
}

static int rows_run(const void* X, bool x_half, int64_t ldx, const int8_t* w, const float* scales,
                    int32_t group_size, float* y0, int64_t ld0, float* y1, int64_t ld1, float* y2, int64_t ld2,
                    int32_t k0, int32_t k1, const float* residual, int64_t ldr, int32_t T, int32_t M, int32_t K,
                    void* split_ws, size_t split_ws_bytes, hipStream_t s, const RowsArgs* fused = nullptr) {
  const int mode = fused ? fused->mode : ROWS_PLAIN;
  if (!X || !w || !scales || T <= 0 || M <= 0 || K <= 0 || group_size <= 0) return KX_EINVAL;
  if (!y0 && mode != ROWS_SWIGLU) return KX_EINVAL;
  if (!y1 && !y2) { k0 = K; k1 = K; }
  if (k0 < 0 || k1 < k0 || k1 > K) return KX_EINVAL;
  if ((k0 < K && !y1 && k1 > k0) || (k1 < K && !y2)) return KX_EINVAL;
  const int xalign = x_half ? 8 : 4;
  // M % 256: a row's scales then start on a 16-byte boundary
  // leading dimensions that could not hold the rows they are given would make the kernel read or
  // write the neighbouring rows (and past the buffer on the last one)
  if (ldx < M || (residual && ldr < k0) || (mode != ROWS_SWIGLU && (ld0 < k0 || (y1 && ld1 < k1 - k0) || (y2 && ld2 < K - k1))))
    return KX_EINVAL;
  if (T > 192 || group_size != 64 || M % 256 != 0 || ldx % xalign != 0 || ((uintptr_t)X % 16) != 0 ||
      ((uintptr_t)w % 16) != 0 || ((uintptr_t)scales % 16) != 0)
    return KX_EUNSUPPORTED;
  const int tiles = (K + FR_BN - 1) / FR_BN, nu = M / FR_KU;
  if (tiles > (int)(kRowsTicketBytes / 4)) return KX_EUNSUPPORTED;
  // the kernel addresses its operands with 32-bit byte offsets
  if ((uint64_t)K * (uint64_t)M >= (1ull << 32) || (uint64_t)T * (uint64_t)ldx * 2ull >= (1ull << 32))
    return KX_EUNSUPPORTED;
  char* ws = static_cast<char*>(split_ws);
  const bool have_ws = ws && split_ws_bytes > kRowsTicketBytes + kRowsX16Bytes;
  if (!x_half) {
    // fp32 rows: an fp16 copy in the workspace first (the kernel takes fp16 rows only)
    if (!have_ws || (size_t)T * M * 2 > kRowsX16Bytes) return KX_EUNSUPPORTED;
    _Float16* x16 = reinterpret_cast<_Float16*>(ws + kRowsTicketBytes);
    const int64_t n4 = (int64_t)T * (M / 4);
    hipLaunchKernelGGL(rows_f32_to_f16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                       static_cast<const float*>(X), ldx, x16, M, n4 KX_BPASS(2, X, x16));
    KX_LAUNCH_CHECK();
    X = x16;
    ldx = M;
  }
  const int TB = T <= 16 ? 1 : T <= 32 ? 2 : T <= 64 ? 4 : T <= 96 ? 6 : T <= 128 ? 8 : T <= 160 ? 10 : 12;
  const size_t slab_bytes = have_ws ? split_ws_bytes - kRowsTicketBytes - kRowsX16Bytes : 0;
  int S = rows_split_choice(tiles, nu, TB, slab_bytes);
  if (const char* e = getenv("KX_ROWS_SPLIT")) {  // A/B runs and tests (which switch it between calls): 1, 2, 4, 8 or 16
    const int want = atoi(e);
    if (want == 1 || ((want == 2 || want == 4 || want == 8 || want == 16) && nu >= want &&
                      (size_t)tiles * want * TB * 8192 <= slab_bytes))
      S = want;
  }
  RowsArgs a{};
  a.X = static_cast<const _Float16*>(X); a.w = w; a.scales = scales; a.residual = residual;
  a.y0 = y0; a.y1 = y1 ? y1 : y0; a.y2 = y2 ? y2 : y0;
  a.ldx = ldx; a.ld0 = ld0; a.ld1 = ld1; a.ld2 = ld2; a.ldr = ldr;
  a.k0 = k0; a.k1 = k1; a.T = T; a.M = M; a.K = K;
  a.nsplit = S;
  if (fused) {
    a.mode = fused->mode;
    a.h16 = fused->h16; a.ldh = fused->ldh; a.half = fused->half;
    a.sin_cache = fused->sin_cache; a.cos_cache = fused->cos_cache; a.bias = fused->bias;
    a.head_size = fused->head_size; a.rope_variant = fused->rope_variant; a.pos0 = fused->pos0;
  }
  a.tickets = reinterpret_cast<unsigned*>(ws);
  a.slabs = have_ws ? reinterpret_cast<float*>(ws + kRowsTicketBytes + kRowsX16Bytes) : nullptr;
  KX_BSET(a, RB_X, a.X); KX_BSET(a, RB_W, a.w); KX_BSET(a, RB_S, a.scales); KX_BSET(a, RB_RES, a.residual);
  KX_BSET(a, RB_Y0, a.y0); KX_BSET(a, RB_Y1, a.y1); KX_BSET(a, RB_Y2, a.y2); KX_BSET(a, RB_SLABS, a.slabs);
  KX_BSET(a, RB_TICKETS, a.tickets); KX_BSET(a, RB_H16, a.h16); KX_BSET(a, RB_SIN, a.sin_cache);
  KX_BSET(a, RB_COS, a.cos_cache); KX_BSET(a, RB_BIAS, a.bias);
  switch (TB) {
    case 1: return launch_rows<1, 1>(a, s);
    case 2: return launch_rows<2, 1>(a, s);
#ifdef KX_ROWS_NO_PAIRS
    case 4: return launch_rows<4, 1>(a, s);
    case 6: return launch_rows<6, 1>(a, s);
    case 8: return launch_rows<8, 1>(a, s);
    case 10: return launch_rows<10, 1>(a, s);
    default: return launch_rows<12, 1>(a, s);
#else
    // 64 rows run the same either way (measured); from 96 rows on the paired form is 4-8 % faster
    case 4: return launch_rows<4, 1>(a, s);
    case 6: return launch_rows<6, 2>(a, s);
    case 8: return launch_rows<8, 2>(a, s);
    case 10: return launch_rows<10, 2>(a, s);
    default: return launch_rows<12, 2>(a, s);
#endif
  }
}

}  // namespace kx

extern "C" int kx_gemm_w8_rows_f32(const float* X, int64_t ldx, const int8_t* w, const float* scales,
                                   int32_t group_size, float* y0, int64_t ld0, float* y1, int64_t ld1,
                                   float* y2, int64_t ld2, int32_t k0, int32_t k1, const float* residual,
                                   int64_t ldr, int32_t T, int32_t M, int32_t K, void* split_workspace,
                                   size_t split_workspace_bytes, void* stream) {
  return kx::rows_run(X, false, ldx, w, scales, group_size, y0, ld0, y1, ld1, y2, ld2, k0, k1, residual, ldr,
                      T, M, K, split_workspace, split_workspace_bytes, (hipStream_t)stream);
}

extern "C" int kx_gemm_w8_rows_f16act(const void* X16, int64_t ldx, const int8_t* w, const float* scales,
                                      int32_t group_size, float* y0, int64_t ld0, float* y1, int64_t ld1,
                                      float* y2, int64_t ld2, int32_t k0, int32_t k1, const float* residual,
                                      int64_t ldr, int32_t T, int32_t M, int32_t K, void* split_workspace,
                                      size_t split_workspace_bytes, void* stream) {
  return kx::rows_run(X16, true, ldx, w, scales, group_size, y0, ld0, y1, ld1, y2, ld2, k0, k1, residual, ldr,
                      T, M, K, split_workspace, split_workspace_bytes, (hipStream_t)stream);
}

extern "C" int kx_gemm_w8_rows_swiglu_f16act(const void* X16, int64_t ldx, const int8_t* w13, const float* scales13,
                                             int32_t group_size, void* out16, int64_t ld_out, int32_t T, int32_t M,
                                             int32_t H, void* split_workspace, size_t split_workspace_bytes,
                                             void* stream) {
  if (!out16 || H <= 0 || ld_out < H) return KX_EINVAL;
  if (H % 64 != 0 || ((uintptr_t)out16 % 2) != 0) return KX_EUNSUPPORTED;
  kx::RowsArgs f{};
  f.mode = kx::ROWS_SWIGLU;
  f.h16 = static_cast<_Float16*>(out16); f.ldh = ld_out; f.half = H;
  return kx::rows_run(X16, true, ldx, w13, scales13, group_size, nullptr, 0, nullptr, 0, nullptr, 0, 2 * H, 2 * H,
                      nullptr, 0, T, M, 2 * H, split_workspace, split_workspace_bytes, (hipStream_t)stream, &f);
}

extern "C" int kx_gemm_w8_rows_qkv_rope_f16act(const void* X16, int64_t ldx, const int8_t* w, const float* scales,
                                               int32_t group_size, float* q, int64_t ld_q, float* k, int64_t ld_k,
                                               float* v, int64_t ld_v, int32_t q_dim, int32_t kv_dim,
                                               int32_t head_size, int32_t rope_variant, const float* sin_cache,
                                               const float* cos_cache, int32_t pos0, const float* bias, int32_t T,
                                               int32_t M, void* split_workspace, size_t split_workspace_bytes,
                                               void* stream) {
  if (!q || !k || !v || !sin_cache || !cos_cache || q_dim <= 0 || kv_dim <= 0 || pos0 < 0) return KX_EINVAL;
  if (rope_variant != KX_ROPE_INTERLEAVED && rope_variant != KX_ROPE_HALF_SPLIT) return KX_EINVAL;
  // whole 128-row tiles of one kind (q, k or v), rotation pairs inside a wave's 32 rows
  if ((head_size != 32 && head_size != 64 && head_size != 128) || q_dim % 128 != 0 || kv_dim % 128 != 0)
    return KX_EUNSUPPORTED;
  kx::RowsArgs f{};
  f.mode = kx::ROWS_QKV_ROPE;
  f.sin_cache = sin_cache; f.cos_cache = cos_cache; f.bias = bias;
  f.head_size = head_size; f.rope_variant = rope_variant; f.pos0 = pos0;
  return kx::rows_run(X16, true, ldx, w, scales, group_size, q, ld_q, k, ld_k, v, ld_v, q_dim, q_dim + kv_dim,
                      nullptr, 0, T, M, q_dim + 2 * kv_dim, split_workspace, split_workspace_bytes,
                      (hipStream_t)stream, &f);
}
