// Flash-attention forward over the fp32 KV cache for Tq > 1 query rows (prefill), gfx950.
//
// Stands in for flash_attention_fwd_kernel / flash_attention_kernel_cu
// (kuiper/source/op/kernels/cuda/flash_attention_kernel.cu:40-303).  Semantics are those of the
// layer interface, not of that kernel (which ignores pos, the layer and GQA): query row i sits at
// absolute position pos-(Tq-1)+i and attends cache rows 0..that position (causal) or 0..pos.
//
// Structure
//   - one workgroup = one kv head x a block of query rows for ALL the query heads that share the
//     kv head (wave w -> head w % HPB, 32-row sub-block w / HPB), so a staged K/V tile feeds
//     kv_mul times more MFMA work than it would per head.
//   - with a workspace of kx_flash_attention_workspace_bytes() the layer's K/V rows are converted
//     to fp16 once per call (K row-major, V transposed with its keys in score-register order);
//     tiles of 64 keys then come in by LDS-DMA into XOR-swizzled unpadded images, two buffers,
//     one barrier per tile, the next tile requested at the top of the current one.  Without a
//     workspace (or for caches of at most three tiles) each workgroup rounds the fp32 rows it
//     stages on their way through registers into padded images.
//   - S^T = K.Q^T and O^T = V^T.P^T on v_mfma_f32_32x32x16_f16.  With the key index on the
//     register axis and the query row on the lane, the exponentiated score registers are already
//     the B operand of the second product: probabilities never leave registers, and the
//     online-softmax state (running max, running sum, rescale factor) is one value per lane.
//   - LDS fragments are requested a stage ahead of the MFMAs that use them, and the exponentials
//     of one 16-key step run beside the MFMAs of the step before (scheduling fences between the
//     stages keep hipcc from pulling the reads back next to their uses).
//   - fp32 everywhere outside the two MFMA operand roundings (Q*scale, K, V, P to fp16).
//   - the output tile leaves through LDS as whole row segments (fa_store_rows).
#include <math.h>

#include <type_traits>

#define KX_FILE_ID 6
#include "kx_common.hpp"

namespace kx {

constexpr int FA_KV = 64;       // keys per tile
constexpr int FA_VT_ROW = 68;   // halves per padded V^T row (64 keys + 4), fp32-source path
constexpr size_t FA_WS_SKIP = 16384;  // head of the workspace left alone (decode arrival tickets)

// fp16 copies of one layer's K and V rows 0..pos, made once per call when the caller provides the
// workspace: K16 [kvh][S_pad][HS] row-major; V16T [kvh][HS][S_pad] transposed, and inside every
// block of 16 keys position 8h+4g+e holds key 8g+4h+e -- the order in which the score registers
// of the S^T tile come out -- so the V^T fragment of the second product is one 16-byte read.
__device__ __forceinline__ int fa_key_of_pos(int p) {  // swaps bits 2 and 3; its own inverse
  return (p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1);
}

__device__ __forceinline__ f32x4 fa_load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 fa_load4(const _Float16* p) {  // 16-bit cache: widened, exact
  const f16x4 h = *reinterpret_cast<const f16x4*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

template <int HS, typename KVT>
__global__ __launch_bounds__(256) void kv_to_f16_kernel(const KVT* __restrict__ kc,
                                                        const KVT* __restrict__ vc,
                                                        _Float16* __restrict__ k16,
                                                        _Float16* __restrict__ v16t, int kv_dim,
                                                        int pos, int s_pad KX_BPARAM(4)) {
  constexpr int VROW = HS + 2;  // halves per LDS row: odd word stride for the column gathers
  __shared__ _Float16 vt[FA_KV * VROW];
  const int tid = threadIdx.x;
  const int key0 = blockIdx.x * FA_KV, kvh = blockIdx.y;
  for (int item = tid; item < FA_KV * (HS / 4); item += 256) {
    const int key = item / (HS / 4), d4 = (item % (HS / 4)) * 4;
    const size_t src = (size_t)min(key0 + key, pos) * kv_dim + (size_t)kvh * HS + d4;
    const f32x4 kv = fa_load4(KX_CKPN(0, kc + src, 4 * sizeof(KVT)));
    const f32x4 vv = fa_load4(KX_CKPN(1, vc + src, 4 * sizeof(KVT)));
    f16x4 h;
    h[0] = sat_f16(kv.x); h[1] = sat_f16(kv.y); h[2] = sat_f16(kv.z); h[3] = sat_f16(kv.w);
    *KX_CKP(2, reinterpret_cast<f16x4*>(k16 + ((size_t)kvh * s_pad + key0 + key) * HS + d4)) = h;
    _Float16* row = vt + key * VROW + d4;
    row[0] = sat_f16(vv.x); row[1] = sat_f16(vv.y); row[2] = sat_f16(vv.z); row[3] = sat_f16(vv.w);
  }
  __syncthreads();
  for (int item = tid; item < HS * (FA_KV / 8); item += 256) {
    const int d = item / (FA_KV / 8), c = item % (FA_KV / 8);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = vt[fa_key_of_pos(8 * c + j) * VROW + d];
    *KX_CKP(3, reinterpret_cast<f16x8*>(v16t + ((size_t)kvh * HS + d) * s_pad + key0 + 8 * c)) = o;
  }
}

struct FaArgs {
  const float* q;   // [Tq][H][HS]
  const float* kc;  // layer base [S][kv_dim]
  const float* vc;
  float* out;       // [Tq][H][HS]
  int32_t Tq, H, kv_mul, kv_dim, pos, causal;
  int32_t hpb;      // query heads per workgroup (divides kv_mul)
  int32_t qsubs;    // 32-row sub-blocks per workgroup
  float scale;
  _Float16* out16;       // when set, the output is written here as fp16 instead of to `out`
  const _Float16* k16;   // fp16 copies (see kv_to_f16_kernel), or null
  const _Float16* v16t;
  const _Float16* kc16;  // 16-bit cache: the layer's rows (kc / vc are null then)
  const _Float16* vc16;
  int32_t s_pad;
  KX_BFIELD(7)  // operands of the bounds-checking build, in the order of FaOperand
};
enum FaOperand { PB_Q, PB_KC, PB_VC, PB_OUT, PB_OUT16, PB_K16, PB_V16T };

#ifdef FA_TRACE
// diagnostic build only (tools/fa_trace.py): cycles per loop phase of the heaviest workgroup
__device__ unsigned long long fa_trace_buf[8][16];
#define FA_STAMP(i)                                                 \
  do {                                                              \
    if (traced) {                                                   \
      const unsigned long long now_ = __builtin_readcyclecounter(); \
      tr_acc[i] += now_ - tr_last;                                  \
      tr_last = now_;                                               \
    }                                                               \
  } while (0)
#else
#define FA_STAMP(i)
#endif

// A wave's 32 x HS output tile leaves through LDS: the accumulators hold "lane = query row", so
// direct stores put 16 bytes per lane at a row stride (64 row segments per instruction, every
// 128-byte line written by four instructions); turned around in the wave's own LDS region, an
// instruction stores whole row segments of up to 256 bytes.  Rounds of SEG bytes per row; 16-byte
// chunk c of a row sits at chunk position c ^ key(row) (the K image's rule), so the reads of 16
// consecutive lanes -- one row segment, or two -- cover the banks once.
template <int HS, typename OutT>
__device__ __forceinline__ void fa_store_rows(const FaArgs& a, int operand, char* region,
                                              const f32x16 (&o)[HS / 32], float inv, OutT* out_row0,
                                              size_t row_stride, int rows_valid, int lane) {
  constexpr int ES = sizeof(OutT);
  constexpr int SEG = HS * ES < 256 ? HS * ES : 256;  // bytes per row and round
  constexpr int C = SEG / 16;                          // chunks per row segment
  constexpr int ROUNDS = HS * ES / SEG;
  constexpr int DT_PER_ROUND = (HS / 32) / ROUNDS;
  constexpr int RPI = 64 / C;                          // rows per store instruction
  const int qcol = lane & 31, half = lane >> 5;
  const int wkey = (qcol * C / 16) % C;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
    for (int dtl = 0; dtl < DT_PER_ROUND; ++dtl)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dt = r * DT_PER_ROUND + dtl;
        const f32x4 v = {o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv,
                         o[dt][4 * g + 3] * inv};
        if constexpr (ES == 4) {
          const int chunk = 8 * dtl + 2 * g + half;
          *reinterpret_cast<f32x4*>(region + qcol * SEG + ((chunk ^ wkey) << 4)) = v;
        } else {
          const int chunk = 4 * dtl + g;
          f16x4 h;
          h[0] = sat_f16(v.x); h[1] = sat_f16(v.y); h[2] = sat_f16(v.z); h[3] = sat_f16(v.w);
          *reinterpret_cast<f16x4*>(region + qcol * SEG + ((chunk ^ wkey) << 4) + 8 * half) = h;
        }
      }
    // the wave reads back what its own lanes wrote: LDS serves a wave's instructions in order
#pragma unroll
    for (int j = 0; j < 32 / RPI; ++j) {
      const int row = lane / C + RPI * j, c = lane % C;
      const int rkey = (row * C / 16) % C;
      const i32x4 val = *reinterpret_cast<const i32x4*>(region + row * SEG + ((c ^ rkey) << 4));
      if (row < rows_valid)
        *KX_CK(a, operand, reinterpret_cast<i32x4*>(reinterpret_cast<char*>(out_row0 + (size_t)row * row_stride) +
                                                    r * SEG + c * 16)) = val;
    }
  }
}

// DMA: the K/V tiles come from the fp16 copies by LDS-DMA (global_load_lds: 1 KiB per wave
// instruction straight into LDS, no staging registers, no ds_write pass; waves 0..3 of the
// workgroup bring in a quarter of a tile's pieces each: in an eight-wave workgroup they are the
// waves that otherwise wait at the barrier for their SIMD partners, and who issues made no
// difference to the time).  An instruction fills LDS linearly, so rows cannot
// be padded: 16-byte chunk c of a row sits at chunk position c ^ key(row) instead -- applied to the
// SOURCE address when filling and to the LDS address when reading -- with the key chosen so that
// the 16 rows read by 16 consecutive lanes cover all 64 banks.
// !DMA: the fp32 cache rows are rounded to fp16 on their way through registers into padded images
// (no workspace, or a cache of at most three tiles).
template <int HS, bool DMA>
__device__ __forceinline__ void attn_prefill_body(const FaArgs& a) {
  constexpr int NW = 4;                 // waves that issue the LDS-DMA pieces
  constexpr int K_ROW = HS + 8;         // halves per padded K row (!DMA)
  constexpr int KS = HS / 16;           // 16-dim MFMA steps of Q.K
  constexpr int DT = HS / 32;           // 32-dim output tiles
  constexpr int K_TILE = FA_KV * K_ROW; // halves
  constexpr int VT_ROW = FA_VT_ROW;
  constexpr int V_TILE = HS * VT_ROW;
  constexpr int BUF = K_TILE + V_TILE;
  // DMA images: K [64 keys][HS] and V^T [HS][64 keys], unpadded
  constexpr int DK_ROWB = HS * 2;               // bytes per K row
  constexpr int DK_TILE_B = FA_KV * DK_ROWB;
  constexpr int DV_TILE_B = HS * FA_KV * 2;     // V^T rows are 128 bytes
  constexpr int DBUF_B = DK_TILE_B + DV_TILE_B;
  constexpr int DK_PIECES = DK_TILE_B / 1024;
  constexpr int DPER = DMA ? (DK_TILE_B + DV_TILE_B) / 1024 / NW : 1;  // pieces per wave and tile
  constexpr int DKCH = DK_ROWB / 16;            // 16-byte chunks per K row
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  using lds_ptr_t = __attribute__((address_space(3))) void*;
  _Float16* lds = reinterpret_cast<_Float16*>(smem);

#ifdef FA_TRACE
  const unsigned long long tr_start = __builtin_readcyclecounter();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const int qcol = lane & 31, half = lane >> 5;
  const int groups_per_kvh = a.kv_mul / a.hpb;
  const int kvh = blockIdx.x / groups_per_kvh;
  const int hgroup = blockIdx.x % groups_per_kvh;
  const int head = kvh * a.kv_mul + hgroup * a.hpb + (wave % a.hpb);
  const int rows_per_block = 32 * a.qsubs;
  // heavy (late, long-context) query blocks first
  const int qblock = (int)gridDim.y - 1 - (int)blockIdx.y;
  const int q0_block = qblock * rows_per_block;
  const int q0 = q0_block + (wave / a.hpb) * 32;       // first query row of this wave
  const int qpos0 = a.pos - (a.Tq - 1);                // absolute position of query row 0
  const int my_row = q0 + qcol;
  const int my_pos = a.causal ? min(qpos0 + my_row, a.pos) : a.pos;

  // keys this workgroup needs: up to the position of its last query row
  const int last_row_block = min(q0_block + rows_per_block, a.Tq) - 1;
  const int max_key = a.causal ? qpos0 + last_row_block : a.pos;
  const int ntiles = max_key / FA_KV + 1;

  // ---- LDS-DMA roles: piece p = wave + NW i of a tile; pieces 0..DK_PIECES-1 are K rows, the
  //      rest V^T rows; lane l fills bytes 16 l .. 16 l + 15 of the piece ----
  // per piece: a wave-uniform base (the kv head's K or V^T image) and this lane's byte offset, which
  // moves on by a wave-uniform step from one tile to the next
  const char* dbase[DPER];
  unsigned doff[DPER], dstep[DPER];
  if constexpr (DMA) {
    const char* k16base = reinterpret_cast<const char*>(a.k16 + (size_t)kvh * a.s_pad * HS);
    const char* v16base = reinterpret_cast<const char*>(a.v16t + (size_t)kvh * HS * a.s_pad);
#pragma unroll
    for (int i = 0; i < DPER; ++i) {
      const int p = wave % NW + NW * i;
      if (p < DK_PIECES) {
        const int row = p * (1024 / DK_ROWB) + lane / DKCH, pos = lane % DKCH;
        const int chunk = pos ^ ((row * DKCH / 16) % DKCH);
        dbase[i] = k16base;
        doff[i] = (unsigned)(row * DK_ROWB + 16 * chunk);
        dstep[i] = FA_KV * DK_ROWB;
      } else {
        const int d = (p - DK_PIECES) * 8 + (lane >> 3), pos = lane & 7;
        const int chunk = pos ^ ((d >> 1) & 7);
        dbase[i] = v16base;
        doff[i] = (unsigned)(d * a.s_pad * 2 + 16 * chunk);
        dstep[i] = FA_KV * 2;
      }
    }
  }
  auto issue_tile = [&](int t, int buf) {
    if (wave >= NW) return;
#pragma unroll
    for (int i = 0; i < DPER; ++i)
      __builtin_amdgcn_global_load_lds(
          KX_CKN(a, (wave % NW + NW * i) < DK_PIECES ? PB_K16 : PB_V16T, dbase[i] + (doff[i] + (unsigned)t * dstep[i]), 16),
          (lds_ptr_t)(smem + buf * DBUF_B + (wave + NW * i) * 1024), 16, 0, 0);
  };
  if constexpr (DMA) issue_tile(0, 0);
  // Waves 4..7 of an eight-wave workgroup are the later-dispatched wave of each SIMD and lose the
  // issue arbitration to their partners in every phase, which then wait for them at the barrier;
  // one static priority step evens that out (158.7 -> 155.9 us on 32 heads x 4096 rows).
  if (nthreads == 512 && wave >= 4) __builtin_amdgcn_s_setprio(1);

  // ---- Q fragment (B operand): lane (q row, half) holds Q[row][16ks + 8half + j] * scale ----
  // scores are kept in the log2 domain: Q is scaled by softmax_scale * log2(e)
  constexpr float kLog2e = 1.4426950408889634f;
  const float qscale_all = a.scale * kLog2e;
  f16x8 qf[KS];
  {
    // rows past the end re-read the last row and scale it by zero, so all the loads are requested
    // at once (under a lane test hipcc branches around each pair and waits inside some branches)
    const float* qp = a.q + ((size_t)min(my_row, a.Tq - 1) * a.H + head) * HS + 8 * half;
    f32x4 qraw[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qraw[ks][0] = *KX_CK(a, PB_Q, reinterpret_cast<const f32x4*>(qp + 16 * ks));
      qraw[ks][1] = *KX_CK(a, PB_Q, reinterpret_cast<const f32x4*>(qp + 16 * ks + 4));
    }
    const float qscale = my_row < a.Tq ? qscale_all : 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 v0 = qraw[ks][0], v1 = qraw[ks][1];
      qf[ks][0] = (_Float16)(v0.x * qscale);
      qf[ks][1] = (_Float16)(v0.y * qscale);
      qf[ks][2] = (_Float16)(v0.z * qscale);
      qf[ks][3] = (_Float16)(v0.w * qscale);
      qf[ks][4] = (_Float16)(v1.x * qscale);
      qf[ks][5] = (_Float16)(v1.y * qscale);
      qf[ks][6] = (_Float16)(v1.z * qscale);
      qf[ks][7] = (_Float16)(v1.w * qscale);
    }
  }

  // ---- fragment addresses (bytes inside a buffer) ----
  // DMA: K row qcol (+32 kt), chunk 2ks+half swizzled; V^T row qcol (+32 dt), chunk 4kt+2st+half
  // (LDS pointers of buffer 0, worked out once: inside the loop a fragment address is one of these
  // plus a compile-time offset for buffer, key half and output tile, i.e. an instruction immediate)
  // Each is pinned in a register of its own (the empty asm): left to itself hipcc splits them into
  // a common part and a per-step part and adds the two again for every read of every tile.
  using lds_cptr_t = const __attribute__((address_space(3))) char*;
  using lds_frag_t = const __attribute__((address_space(3))) f16x8*;
  lds_cptr_t kfrag[DMA ? KS : 1];
  lds_cptr_t vfrag[DMA ? 4 : 1];
  if constexpr (DMA) {
    const int kkey = (qcol * DKCH / 16) % DKCH, vkey = (qcol >> 1) & 7;
    lds_cptr_t base = (lds_cptr_t)smem;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kfrag[ks] = base + qcol * DK_ROWB + (((2 * ks + half) ^ kkey) << 4);
      asm volatile("" : "+v"(kfrag[ks]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vfrag[j] = base + DK_TILE_B + qcol * 128 + (((2 * j + half) ^ vkey) << 4);
      asm volatile("" : "+v"(vfrag[j]));
    }
  }

  // ---- register staging roles (!DMA) ----
  // K: item = (key, 4-dim group): 64 * HS/4 items.  V: item = (4-key group, 4-dim group).
  constexpr int K_ITEMS = FA_KV * (HS / 4);
  constexpr int V_ITEMS = (FA_KV / 4) * (HS / 4);
  constexpr int MAXT = 256;  // fewest threads a launch uses
  constexpr int K_PER = (K_ITEMS + MAXT - 1) / MAXT;
  constexpr int V_PER = (V_ITEMS + MAXT - 1) / MAXT;
  f32x4 kreg[DMA ? 1 : K_PER];
  f32x4 vreg[DMA ? 1 : V_PER][4];
  const float* kbase = a.kc + (size_t)kvh * HS;
  const float* vbase = a.vc + (size_t)kvh * HS;

  auto load_tile = [&](int t) {
    const int key0 = t * FA_KV;
#pragma unroll
    for (int i = 0; i < K_PER; ++i) {
      const int item = tid + i * nthreads;
      if (item < K_ITEMS) {
        const int key = item / (HS / 4), dg = item % (HS / 4);
        kreg[i] = *KX_CK(a, PB_KC, reinterpret_cast<const f32x4*>(
            kbase + (size_t)min(key0 + key, a.pos) * a.kv_dim + 4 * dg));
      }
    }
#pragma unroll
    for (int i = 0; i < V_PER; ++i) {
      const int item = tid + i * nthreads;
      if (item < V_ITEMS) {
        const int kg = item / (HS / 4), dg = item % (HS / 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          vreg[i][e] = *KX_CK(a, PB_VC, reinterpret_cast<const f32x4*>(
              vbase + (size_t)min(key0 + 4 * kg + e, a.pos) * a.kv_dim + 4 * dg));
      }
    }
  };
  auto store_tile = [&](int buf) {
    _Float16* Kl = lds + buf * BUF;
    _Float16* Vl = Kl + K_TILE;
#pragma unroll
    for (int i = 0; i < K_PER; ++i) {
      const int item = tid + i * nthreads;
      if (item < K_ITEMS) {
        const int key = item / (HS / 4), dg = item % (HS / 4);
        f16x4 h;
        h[0] = (_Float16)kreg[i].x; h[1] = (_Float16)kreg[i].y;
        h[2] = (_Float16)kreg[i].z; h[3] = (_Float16)kreg[i].w;
        *reinterpret_cast<f16x4*>(Kl + key * K_ROW + 4 * dg) = h;
      }
    }
#pragma unroll
    for (int i = 0; i < V_PER; ++i) {
      const int item = tid + i * nthreads;
      if (item < V_ITEMS) {
        const int kg = item / (HS / 4), dg = item % (HS / 4);
        // 4 keys x 4 dims in registers -> four 4-key runs of V^T
        const float vals[4][4] = {{vreg[i][0].x, vreg[i][1].x, vreg[i][2].x, vreg[i][3].x},
                                  {vreg[i][0].y, vreg[i][1].y, vreg[i][2].y, vreg[i][3].y},
                                  {vreg[i][0].z, vreg[i][1].z, vreg[i][2].z, vreg[i][3].z},
                                  {vreg[i][0].w, vreg[i][1].w, vreg[i][2].w, vreg[i][3].w}};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          f16x4 h;
          h[0] = (_Float16)vals[d][0]; h[1] = (_Float16)vals[d][1];
          h[2] = (_Float16)vals[d][2]; h[3] = (_Float16)vals[d][3];
          *reinterpret_cast<f16x4*>(Vl + (4 * dg + d) * VT_ROW + 4 * kg) = h;
        }
      }
    }
  };

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  if constexpr (DMA) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
#ifdef FA_TRACE
  const bool traced = blockIdx.x == 0 && blockIdx.y == 0;
  unsigned long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tr_last = __builtin_readcyclecounter();
  const unsigned long long tr_c0 = tr_last, tr_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  // tiles past this wave's own last query row contribute nothing to it
  const int wave_last_pos = a.causal ? min(qpos0 + min(q0 + 31, a.Tq - 1), a.pos) : a.pos;

  auto compute_tile = [&](int t, int buf) {
    const int key0 = t * FA_KV;
    if (key0 <= wave_last_pos) {
      const _Float16* Kl = lds + buf * BUF;
      const _Float16* Vl = Kl + K_TILE;
      // ---- S^T[key][q] for the tile's two 32-key halves ----
      f32x16 s[2];
      f16x8 vf[2][DT];  // V^T fragments of two 16-key steps (DMA form)
      if constexpr (DMA) {
        // fragments are requested a stage ahead of the MFMAs that use them (read, then multiplied
        // at once, every pair of MFMAs waited a full LDS round trip), half a 32-key block of K at
        // a time, so the first MFMAs of a tile wait for four reads, not eight; the first V^T
        // step's fragments travel under the last stage's MFMAs
        constexpr int KH = KS / 2 > 0 ? KS / 2 : 1;
        f16x8 ka[KS], kb[KS];
        auto kread = [&](f16x8 (&dst)[KS], int kt, int k0, int k1) {
#pragma unroll
          for (int ks = k0; ks < k1; ++ks)
            dst[ks] = *(lds_frag_t)(kfrag[ks] + buf * DBUF_B + kt * 32 * DK_ROWB);
        };
        auto kmul = [&](f32x16& acc, const f16x8 (&src)[KS], int k0, int k1) {
#pragma unroll
          for (int ks = k0; ks < k1; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(src[ks], qf[ks], acc, 0, 0, 0);
        };
        kread(ka, 0, 0, KH);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[0][r] = 0.f;
// hipcc waits for ALL outstanding LDS reads in front of the first MFMA that needs one (always
// lgkmcnt(0) in a kernel with LDS-DMA), so a stage opens with one MFMA -- whose operands were
// requested a stage ago -- then requests the next stage's fragments, then issues its other MFMAs
#define FA_STAGE_ORDER(reads, mfmas)                                       \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           /* MFMA */  \
  __builtin_amdgcn_sched_group_barrier(0x100, reads, 0);       /* reads */ \
  __builtin_amdgcn_sched_group_barrier(0x008, (mfmas) - 1, 0);             \
  __builtin_amdgcn_sched_barrier(0)
        kread(ka, 0, KH, KS);
        kmul(s[0], ka, 0, KH);
        FA_STAGE_ORDER(KS - KH, KH);
        kread(kb, 1, 0, KH);
        kmul(s[0], ka, KH, KS);
        FA_STAGE_ORDER(KH, KS - KH);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[1][r] = 0.f;
        kread(kb, 1, KH, KS);
        kmul(s[1], kb, 0, KH);
        FA_STAGE_ORDER(KS - KH, KH);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          vf[0][dt] = *(lds_frag_t)(vfrag[0] + buf * DBUF_B + dt * 32 * 128);
        kmul(s[1], kb, KH, KS);
        FA_STAGE_ORDER(DT, KS - KH);
      } else {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
          const _Float16* kp = Kl + (kt * 32 + qcol) * K_ROW + 8 * half;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const f16x8 kf = *reinterpret_cast<const f16x8*>(kp + 16 * ks);
            s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kt], 0, 0, 0);
          }
        }
      }
      FA_STAMP(2);
      // ---- mask + online softmax (one query row per lane; the other half-wave holds the
      //      other 32 keys of the same row) ----
      // only the tiles on the diagonal (or at the end of the cache) hold keys some row must not
      // see: a wave-uniform test keeps the masking code off the common path
      if (__any(key0 + FA_KV - 1 > my_pos)) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key > my_pos) s[kt][r] = -INFINITY;
          }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
      mx = xor32_max(mx);
      // Lazy running max: the reference point of the exponentials only moves when some row's
      // new maximum exceeds the old one by more than 2^kLazy (scores are in the log2 domain).
      // Softmax does not depend on the reference point; keeping it makes alpha = 1 and skips the
      // rescale of the whole output tile, and p <= 2^kLazy stays far inside fp16 range.
      constexpr float kLazy = 8.0f;
      if (__any(mx > m_run + kLazy)) {  // also every row's first unmasked tile (m_run = -inf)
        const float m_new = fmaxf(m_run, mx);
        // a row whose keys are all masked so far keeps m = -inf; guard the inf - inf
        const float m_old = m_run == -INFINITY ? 0.f : m_run;
        const float m_ref = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_old - m_ref);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
      FA_STAMP(3);
      const float m_use = m_run == -INFINITY ? 0.f : m_run;
      float psum = 0.f;
      if constexpr (DMA) {
        // ---- O^T[d][q] += V^T . P^T in four 16-key steps g = 2kt + st; registers 8st..8st+7 of
        //      score tile kt are the B fragment of the step, element j <-> key 16st + 8(j>>2) +
        //      4half + (j&3), the order the V^T image keeps its keys in.  Step g's MFMAs run beside
        //      the exponentials of step g+1 and the V^T reads of step g+1 ----
        auto probs = [&](int g, f16x8& pf) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            // raw v_exp_f32: the argument is <= kLazy, and a result below 2^-126 may flush to zero
            const float p = __builtin_amdgcn_exp2f(s[g >> 1][8 * (g & 1) + j] - m_use);
            psum += p;
            pf[j] = (_Float16)p;
          }
        };
        f16x8 pf[2];
        probs(0, pf[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (g + 1 < 4) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
              vf[(g + 1) & 1][dt] =
                  *(lds_frag_t)(vfrag[g + 1] + buf * DBUF_B + dt * 32 * 128);
          }
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[g & 1][dt], pf[g & 1], o[dt], 0, 0, 0);
          if (g + 1 < 4) probs(g + 1, pf[(g + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        l_run += psum;
      } else {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            // raw v_exp_f32: the argument is <= kLazy, and a result below 2^-126 may flush to zero
            const float p = __builtin_amdgcn_exp2f(s[kt][r] - m_use);
            s[kt][r] = p;
            psum += p;
          }
        l_run += psum;
        // ---- O^T[d][q] += V^T . P^T ; registers 8s..8s+7 of a score tile are the B fragment of
        //      16-key step s, element j <-> key 16s + 8(j>>2) + 4half + (j&3) ----
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (_Float16)s[kt][8 * st + j];
            const _Float16* vp = Vl + qcol * VT_ROW + kt * 32 + 16 * st + 4 * half;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              const f16x4 lo = *reinterpret_cast<const f16x4*>(vp + dt * 32 * VT_ROW);
              const f16x4 hi = *reinterpret_cast<const f16x4*>(vp + dt * 32 * VT_ROW + 8);
              f16x8 vfr;
              vfr[0] = lo[0]; vfr[1] = lo[1]; vfr[2] = lo[2]; vfr[3] = lo[3];
              vfr[4] = hi[0]; vfr[5] = hi[1]; vfr[6] = hi[2]; vfr[7] = hi[3];
              o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfr, pf, o[dt], 0, 0, 0);
            }
          }
      }
    }
  };

  if constexpr (DMA) {
    // two buffers, one barrier per tile: the pieces of tile t+1 are requested at the top of
    // interval t, into the buffer whose tile every wave left behind at the last barrier.  The loop
    // is written out for a pair of tiles so that the buffer is a compile-time constant and every
    // fragment address is a register plus an immediate.
    auto interval = [&](int t, auto buf_c) {
      constexpr int buf = decltype(buf_c)::value;
      FA_STAMP(0);
      if (t + 1 < ntiles) issue_tile(t + 1, buf ^ 1);
      FA_STAMP(1);
      compute_tile(t, buf);
      FA_STAMP(4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next tile are in
      FA_STAMP(5);
      __syncthreads();
      FA_STAMP(6);
    };
    for (int t = 0; t < ntiles; t += 2) {
      interval(t, std::integral_constant<int, 0>{});
      if (t + 1 < ntiles) interval(t + 1, std::integral_constant<int, 1>{});
    }
  } else {
    for (int t = 0; t < ntiles; ++t) {
      const int buf = t & 1;
      if (t + 1 < ntiles) load_tile(t + 1);
      compute_tile(t, buf);
      if (t + 1 < ntiles) store_tile(buf ^ 1);
      __syncthreads();
    }
  }
#ifdef FA_TRACE
  if (traced && lane == 0) {
    for (int i = 0; i < 8; ++i) fa_trace_buf[wave][i] = tr_acc[i];
    fa_trace_buf[wave][8] = __builtin_readcyclecounter() - tr_c0;
    fa_trace_buf[wave][9] = __builtin_amdgcn_s_memrealtime() - tr_r0;
    fa_trace_buf[wave][10] = ntiles;
    fa_trace_buf[wave][11] = tr_c0 - tr_start;
  }
  const unsigned long long tr_loop_end = __builtin_readcyclecounter();
#endif

  // ---- epilogue: lane (q row, half) holds O[row][32dt + 8g + 4half + e] in o[dt][4g+e] ----
  l_run = xor32_sum(l_run);
  const float inv = 1.0f / l_run;
  bool stored = false;
  if constexpr (DMA) {
    // every wave is past the last tile's barrier: the tile buffers are free
    const int nwaves = nthreads >> 6;
    const size_t head_off = ((size_t)q0 * a.H + head) * HS;
    const int rows_valid = a.Tq - q0;  // <= 0: a wave whose rows are all past the end
    if (a.out16) {
      constexpr int RB = 32 * (HS * 2 < 256 ? HS * 2 : 256);
      if (nwaves * RB <= 2 * DBUF_B) {
        fa_store_rows<HS, _Float16>(a, PB_OUT16, smem + wave * RB, o, inv, a.out16 + head_off, (size_t)a.H * HS,
                                    rows_valid, lane);
        stored = true;
      }
    } else {
      constexpr int RB = 32 * (HS * 4 < 256 ? HS * 4 : 256);
      if (nwaves * RB <= 2 * DBUF_B) {
        fa_store_rows<HS, float>(a, PB_OUT, smem + wave * RB, o, inv, a.out + head_off, (size_t)a.H * HS, rows_valid,
                                 lane);
        stored = true;
      }
    }
  }
  if (!stored && my_row < a.Tq) {
    const size_t off = ((size_t)my_row * a.H + head) * HS + 4 * half;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv,
                   o[dt][4 * g + 3] * inv};
        if (a.out16) {
          f16x4 h;
          h[0] = sat_f16(v.x); h[1] = sat_f16(v.y); h[2] = sat_f16(v.z); h[3] = sat_f16(v.w);
          *KX_CK(a, PB_OUT16, reinterpret_cast<f16x4*>(a.out16 + off + 32 * dt + 8 * g)) = h;
        } else {
          *KX_CK(a, PB_OUT, reinterpret_cast<f32x4*>(a.out + off + 32 * dt + 8 * g)) = v;
        }
      }
  }
#ifdef FA_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (traced && lane == 0) fa_trace_buf[wave][12] = __builtin_readcyclecounter() - tr_loop_end;
#endif
}

// the body lives in a device-only function: the host pass then never instantiates its LDS-DMA calls
template <int HS, bool DMA>
__global__ __launch_bounds__(512) void attn_prefill_kernel(const FaArgs a) {
  attn_prefill_body<HS, DMA>(a);
}

template <int HS, bool DMA>
static int launch_prefill_kernel(const FaArgs& a, int kvh, int threads, int qblocks, hipStream_t s) {
  constexpr size_t lds = DMA ? 2 * (size_t)(2 * FA_KV * HS * 2)
                             : 2 * (size_t)(FA_KV * (HS + 8) + HS * FA_VT_ROW) * 2;
  static OncePerDevice raised;
  if (lds > 64 * 1024)
    if (int rc = raised.raise(reinterpret_cast<const void*>(&attn_prefill_kernel<HS, DMA>), (int)lds)) return rc;
  const dim3 grid(kvh * (a.kv_mul / a.hpb), qblocks);
  hipLaunchKernelGGL((attn_prefill_kernel<HS, DMA>), grid, dim3(threads), lds, s, a);
  KX_LAUNCH_CHECK();
  return KX_OK;
}
template <int HS>
static int launch_prefill(FaArgs a, int kvh, int threads, int qblocks, void* workspace,
                          size_t workspace_bytes, hipStream_t s) {
  const int s_pad = (a.pos / FA_KV + 1) * FA_KV;
  const size_t half_bytes = (size_t)kvh * s_pad * HS * 2;
  const bool have_ws = workspace && workspace_bytes >= FA_WS_SKIP + 2 * half_bytes;
  if (a.kc16 && !have_ws) return KX_EWORKSPACE;  // a 16-bit cache is only read through the copies
  // up to three key tiles the copies cost a launch (4 us) for nothing: the tiles are converted once
  // or twice each on their way into LDS anyway
  if (have_ws && (a.kc16 || a.pos >= 3 * FA_KV)) {
    char* ws = static_cast<char*>(workspace) + FA_WS_SKIP;
    _Float16* k16 = reinterpret_cast<_Float16*>(ws);
    _Float16* v16t = reinterpret_cast<_Float16*>(ws + half_bytes);
    if (a.kc16)
      hipLaunchKernelGGL((kv_to_f16_kernel<HS, _Float16>), dim3(s_pad / FA_KV, kvh), dim3(256), 0, s,
                         a.kc16, a.vc16, k16, v16t, a.kv_dim, a.pos, s_pad KX_BPASS(4, a.kc16, a.vc16, k16, v16t));
    else
      hipLaunchKernelGGL((kv_to_f16_kernel<HS, float>), dim3(s_pad / FA_KV, kvh), dim3(256), 0, s, a.kc,
                         a.vc, k16, v16t, a.kv_dim, a.pos, s_pad KX_BPASS(4, a.kc, a.vc, k16, v16t));
    KX_LAUNCH_CHECK();
    a.k16 = k16;
    a.v16t = v16t;
    a.s_pad = s_pad;
    KX_BSET(a, PB_K16, a.k16); KX_BSET(a, PB_V16T, a.v16t);
    return launch_prefill_kernel<HS, true>(a, kvh, threads, qblocks, s);
  }
  return launch_prefill_kernel<HS, false>(a, kvh, threads, qblocks, s);
}

}  // namespace kx

extern "C" size_t kx_flash_attention_workspace_bytes(int32_t pos, int32_t kv_head_num,
                                                     int32_t head_size) {
  if (pos < 0 || kv_head_num <= 0 || head_size <= 0) return 0;
  const size_t s_pad = (size_t)(pos / kx::FA_KV + 1) * kx::FA_KV;
  return kx::FA_WS_SKIP + 2 * (size_t)kv_head_num * s_pad * head_size * 2;
}

static int flash_attention_run(const float* query, const void* key_cache_any,
                               const void* value_cache_any, bool kv16, float* out, _Float16* out16,
                               int32_t Tq, int32_t head_num,
                               int32_t kv_head_num, int32_t head_size, int32_t seq_len, int32_t pos,
                               int32_t layer_idx, float softmax_scale, int32_t is_causal,
                               void* workspace, size_t workspace_bytes, void* stream) {
  using namespace kx;
  const float* key_cache = kv16 ? nullptr : static_cast<const float*>(key_cache_any);
  const float* value_cache = kv16 ? nullptr : static_cast<const float*>(value_cache_any);
  const _Float16* key_cache16 = kv16 ? static_cast<const _Float16*>(key_cache_any) : nullptr;
  const _Float16* value_cache16 = kv16 ? static_cast<const _Float16*>(value_cache_any) : nullptr;
  if (!query || !key_cache_any || !value_cache_any || (!out && !out16)) return KX_EINVAL;
  if (Tq <= 0 || head_num <= 0 || kv_head_num <= 0 || head_num % kv_head_num) return KX_EINVAL;
  if (pos < 0 || pos >= seq_len || pos - (Tq - 1) < 0) return KX_EINVAL;
  const int kv_mul = head_num / kv_head_num;
  const int kv_dim = kv_head_num * head_size;
  const float default_scale = 1.0f / sqrtf((float)head_size);
  const bool tiled = (head_size == 32 || head_size == 64 || head_size == 128) &&
                     (((uintptr_t)query | (uintptr_t)key_cache_any | (uintptr_t)value_cache_any |
                       (uintptr_t)out) % 16 == 0);
  if (out16 && (Tq == 1 || !tiled || (uintptr_t)out16 % 8)) return KX_EUNSUPPORTED;
  if (Tq == 1 || !tiled) {
    // single rows (and head sizes the tiled kernel is not built for) go through the decode
    // kernel row by row; it takes the default scale only
    if (fabsf(softmax_scale - default_scale) > 1e-6f * default_scale) return KX_EUNSUPPORTED;
    for (int32_t i = 0; i < Tq; ++i) {
      const int32_t p = is_causal ? pos - (Tq - 1) + i : pos;
      const float* qi = query + (size_t)i * head_num * head_size;
      float* oi = out + (size_t)i * head_num * head_size;
      int rc = kv16 ? kx_mha_decode_kv16(qi, key_cache16, value_cache16, oi, p, nullptr, head_num, kv_mul,
                                         head_size, kv_dim, seq_len, layer_idx, workspace,
                                         workspace_bytes, stream)
                    : kx_mha_decode_f32(qi, key_cache, value_cache, oi, p, head_num, kv_mul, head_size,
                                        kv_dim, seq_len, layer_idx, workspace, workspace_bytes, stream);
      if (rc) return rc;
    }
    return KX_OK;
  }
  // query heads per workgroup: the largest divisor of kv_mul that fits 8 waves
  int hpb = 1;
  for (int d = 1; d <= 8 && d <= kv_mul; ++d)
    if (kv_mul % d == 0) hpb = d;
  int qsubs = hpb >= 4 ? 1 : 4 / hpb;
  if (hpb * qsubs < 4) qsubs = (4 + hpb - 1) / hpb;
  // With enough query rows a workgroup takes twice the rows (8 waves): a staged K/V tile then
  // feeds twice the MFMA work (182 vs 194 us on 32 heads x 4096 rows).
  if (hpb * qsubs * 2 <= 8 && Tq >= 64 * qsubs * 2) qsubs *= 2;
  // Fewer than two such workgroups per CU (short prompts; the few heads of a tensor-parallel
  // rank): the causal triangle leaves the CUs that drew light query blocks idle while the heavy
  // ones finish.  Four-wave workgroups (two per CU) double the count and pair heavy with light:
  // 8 heads x 8192 rows 257 -> 236 us, 32 heads x 2048 rows 87 -> 78 us; from 512 workgroups on the
  // eight-wave form wins (32 heads x 4096 rows: 186 vs 197 us).
  {
    const int qblocks8 = (Tq + 32 * qsubs - 1) / (32 * qsubs);
    // Heads of at most 64 dims need half the registers: four four-wave workgroups fit a CU and
    // overlap each other's prologues and epilogues (32 heads x 64 x 4096 rows: 110 -> 101 us).
    const bool small_grid = (long)kv_head_num * (kv_mul / hpb) * qblocks8 < 512;
    if ((small_grid || head_size <= 64) && hpb * qsubs > 4) {
      int h4 = 1;
      for (int d = 1; d <= 4 && d <= kv_mul; ++d)
        if (kv_mul % d == 0) h4 = d;
      hpb = h4;
      qsubs = (4 + hpb - 1) / hpb;
    }
  }
  const int threads = 64 * hpb * qsubs;
  const size_t layer_off = (size_t)layer_idx * (size_t)seq_len * (size_t)kv_dim;
  FaArgs a{};
  a.q = query; a.out = out;
  if (kv16) {
    a.kc16 = key_cache16 + layer_off; a.vc16 = value_cache16 + layer_off;
  } else {
    a.kc = key_cache + layer_off; a.vc = value_cache + layer_off;
  }
  a.out16 = out16;
  a.Tq = Tq; a.H = head_num; a.kv_mul = kv_mul; a.kv_dim = kv_dim; a.pos = pos;
  a.causal = is_causal ? 1 : 0; a.hpb = hpb; a.qsubs = qsubs; a.scale = softmax_scale;
  const int qblocks = (Tq + 32 * qsubs - 1) / (32 * qsubs);
  KX_BSET(a, PB_Q, a.q); KX_BSET(a, PB_KC, a.kc); KX_BSET(a, PB_VC, a.vc); KX_BSET(a, PB_OUT, a.out);
  KX_BSET(a, PB_OUT16, a.out16); KX_BSET(a, PB_K16, nullptr); KX_BSET(a, PB_V16T, nullptr);
  hipStream_t s = (hipStream_t)stream;
  switch (head_size) {
    case 32: return launch_prefill<32>(a, kv_head_num, threads, qblocks, workspace, workspace_bytes, s);
    case 64: return launch_prefill<64>(a, kv_head_num, threads, qblocks, workspace, workspace_bytes, s);
    default: return launch_prefill<128>(a, kv_head_num, threads, qblocks, workspace, workspace_bytes, s);
  }
}

extern "C" int kx_flash_attention_f32(const float* query, const float* key_cache,
                                      const float* value_cache, float* out, int32_t Tq,
                                      int32_t head_num, int32_t kv_head_num, int32_t head_size,
                                      int32_t seq_len, int32_t pos, int32_t layer_idx,
                                      float softmax_scale, int32_t is_causal, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (!out) return KX_EINVAL;
  return flash_attention_run(query, key_cache, value_cache, false, out, nullptr, Tq, head_num, kv_head_num,
                             head_size, seq_len, pos, layer_idx, softmax_scale, is_causal, workspace,
                             workspace_bytes, stream);
}

extern "C" int kx_flash_attention_f16out(const float* query, const float* key_cache,
                                         const float* value_cache, void* out16, int32_t Tq,
                                         int32_t head_num, int32_t kv_head_num, int32_t head_size,
                                         int32_t seq_len, int32_t pos, int32_t layer_idx,
                                         float softmax_scale, int32_t is_causal, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (!out16) return KX_EINVAL;
  return flash_attention_run(query, key_cache, value_cache, false, nullptr, static_cast<_Float16*>(out16),
                             Tq, head_num, kv_head_num, head_size, seq_len, pos, layer_idx,
                             softmax_scale, is_causal, workspace, workspace_bytes, stream);
}

extern "C" int kx_flash_attention_kv16(const float* query, const void* key_cache16,
                                       const void* value_cache16, float* out, void* out16, int32_t Tq,
                                       int32_t head_num, int32_t kv_head_num, int32_t head_size,
                                       int32_t seq_len, int32_t pos, int32_t layer_idx,
                                       float softmax_scale, int32_t is_causal, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  if ((out == nullptr) == (out16 == nullptr)) return KX_EINVAL;  // exactly one output form
  return flash_attention_run(query, key_cache16, value_cache16, true, out, static_cast<_Float16*>(out16),
                             Tq, head_num, kv_head_num, head_size, seq_len, pos, layer_idx,
                             softmax_scale, is_causal, workspace, workspace_bytes, stream);
}

#ifdef FA_TRACE
extern "C" int kx_fa_trace_read(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(kx::fa_trace_buf), sizeof(kx::fa_trace_buf)) == hipSuccess ? 0 : 1;
}
#endif
