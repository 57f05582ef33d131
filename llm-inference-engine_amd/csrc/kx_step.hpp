// The end of a decode step on the device -- greedy pick over (value, index) pairs, token / position /
// history bookkeeping, the next step's embedding row -- shared by the stand-alone step kernels
// (support.hip) and the classifier GEMV that finishes the step itself (gemv_w8.hip).
#pragma once
#include "kx_common.hpp"

namespace kx {

// first maximum wins (cuda/argmax_kernel.cu:49-71, argmax_sampler.cpp:6-8)
__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}

// pair slots of the step workspace: the stand-alone kernel uses kStepBlocks of them, the classifier
// GEMV one per workgroup
constexpr int kStepBlocks = 128;
constexpr int kStepSlots = 2048;
// workspace: kStepSlots values, kStepSlots indices, then the arrival ticket in a 256-byte block of
// its own
constexpr size_t kStepTicketOff = (size_t)kStepSlots * 8;
constexpr size_t kStepWorkspaceBytes = kStepTicketOff + 256;

__device__ __forceinline__ void block_argmax(float& best, int& bi, float* sv, int* si) {
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    argmax_merge(best, bi, ov, oi);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) {
    sv[wave] = best;
    si[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    for (int w = 1; w < nw; ++w) argmax_merge(best, bi, sv[w], si[w]);
}

// One launch for both stages.  Every workgroup publishes its pair with write-through stores and
// draws an arrival ticket (the counter wraps back to zero on the last arrival, so a workspace
// that was zero before its first use stays usable from launch to launch); the workgroup that
// draws the last ticket reads all pairs with device-scope loads and finishes the step.  With a
// table it also leaves the embedding row of the token at the NEW position (the one just picked,
// or, when the history is not written, the prompt token already stored there) in x_out, which
// is the input of the next step.
struct StepArgs {
  const float* logits; int n;
  float* pv; int32_t* pi; unsigned* ticket;
  int index_base; float* cand;                    // candidate form: {value, global index} and no bookkeeping
  int32_t* token; int32_t* pos; int32_t* history; int history_len; int write_history;
  const float* table; int dim; int vocab; float* x_out;
  KX_BFIELD(10)  // operands of the bounds-checking build, in the order of StepOperand
};
enum StepOperand { SB_LOGITS, SB_PV, SB_PI, SB_TICKET, SB_CAND, SB_TOKEN, SB_POS, SB_HISTORY, SB_TABLE, SB_XOUT };

__device__ __forceinline__ void step_bookkeeping(const StepArgs& a, int bi, int* sm_tok) {
  // thread 0 only
  if (bi == 0x7fffffff) bi = 0;
  const int p = *KX_CK(a, SB_POS, a.pos);
  if (a.history && a.write_history && p + 1 < a.history_len) *KX_CK(a, SB_HISTORY, a.history + (p + 1)) = bi;
  *KX_CK(a, SB_TOKEN, a.token) = bi;
  *a.pos = p + 1;
  int nxt = -1;
  if (a.table && a.x_out && a.history && p + 1 < a.history_len)
    nxt = a.write_history ? bi : *KX_CK(a, SB_HISTORY, a.history + (p + 1));
  *sm_tok = nxt;
}

__device__ __forceinline__ void step_embed(const StepArgs& a, int tok) {
  // whole workgroup; tok < 0: nothing to do
  if (tok < 0) return;
  if (tok >= a.vocab) tok = 0;
  const float* src = a.table + (size_t)tok * a.dim;
  if (a.dim % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)a.x_out % 16 == 0)) {
    for (int i = threadIdx.x; i < a.dim / 4; i += blockDim.x)
      *KX_CK(a, SB_XOUT, reinterpret_cast<f32x4*>(a.x_out) + i) = *KX_CK(a, SB_TABLE, reinterpret_cast<const f32x4*>(src) + i);
  } else {
    for (int i = threadIdx.x; i < a.dim; i += blockDim.x) *KX_CK(a, SB_XOUT, a.x_out + i) = *KX_CK(a, SB_TABLE, src + i);
  }
}

}  // namespace kx
