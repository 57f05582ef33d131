// Shared device helpers for the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <initializer_list>
#include <mutex>

#include "../../include/kx_hip.h"

// fp32 -> fp16 with saturation: a value past the fp16 range becomes +-65504 instead of an
// infinity that would turn into NaN in the next product (one v_med3_f32 ahead of the conversion).
__device__ __forceinline__ _Float16 sat_f16(float x) {
  return (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
}

// A kernel's raised dynamic-LDS limit (hipFuncSetAttribute) holds for the device that was current
// when it was set: one flag per device and call site.  raise() sets the attribute the first time a
// device comes by and returns the HIP status; the flag goes up only after the call succeeded, under
// a lock, so neither a second host thread (ranks of one process) nor a retry after a failure can
// launch ahead of the attribute.
struct OncePerDevice {
  std::atomic<bool> done[64] = {};
  std::mutex mu;
  int raise(const void* kernel, int lds_bytes) {
    int d = 0;
    const bool known = hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64;
    if (known && done[d].load(std::memory_order_acquire)) return 0;
    std::lock_guard<std::mutex> lock(mu);
    if (known && done[d].load(std::memory_order_relaxed)) return 0;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    if (known) done[d].store(true, std::memory_order_release);
    return 0;
  }
};

#define KX_HIP_TRY(expr)                        \
  do {                                          \
    hipError_t e_ = (expr);                     \
    if (e_ != hipSuccess) return (int)e_;       \
  } while (0)

// A kernel launch reports configuration errors through hipGetLastError.
#define KX_LAUNCH_CHECK()                       \
  do {                                          \
    hipError_t e_ = hipGetLastError();          \
    if (e_ != hipSuccess) return (int)e_;       \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Bounds-checking build (make BOUNDS=1, -DKX_BOUNDS).  A read a few bytes past an array cannot be
// seen by a parity test (nothing uses the value) and faults the device only when the array ends on
// the last mapped page.  In this build every kernel's argument block carries, per operand, the
// byte range of the ALLOCATION the operand points into (resolved on the host, at launch, from the
// library's own table of kx_malloc blocks), and every global load / store / LDS-DMA address is
// compared with its operand's range first.  A violation never traps: the access is redirected to
// the first byte of the range and {site, operand, offset, extent} goes into a device record that
// kx_bounds_report() reads back.  In the normal build every macro below expands to its pointer
// argument and the argument blocks are unchanged.
//   KX_BFIELD(N)              member of an argument struct: N operand ranges
//   KX_BSET(a, i, ptr)        host, before the launch: operand i of `a` is the block holding ptr
//   KX_CK(a, i, ptr)          device: ptr (of operand i), checked for sizeof(*ptr) bytes
//   KX_CKN(a, i, ptr, bytes)  the same for an access of `bytes` bytes
//   KX_TOUCH(a, i, ptr, bytes) check and record only (buffer instructions, whose descriptor already
//                             keeps the hardware inside its range)
// Kernels that take plain pointer parameters instead of a struct: KX_BPARAM(N) as the last
// parameter, KX_BPASS(N, ...pointers) as the last launch argument, KX_CKP / KX_CKPN in the body
// (KX_BFWD hands the ranges on to a device function declared with KX_BPARAM).
#ifndef KX_FILE_ID
#define KX_FILE_ID 0
#endif
#define KX_SITE ((unsigned)((KX_FILE_ID << 16) | __LINE__))
#ifdef KX_BOUNDS
namespace kx {
struct BRange { const char* lo; const char* hi; };  // hi == nullptr: memory the table does not know
struct BoundsEntry { unsigned site, operand; long long offset, extent; };
struct BoundsLog { unsigned long long total; BoundsEntry e[1024]; };
template <int N>
struct BArgs { BRange r[N]; BoundsLog* log; };

// host side (runtime.hip)
BRange bounds_lookup(const void* p);
BoundsLog* bounds_log();
void bounds_track(const void* p, size_t bytes);
void bounds_untrack(const void* p);
template <int N>
inline void bounds_set(BArgs<N>& b, int i, const void* p) {
  b.r[i] = p ? bounds_lookup(p) : BRange{nullptr, nullptr};
  b.log = bounds_log();
}
template <int N>
inline BArgs<N> bounds_make(std::initializer_list<const void*> ps) {
  BArgs<N> b{};
  int i = 0;
  for (const void* p : ps) bounds_set(b, i++, p);
  b.log = bounds_log();
  return b;
}

__device__ __noinline__ static void bounds_note(BoundsLog* log, unsigned site, unsigned operand,
                                                long long offset, long long extent) {
  if (!log) return;
  const unsigned long long n = atomicAdd(&log->total, 1ull);
  if (n < 1024) log->e[n] = BoundsEntry{site, operand, offset, extent};
}
template <int N, class T>
__device__ __forceinline__ T* bounds_ck(const BArgs<N>& b, int i, unsigned site, T* p, size_t bytes) {
  const BRange r = b.r[i];
  const char* c = reinterpret_cast<const char*>(p);
  if (r.hi && (c < r.lo || c + bytes > r.hi)) {
    bounds_note(b.log, site, (unsigned)i, (long long)(c - r.lo), (long long)(r.hi - r.lo));
    return reinterpret_cast<T*>(const_cast<char*>(r.lo));
  }
  return p;
}
}  // namespace kx
#define KX_BFIELD(N) kx::BArgs<N> kxb;
#define KX_BSET(a, i, ptr) kx::bounds_set((a).kxb, (i), (ptr))
#define KX_CK(a, i, ptr) kx::bounds_ck((a).kxb, (i), KX_SITE, (ptr), sizeof(*(ptr)))
#define KX_CKN(a, i, ptr, bytes) kx::bounds_ck((a).kxb, (i), KX_SITE, (ptr), (size_t)(bytes))
#define KX_TOUCH(a, i, ptr, bytes) ((void)kx::bounds_ck((a).kxb, (i), KX_SITE, (ptr), (size_t)(bytes)))
#define KX_BPARAM(N) , const kx::BArgs<N> kxb
#define KX_BPASS(N, ...) , kx::bounds_make<N>({__VA_ARGS__})
#define KX_BFWD , kxb
#define KX_TOUCHP(i, ptr, bytes) ((void)kx::bounds_ck(kxb, (i), KX_SITE, (ptr), (size_t)(bytes)))
#define KX_CKP(i, ptr) kx::bounds_ck(kxb, (i), KX_SITE, (ptr), sizeof(*(ptr)))
#define KX_CKPN(i, ptr, bytes) kx::bounds_ck(kxb, (i), KX_SITE, (ptr), (size_t)(bytes))
#else
#define KX_BFIELD(N)
#define KX_BSET(a, i, ptr) ((void)0)
#define KX_CK(a, i, ptr) (ptr)
#define KX_CKN(a, i, ptr, bytes) (ptr)
#define KX_TOUCH(a, i, ptr, bytes) ((void)0)
#define KX_BPARAM(N)
#define KX_BPASS(N, ...)
#define KX_BFWD
#define KX_TOUCHP(i, ptr, bytes) ((void)0)
#define KX_CKP(i, ptr) (ptr)
#define KX_CKPN(i, ptr, bytes) (ptr)
#endif

namespace kx {

constexpr int kWave = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

// ---- DPP cross-lane moves inside a 16-lane row ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;  // every lane of the 16-lane row holds the row's sum
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}

// xor-16 / xor-32 lane exchange (across DPP rows)
__device__ __forceinline__ float lane_xor16(float v) {
  // ds_swizzle bit-mode: and_mask=0x1F, or_mask=0, xor_mask=0x10 -> within each 32-lane half
  return __builtin_bit_cast(
      float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (0x10 << 10) | 0x1F));
}
__device__ __forceinline__ float lane_xor32(float v) {
  int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  return __builtin_bit_cast(
      float, __builtin_amdgcn_ds_bpermute(((lane ^ 32) << 2), __builtin_bit_cast(int, v)));
}

// v_permlane16_swap / v_permlane32_swap exchange the odd 16-lane rows (upper 32 lanes) of the first
// register with the even rows (lower 32 lanes) of the second.  Called with two copies of one value,
// the first register ends up holding the even rows' (lower half's) values in both positions and the
// second the odd rows' (upper half's): "mine" and "my partner's" in every lane, on the vector ALU,
// where ds_swizzle / ds_bpermute above take a trip through the LDS pipeline and an lgkmcnt wait.
// Written as inline assembly because this toolchain's __builtin_amdgcn_permlane{16,32}_swap returns
// its first result in both elements of the pair.  The s_nop covers the wait states the swap needs
// after a vector-ALU write of either register.
__device__ __forceinline__ void permlane16_swap(unsigned& a, unsigned& b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane32_swap(unsigned& a, unsigned& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// Sum / maximum of a value with its partner 16 or 32 lanes away, in every lane.  Addition and
// maximum are commutative, so the results equal those of v + lane_xor16(v) etc. bit for bit.
__device__ __forceinline__ float xor16_sum(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane16_swap(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float xor32_sum(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane32_swap(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float xor16_max(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane16_swap(a, b);
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ float xor32_max(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane32_swap(a, b);
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}

// Full 64-lane sum; every lane receives it.
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v = xor16_sum(v);
  v = xor32_sum(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = xor16_max(v);
  v = xor32_max(v);
  return v;
}

__device__ __forceinline__ int lane_id() {
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

}  // namespace kx
