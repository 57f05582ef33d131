// Probe: a chain of dependent int8 GEMV launches (8192 x 8192, six weight buffers in rotation, HIP
// graph).  Does it pay to let every wave, as its last act, touch the bytes that the same wave of the
// NEXT launch will ask for first (its first row of the next matrix), so that launch starts on L2 hits
// instead of HBM misses?  Modes: plain chain; touch and leave without waiting; touch and wait.
// Standalone: hipcc --offload-arch=gfx950 -O3 tools/probe/tail_prefetch_probe.hip -o tail_prefetch_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kThreads = 512;

__device__ __forceinline__ float dot16(v4i w, const float* xs) {
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = w[j];
    acc += (float)(int8_t)(v & 0xff) * xs[4 * j] + (float)(int8_t)((v >> 8) & 0xff) * xs[4 * j + 1] +
           (float)(int8_t)((v >> 16) & 0xff) * xs[4 * j + 2] + (float)(int8_t)(v >> 24) * xs[4 * j + 3];
  }
  return acc;
}

// MODE 0: plain.  1: touch the next matrix's first row of this wave, do not wait.  2: touch and wait.
template <int MODE>
__global__ __launch_bounds__(kThreads) void probe_gemv(const int8_t* __restrict__ W, const int8_t* __restrict__ Wnext,
                                                       const float* xin, float* yout, int K) {
  constexpr int M = 8192;
  __shared__ float xs[M];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int waves_total = gridDim.x * (kThreads / 64);
  const int gw = blockIdx.x * (kThreads / 64) + wave;
  const int rows_per = (K + waves_total - 1) / waves_total;
  int row = gw * rows_per;
  const int row0 = row;
  const int row_end = min(K, row + rows_per);
  v4i w[8];
  if (row < row_end) {
    const v4i* wp = reinterpret_cast<const v4i*>(W + (size_t)row * M) + lane;
#pragma unroll
    for (int c = 0; c < 8; ++c) w[c] = __builtin_nontemporal_load(wp + 64 * c);
  }
#pragma unroll
  for (int i = 0; i < M / kThreads; ++i) xs[threadIdx.x + i * kThreads] = xin[threadIdx.x + i * kThreads];
  __syncthreads();
  while (row < row_end) {
    v4i cur[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) cur[c] = w[c];
    if (row + 1 < row_end) {
      const v4i* wp = reinterpret_cast<const v4i*>(W + (size_t)(row + 1) * M) + lane;
#pragma unroll
      for (int c = 0; c < 8; ++c) w[c] = __builtin_nontemporal_load(wp + 64 * c);
    }
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc += dot16(cur[c], xs + c * 1024 + lane * 16);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    acc *= 1.0f / 8192.f;  // keep the chain bounded
    if (lane == 0) yout[row] = acc;
    ++row;
  }
  if (MODE != 0 && row0 < row_end) {
    // one dword per 128-byte line of the wave's first row of the next matrix: the lines land in
    // this XCD's L2, which is also the XCD of the same workgroup index of the next launch
    const int* p = reinterpret_cast<const int*>(Wnext + (size_t)row0 * M) + lane * 32;
    int sink0;
    asm volatile("global_load_dword %0, %1, off" : "=v"(sink0) : "v"(p) : "memory");
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

int main(int argc, char** argv) {
  const int K = 8192, M = 8192, NB = 6, CHAIN = 60;
  const int grid = argc > 1 ? atoi(argv[1]) : 256;
  std::vector<int8_t*> W(NB);
  std::vector<int8_t> h((size_t)K * M);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (int8_t)((i * 2654435761u >> 13) & 0xff);
  for (int b = 0; b < NB; ++b) { CK(hipMalloc(&W[b], h.size())); CK(hipMemcpy(W[b], h.data(), h.size(), hipMemcpyHostToDevice)); }
  float* buf[3];
  for (auto& p : buf) { CK(hipMalloc(&p, M * 4)); CK(hipMemset(p, 0, M * 4)); }
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int round = 0; round < 2; ++round)
  for (int mode = 0; mode < 3; ++mode) {
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < CHAIN; ++i) {
      const int8_t* w = W[i % NB]; const int8_t* wn = W[(i + 1) % NB];
      float* x = buf[i % 3]; float* y = buf[(i + 1) % 3];
      if (mode == 0) hipLaunchKernelGGL(probe_gemv<0>, dim3(grid), dim3(kThreads), 0, s, w, wn, x, y, K);
      else if (mode == 1) hipLaunchKernelGGL(probe_gemv<1>, dim3(grid), dim3(kThreads), 0, s, w, wn, x, y, K);
      else hipLaunchKernelGGL(probe_gemv<2>, dim3(grid), dim3(kThreads), 0, s, w, wn, x, y, K);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const int reps = 20;
    float total = 0.f, best = 1e9f;
    for (int r = 0; r < reps + 3; ++r) {
      CK(hipEventRecord(t0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(t1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (r >= 3) { total += ms; best = ms < best ? ms : best; }
    }
    const double us = total / reps * 1000.0 / CHAIN;
    printf("grid %d %-28s avg %.2f us/launch (best %.2f)  %.0f GB/s\n", grid,
           mode == 0 ? "plain chain" : mode == 1 ? "tail touch, no wait" : "tail touch, waited", us,
           best * 1000.0 / CHAIN, (double)K * M / us * 1e-3);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
