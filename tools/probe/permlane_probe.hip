// Prints what xor16_sum / xor32_sum / wave_sum return for v = 1 << (lane / 16) style inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../llm-inference-engine_amd/csrc/kx_common.hpp"
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  const float v = (float)(1 << (lane >> 4)) + 0.001f * (lane & 15);
  out[lane] = kx::xor16_sum(v);
  out[64 + lane] = kx::xor32_sum(v);
  out[128 + lane] = kx::wave_sum((float)lane);
  out[192 + lane] = kx::lane_xor16(v) + v;
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) { for (int i = 0; i < 64; i += 5) printf("%8.3f ", h[r * 64 + i]); printf("\n"); }
  return 0;
}
