// LDS read rate per CU for ds_read_b128, two ds_read_b64 and ds_read2_b64 (16 bytes per lane each
// way), lane-linear addresses, 4 and 8 waves per CU.  Prints bytes per cycle per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0;
  const unsigned a16 = (unsigned)(size_t)(wave * 4096 + lane * 16);  // LDS byte address, 16 per lane
  const unsigned a8 = (unsigned)(size_t)(wave * 4096 + lane * 8);    // 8 per lane
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u32x4 v0, v1, v2, v3;
    u32x2 p0, p1, p2, p3, p4, p5, p6, p7;
    if (MODE == 0) {
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                   "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a16));
      acc += v0.x ^ v1.y ^ v2.z ^ v3.w;
    } else if (MODE == 1) {
      asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:1024\n\t"
                   "ds_read_b64 %3, %8 offset:1032\n\tds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2056\n\t"
                   "ds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3080\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7) : "v"(a16));
      acc += p0.x ^ p1.y ^ p2.x ^ p3.y ^ p4.x ^ p5.y ^ p6.x ^ p7.y;
    } else if (MODE == 2) {
      asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\t"
                   "ds_read_b64 %3, %8 offset:1536\n\tds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\t"
                   "ds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7) : "v"(a8));
      acc += p0.x ^ p1.y ^ p2.x ^ p3.y ^ p4.x ^ p5.y ^ p6.x ^ p7.y;
    } else {
      // ds_read2_b64: two 8-byte reads per instruction at offsets in units of 8 bytes
      asm volatile("ds_read2_b64 %0, %4 offset0:0 offset1:64\n\tds_read2_b64 %1, %4 offset0:128 offset1:192\n\t"
                   "ds_read2st64_b64 %2, %4 offset0:4 offset1:5\n\tds_read2st64_b64 %3, %4 offset0:6 offset1:7\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a8));
      acc += v0.x ^ v1.y ^ v2.z ^ v3.w;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run(const char* name, int threads) {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  k<MODE><<<256, threads>>>(out, cyc, 10);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  const double bytes = (double)iters * 4 * 1024 * (threads / 64);
  printf("%-14s %d waves: %.1f us, %.1f bytes per shader cycle per CU (s_memtime ticks %lld)\n", name, threads / 64,
         ms * 1e3, bytes / (double)h[0], h[0]);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int threads : {256, 512}) {
    run<0>("b128", threads);
    run<1>("2 x b64 (+8)", threads);
    run<2>("2 x b64 linear", threads);
    run<3>("read2_b64", threads);
  }
  return 0;
}
