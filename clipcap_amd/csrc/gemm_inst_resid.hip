#include "gemm.cuh"
#include "gemm_api.h"
namespace cc {
int gemm_resid(int al, int bl, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, float* out, const float* res,
               int ld, const float* bias, hipStream_t st, Drop drop) {
    if ((ld & 7) || (N & 7)) return CC_ERR_SHAPE;
    EpiResid e{out, res, bias, ld, M, N, drop};
    return launch_gemm(al, bl, A, lda, B, ldb, M, N, K, 1, e, st);
}
}  // namespace cc
