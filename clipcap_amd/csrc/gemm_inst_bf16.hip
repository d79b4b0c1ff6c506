#include "gemm.cuh"
#include "gemm_api.h"
namespace cc {
int gemm_bf16out(int al, int bl, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, bf16_t* C, int ldc,
                 const float* bias, int act, bf16_t* pre, hipStream_t st) {
    if ((ldc & 7) || (N & 7)) return CC_ERR_SHAPE;
    EpiBF16 e{C, pre, bias, ldc, M, N, act};
    return launch_gemm(al, bl, A, lda, B, ldb, M, N, K, 1, e, st);
}
}  // namespace cc
