#include "gemm.cuh"
#include "gemm_api.h"
namespace cc {
int gemm_dact(int al, int bl, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, bf16_t* C, int ldc,
              const bf16_t* aux, int act, hipStream_t st) {
    if ((ldc & 7) || (N & 7)) return CC_ERR_SHAPE;
    EpiDAct e{C, aux, ldc, M, N, act};
    return launch_gemm(al, bl, A, lda, B, ldb, M, N, K, 1, e, st);
}
}  // namespace cc
