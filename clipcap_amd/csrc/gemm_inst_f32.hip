#include "gemm.cuh"
#include "gemm_api.h"
namespace cc {
int gemm_f32out(int al, int bl, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, float* C, int ldc,
                const float* bias, int mode, float alpha, int ksplit, hipStream_t st) {
    if ((ldc & 3) || (N & 7)) return CC_ERR_SHAPE;
    if (ksplit > 1 && mode != 2) return CC_ERR_ARG;
    EpiF32 e{C, bias, ldc, M, N, mode, alpha};
    return launch_gemm(al, bl, A, lda, B, ldb, M, N, K, ksplit, e, st);
}
int gemm_wgrad(const bf16_t* X, int ldx, const bf16_t* Y, int ldy, int Mw, int Nw, int K, float* dW, int ldw, hipStream_t st) {
    const int tiles = ((Mw + G_BM - 1) / G_BM) * ((Nw + G_BN - 1) / G_BN);
    int ks = 512 / (tiles > 0 ? tiles : 1);
    const int kmax = (K + 255) / 256;  // at least 4 K-steps per slice
    if (ks > kmax) ks = kmax;
    if (ks < 1) ks = 1;
    return gemm_f32out(1, 1, X, ldx, Y, ldy, Mw, Nw, K, dW, ldw, nullptr, 2, 1.0f, ks, st);
}
}  // namespace cc
