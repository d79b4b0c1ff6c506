#include "gemm.hip.h"
#include "gemm_api.h"
namespace CC_NS {
int gemm_lmhead(const op16_t* A, int lda, const op16_t* B, int ldb, int M, int Vp, int V, int K, op16_t* C, int ldc, float* pmax,
                float* psum, int npart, const int* target, float* tgt_logit, hipStream_t st) {
    cc_shared::ProfScope _all(cc_shared::SITE_ALL_GEMMS, st, 2.0 * M * V * (double)K);
    if ((ldc & 63) || ldc < Vp || npart * 64 < Vp) return CC_ERR_SHAPE;
    EpiLMHead e{C, pmax, psum, target, tgt_logit, ldc, M, V, npart};
    return launch_gemm(0, 0, A, lda, B, ldb, M, Vp, K, 1, e, st);
}
}  // namespace CC_NS
