#include "gemm.hip.h"
#include "gemm_api.h"
namespace CC_NS {
int gemm_lmhead(const act_t* A, int lda, const op16_t* B, int ldb, int M, int Vp, int V, int K, act_t* C, int ldc, float* pmax,
                float* psum, int npart, const int* target, float* tgt_logit, hipStream_t st, const float* cref) {
    cc_shared::ProfScope _all(cc_shared::SITE_ALL_GEMMS, st, 2.0 * M * V * (double)K);
    if ((ldc & 63) || ldc < Vp || npart * 64 < Vp) return CC_ERR_SHAPE;
    const op16_t* A16;
    CC_X3_NT(A, lda, ldb, M, K, A16, 0, 0, st);
    if (cref) {      // exponential form: the caller takes the target logit from cref itself (tgt_logit unused)
        EpiLMHeadExp e{C, pmax, psum, cref, ldc, M, V, npart};
#if CC_OP == 2
        e.img = x3_take_emit(C);      // frozen-LM runs: E goes out as the input-gradient GEMM's operand image
#endif
        return launch_gemm(0, 0, A16, lda, B, ldb, M, Vp, K, 1, e, st);
    }
    EpiLMHead e{C, pmax, psum, target, tgt_logit, ldc, M, V, npart};
    return launch_gemm(0, 0, A16, lda, B, ldb, M, Vp, K, 1, e, st);
}
int gemm_logits_part(const act_t* A, int lda, const op16_t* B, int ldb, int M, int Ns, int V, int K, float* C, int ldc, float* pmax, float* psum,
                     int npart, hipStream_t st) {
    cc_shared::ProfScope _all(cc_shared::SITE_ALL_GEMMS, st, 2.0 * M * V * (double)K);
    if ((ldc & 3) || (Ns & 7) || npart * 64 < Ns) return CC_ERR_SHAPE;
    const op16_t* A16;
    CC_X3_NT(A, lda, ldb, M, K, A16, 0, 0, st);
    EpiLogits e{C, pmax, psum, ldc, M, Ns, V, npart};
    return launch_gemm(0, 0, A16, lda, B, ldb, M, Ns, K, 1, e, st);
}
}  // namespace CC_NS
