#include "gemm.hip.h"
#include "gemm_api.h"
namespace CC_NS {
int gemm_bf16out(int al, int bl, const act_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, act_t* C, int ldc,
                 const float* bias, int act, act_t* pre, hipStream_t st) {
    cc_shared::ProfScope _all(cc_shared::SITE_ALL_GEMMS, st, 2.0 * M * N * (double)K);
    if ((ldc & 7) || (N & 7)) return CC_ERR_SHAPE;
    const op16_t* A16;
    CC_X3_NT(A, lda, ldb, M, K, A16, al, bl, st);
    EpiBF16 e{C, pre, bias, ldc, M, N, act};
#if CC_OP == 2
    e.img = x3_take_emit(C);
#endif
    static const bool nt = []() { const char* v = cc_lab_env("CC_PRE_NT"); return v && atoi(v) != 0; }();      // experiment switch
    e.pre_nt = nt;
    // plain launches (no activation, no pre-activation copy, no operand image): the functor without run-time switches
    static const bool plain_on = []() { const char* v = cc_lab_env("CC_EPI_PLAIN"); return !v || atoi(v) != 0; }();
    if (plain_on && act == 0 && !pre && e.img == 0) {
        EpiBF16Plain p{C, bias, ldc, M, N};
        return launch_gemm(al, bl, A16, lda, B, ldb, M, N, K, 1, p, st);
    }
    static const bool spec_on = []() { const char* v = cc_lab_env("CC_EPI_SPEC"); return !v || atoi(v) != 0; }();
    if (spec_on && act == 3 && pre && !nt && e.img == 0) {       // c_fc forward of the training step: gelu_new, gelu' stored beside
        EpiBF16T<3, 1> g3{C, pre, bias, ldc, M, N, 3};
        return launch_gemm(al, bl, A16, lda, B, ldb, M, N, K, 1, g3, st);
    }
    return launch_gemm(al, bl, A16, lda, B, ldb, M, N, K, 1, e, st);
}
}  // namespace CC_NS
