#include "gemm.hip.h"
#include "gemm_api.h"
namespace CC_NS {
int gemm_dact(int al, int bl, const act_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, act_t* C, int ldc,
              const act_t* aux, int act, hipStream_t st) {
    cc_shared::ProfScope _all(cc_shared::SITE_ALL_GEMMS, st, 2.0 * M * N * (double)K);
    if ((ldc & 7) || (N & 7)) return CC_ERR_SHAPE;
    const op16_t* A16;
    CC_X3_NT(A, lda, ldb, M, K, A16, al, bl, st);
    EpiDAct e{C, aux, ldc, M, N, act};
#if CC_OP == 2
    e.img = x3_take_emit(C);
#endif
    static const bool spec_on = []() { const char* v = cc_lab_env("CC_EPI_SPEC"); return !v || atoi(v) != 0; }();
    if (spec_on && act == 3 && e.img == 0) {       // the forward stored gelu': one multiply, no switch per unit
        EpiDActT<3> m{C, aux, ldc, M, N, 3};
        return launch_gemm(al, bl, A16, lda, B, ldb, M, N, K, 1, m, st);
    }
    return launch_gemm(al, bl, A16, lda, B, ldb, M, N, K, 1, e, st);
}
}  // namespace CC_NS
