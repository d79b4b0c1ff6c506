// Where does a K-step of the 256-row staggered GEMM kernel go?  Includes the product header with CC_STAMP (cycle stamps around each
// phase of the main loop, wave 0 of either group of one block).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCC_STAMP -I../../clipcap_amd/csrc -o stag256_timing stag256_timing.hip
#include "gemm.hip.h"
#include <cstdio>
#include <vector>
using namespace CC_NS;
namespace cc { int g_gemm_tile_mode = -1, g_gemm_s64 = -1, g_gemm_small_x2 = 1; }
template <int NJ, int NI = 8>
static void run(int M, int N, int K) {
    op16_t *A, *B; float* C;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0, (size_t)M * K * 2); hipMemset(B, 0, (size_t)N * K * 2);
    GemmShape g;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.k_chunk = K; g.group_m = 8;
    EpiF32 e{C, nullptr, N, M, N, 0, 1.0f};
    constexpr size_t sh = (size_t)H_NS * (32 * NI + H_BN) * H_BK * 2;
    hipFuncSetAttribute((const void*)gemm_nt_stag256_kernel<EpiF32, NJ, false, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    const dim3 gr((unsigned)(((M + 32 * NI - 1) / (32 * NI)) * ((N + 64 * NJ - 1) / (64 * NJ))));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL((gemm_nt_stag256_kernel<EpiF32, NJ, false, NI>), gr, dim3(512), sh, 0, A, B, g, e);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int it = 0; it < 10; it++) hipLaunchKernelGGL((gemm_nt_stag256_kernel<EpiF32, NJ, false, NI>), gr, dim3(512), sh, 0, A, B, g, e);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long h[16];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(cc_stamp_buf), sizeof(h));
    const int nk = K / 32;
    const char* names[8] = {"prologue+tail", "dma issue", "frag reads+wait", "barrier after reads", "mfma issue", "vmcnt wait", "barrier after mfma", "epilogue"};
    printf("M=%d N=%d K=%d tile %dx%d: %.1f us, %.0f TFLOP/s (with stamps); cycles per K-step, group 0 | group 1\n", M, N, K, 32 * NI, 64 * NJ, ms * 100,
           2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
    double t0 = 0, t1 = 0;
    for (int i = 1; i <= 6; i++) {
        printf("  %-20s %8.1f %8.1f\n", names[i], (double)h[i] / nk, (double)h[8 + i] / nk);
        t0 += (double)h[i] / nk; t1 += (double)h[8 + i] / nk;
    }
    printf("  %-20s %8.1f %8.1f\n  %-20s %8.0f %8.0f   epilogue %8.0f %8.0f\n", "K-step total", t0, t1, "prologue+tail (abs)", (double)h[0], (double)h[8], (double)h[7],
           (double)h[15]);
    hipFree(A); hipFree(B); hipFree(C);
}
int main() {
    run<4>(8192, 8192, 8192);
    run<4>(10240, 50304, 768);
    run<4, 10>(10240, 50304, 768);
    run<4, 10>(12800, 3072, 768);
    run<3>(12800, 768, 3072);
    // epilogue cost against the number of CUs storing at once (K = 768: 24 K-steps per tile)
    run<4>(4096, 4096, 768);
    run<4>(2048, 2048, 768);
    run<4>(512, 512, 768);
    return 0;
}
