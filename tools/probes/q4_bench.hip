// The persistent 4-wave NT kernel (gemm_q4.hip.h: 64-deep full-line stages, instruction-level K loop) against the 8-wave staggered kernel
// and, with -DWITH_VENDOR -lhipblaslt, hipBLASLt's bf16 -> bf16 GEMM — same buffers, same process, HIP events around 20 launches.
// Bit equality of the outputs is checked (both kernels accumulate one MFMA per 32-deep K chunk in ascending order).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../clipcap_amd/csrc [-DWITH_VENDOR] -o q4_bench q4_bench.hip [-lhipblaslt]
#define CC_Q4_VARIANTS
#include "gemm.hip.h"
#include <cstdio>
#include <cstring>
#include <vector>
#ifdef WITH_VENDOR      // measuring stick only: nothing in the product calls a vendor GEMM
#include <hipblaslt/hipblaslt.h>
#endif
using namespace CC_NS;
namespace cc_shared { int g_gemm_tile_mode = -1, g_gemm_s64 = -1, g_gemm_small_x2 = 1, g_decode_last_path = 0; }

template <class F>
static float time_us(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms * 1000.f / reps;
}
static __global__ void k_fill(op16_t* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = f2op(((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f));      // uniform [-1, 1)
    }
}
static __global__ void k_diff(const unsigned short* a, const unsigned short* b, size_t n, unsigned long long* out) {
    unsigned long long d = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d += a[i] != b[i];
    if (d) atomicAdd(out, d);
}
// EpiBF16 with its run-time switches as template parameters
template <int ACT, bool PRE>
struct EpiBF16S : EpiBF16 {
    __device__ __forceinline__ void fin(int row, int col, float (&v)[8], const float (&b)[8]) const {
        if (row >= M || col >= Ns) return;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += b[e];
        if constexpr (ACT == 3) {
            float dg[8];
#pragma unroll
            for (int e = 0; e < 8; e++) gelu_new_both(v[e], v[e], dg[e]);
            if constexpr (PRE) act_st8_nt(pre + (size_t)row * ldc + col, dg);
        } else {
            if constexpr (PRE) act_st8(pre + (size_t)row * ldc + col, v);
        }
        act_st8(C + (size_t)row * ldc + col, v);
    }
};
template <int NI, int NS, int VAR = 0, class E = EpiBF16>
static void launch_q4(const op16_t* A, const op16_t* B, const GemmShape& g, const E& e, int grid) {
    constexpr size_t sh = (size_t)NS * (32 * NI + 256) * 128;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)gemm_nt_q4_kernel<E, NI, NS, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
    const int tiles = ((g.M + 32 * NI - 1) / (32 * NI)) * ((g.N + 255) / 256);
    hipLaunchKernelGGL((gemm_nt_q4_kernel<E, NI, NS, VAR>), dim3(tiles < grid ? tiles : grid), dim3(256), sh, 0, A, B, g, e);
}
template <int NI, int NJ = 4, class E = EpiBF16>
static void launch_stag(const op16_t* A, const op16_t* B, const GemmShape& g, const E& e) {
    constexpr size_t sh = (size_t)H_NS * (32 * NI + H_BN) * H_BK * 2;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)gemm_nt_stag256_kernel<E, NJ, false, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
    const dim3 gr((unsigned)(((g.M + 32 * NI - 1) / (32 * NI)) * ((g.N + 64 * NJ - 1) / (64 * NJ))));
    hipLaunchKernelGGL((gemm_nt_stag256_kernel<E, NJ, false, NI>), gr, dim3(512), sh, 0, A, B, g, e);
}
#ifdef WITH_VENDOR
// C[M][N] (row-major bf16) = A[M][K] . B[N][K]^T  ==  column-major C^T[N][M] = op_T(B as K x N) . (A as K x M)
static float vendor_us(const op16_t* A, const op16_t* B, act_t* C, int M, int N, int K, int reps) {
    static hipblasLtHandle_t h = nullptr;
    static void* ws = nullptr;
    const size_t wsz = 128u << 20;
    if (!h) { hipblasLtCreate(&h); hipMalloc(&ws, wsz); }
    hipblasLtMatmulDesc_t d; hipblasLtMatrixLayout_t la, lb, lc;
    hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F);
    hipblasOperation_t T = HIPBLAS_OP_T, Nn = HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &T, sizeof(T));
    hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &Nn, sizeof(Nn));
    hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, N, K);
    hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, K, M, K);
    hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, N, M, N);
    hipblasLtMatmulPreference_t pref; hipblasLtMatmulPreferenceCreate(&pref);
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz));
    hipblasLtMatmulHeuristicResult_t res[1]; int got = 0;
    hipblasLtMatmulAlgoGetHeuristic(h, d, la, lb, lc, lc, pref, 1, res, &got);
    float us = -1.f;
    if (got > 0) {
        const float one = 1.f, zero = 0.f;
        auto go = [&] { hipblasLtMatmul(h, d, &one, B, la, A, lb, &zero, C, lc, C, lc, &res[0].algo, ws, wsz, 0); };
        for (int i = 0; i < 3; i++) go();
        us = time_us(go, reps); us = time_us(go, reps);
    }
    hipblasLtMatmulPreferenceDestroy(pref); hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(lb); hipblasLtMatrixLayoutDestroy(lc); hipblasLtMatmulDescDestroy(d);
    return us;
}
#endif

static void run(const char* name, int M, int N, int K, int grid) {
    op16_t *A, *B; act_t *C0, *C1; unsigned long long* dcount;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2);
    hipMalloc(&C0, (size_t)M * N * 2); hipMalloc(&C1, (size_t)M * N * 2); hipMalloc(&dcount, 8);
    k_fill<<<1024, 256>>>(A, (size_t)M * K, 17u);
    k_fill<<<1024, 256>>>(B, (size_t)N * K, 91u);
    GemmShape g;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.k_chunk = K; g.group_m = 8; g.stagger = 0;
    EpiBF16 e0{C0, nullptr, nullptr, N, M, N, 0};
    EpiBF16 e1{C1, nullptr, nullptr, N, M, N, 0};
    const double fl = 2.0 * M * N * K;
    printf("%-14s M=%5d N=%5d K=%5d  %6.2f GFLOP\n", name, M, N, K, fl / 1e9);
    launch_stag<8>(A, B, g, e0);        // reference output
    for (int warm = 0; warm < 20; warm++) launch_stag<8>(A, B, g, e0);
    hipDeviceSynchronize();
    auto check = [&](const char* what, auto launch) {
        hipMemset(C1, 0x7f, (size_t)M * N * 2); hipMemset(dcount, 0, 8);
        launch();
        k_diff<<<1024, 256>>>(C0, C1, (size_t)M * N, dcount);
        unsigned long long d = 0;
        hipMemcpy(&d, dcount, 8, hipMemcpyDeviceToHost);
        const hipError_t err = hipGetLastError();
        float t = 0;
        for (int r = 0; r < 3; r++) t = time_us(launch, 20);
        printf("   %-44s %8.1f us (%5.0f TF)   mismatches %llu %s\n", what, t, fl / t / 1e6, d, err == hipSuccess ? "" : hipGetErrorString(err));
    };
#ifdef WITH_VENDOR
    {
        float tv = vendor_us(A, B, C1, M, N, K, 20);
        tv = vendor_us(A, B, C1, M, N, K, 20);
        printf("   %-44s %8.1f us (%5.0f TF)\n", "VENDOR hipBLASLt bf16 -> bf16", tv, fl / tv / 1e6);
    }
#endif
    check("stag 256x256 (8 waves, 32-deep stages)", [&] { launch_stag<8>(A, B, g, e1); });
    check("stag 320x256", [&] { launch_stag<10>(A, B, g, e1); });
    check("stag 160x256", [&] { launch_stag<5>(A, B, g, e1); });
    check("stag 256x192", [&] { (launch_stag<8, 3>(A, B, g, e1)); });
    check("stag 320x192", [&] { (launch_stag<10, 3>(A, B, g, e1)); });
    const bool ok2 = (K % 128) == 0 && K >= 256, ok3 = (K % 192) == 0 && K >= 384;
    if (ok2) check("q4 256x256 NS2 (4 waves, 64-deep full lines)", [&] { (launch_q4<8, 2>(A, B, g, e1, grid)); });
    if (ok3) check("q4 160x256 NS3", [&] { (launch_q4<5, 3>(A, B, g, e1, grid)); });
    if (ok2) check("q4 160x256 NS2", [&] { (launch_q4<5, 2>(A, B, g, e1, grid)); });
    if (ok2) check("q4 256x256 NS2 variant: read every MFMA", [&] { (launch_q4<8, 2, 1>(A, B, g, e1, grid)); });
    if (ok3) check("q4 160x256 NS3 variant: read every MFMA", [&] { (launch_q4<5, 3, 1>(A, B, g, e1, grid)); });
    if (ok2) check("q4 256x256 NS2 ABLATION no LDS-DMA", [&] { (launch_q4<8, 2, 2>(A, B, g, e1, grid)); });
    if (ok3) check("q4 160x256 NS3 ABLATION no LDS-DMA", [&] { (launch_q4<5, 3, 2>(A, B, g, e1, grid)); });
    if (ok2) check("q4 256x256 NS2 ABLATION no fragment reads", [&] { (launch_q4<8, 2, 3>(A, B, g, e1, grid)); });
    if (ok3) check("q4 160x256 NS3 ABLATION no fragment reads", [&] { (launch_q4<5, 3, 3>(A, B, g, e1, grid)); });
    if (N == 2304 || N == 768) {      // product functors of the plain launches: bias (c_attn) — specialised vs run-time switches
        float* bias; hipMalloc(&bias, N * 4); hipMemset(bias, 0, N * 4);
        EpiBF16 f{C1, nullptr, bias, N, M, N, 0};
        EpiBF16S<0, false> fs; static_cast<EpiBF16&>(fs) = f;
        float a = 0, b = 0, c = 0, d = 0;
        for (int r = 0; r < 2; r++) {
            a = time_us([&] { (launch_stag<8, 4, EpiBF16>(A, B, g, f)); }, 20); b = time_us([&] { (launch_stag<5, 4, EpiBF16>(A, B, g, f)); }, 20);
            if (ok2) c = time_us([&] { (launch_q4<8, 2, 0, EpiBF16S<0, false>>(A, B, g, fs, grid)); }, 20);
            if (ok3) d = time_us([&] { (launch_q4<5, 3, 0, EpiBF16S<0, false>>(A, B, g, fs, grid)); }, 20);
        }
        printf("   bias functor: stag256 EpiBF16 %7.1f | stag160 EpiBF16 %7.1f | q4-256 specialised %7.1f | q4-160 specialised %7.1f us\n", a, b, c, d);
        hipFree(bias);
    }
    hipFree(A); hipFree(B); hipFree(C0); hipFree(C1); hipFree(dcount);
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    run("warm-up", 4096, 4096, 1536, grid);
    run("c_attn", 12800, 2304, 768, grid);
    run("c_fc", 12800, 3072, 768, grid);
    run("lm_head", 10240, 50304, 768, grid);
    run("c_proj", 12800, 768, 768, grid);
    run("c_attn dgrad", 12800, 768, 2304, grid);
    run("mlp.c_proj", 12800, 768, 3072, grid);
    run("mapper qkv", 5120, 2304, 768, grid);
    run("4096^3", 4096, 4096, 4096, grid);
    run("8192^3", 8192, 8192, 8192, grid);
    run("edge", 1000, 520, 256, grid);
    return 0;
}
