// Mapper-sized NT GEMMs (M = 5120 = 256 samples x 20 rows): the product's small-grid kernels against each other and, with -DWITH_VENDOR -lhipblaslt,
// hipBLASLt's bf16 -> bf16 GEMM — same buffers, same process, HIP events around 20 launches; K swept to separate the per-launch cost from the K-loop slope.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../clipcap_amd/csrc [-DWITH_VENDOR] -o mapper_gemm_bench mapper_gemm_bench.hip [-lhipblaslt]
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
        p[i] = f2op(((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f));
    }
}
static __global__ void k_diff(const unsigned short* a, const unsigned short* b, size_t n, unsigned long long* out) {
    unsigned long long d = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d += a[i] != b[i];
    if (d) atomicAdd(out, d);
}
#ifdef WITH_VENDOR
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
template <int NI, int NJ = 4>
static void launch_stag(const op16_t* A, const op16_t* B, const GemmShape& g, const EpiBF16Plain& e) {
    constexpr size_t sh = (size_t)H_NS * (32 * NI + H_BN) * H_BK * 2;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)gemm_nt_stag256_kernel<EpiBF16Plain, NJ, false, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
    const dim3 gr((unsigned)(((g.M + 32 * NI - 1) / (32 * NI)) * ((g.N + 64 * NJ - 1) / (64 * NJ))));
    hipLaunchKernelGGL((gemm_nt_stag256_kernel<EpiBF16Plain, NJ, false, NI>), gr, dim3(512), sh, 0, A, B, g, e);
}
template <int NI, int NS>
static void launch_q4(const op16_t* A, const op16_t* B, const GemmShape& g, const EpiBF16Plain& e) {
    constexpr size_t sh = (size_t)NS * (32 * NI + 256) * 128;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)gemm_nt_q4_kernel<EpiBF16Plain, NI, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
    const int tiles = ((g.M + 32 * NI - 1) / (32 * NI)) * ((g.N + 255) / 256);
    hipLaunchKernelGGL((gemm_nt_q4_kernel<EpiBF16Plain, NI, NS>), dim3(tiles < 256 ? tiles : 256), dim3(256), sh, 0, A, B, g, e);
}

static void run(const char* name, int M, int N, int K) {
    op16_t *A, *B; act_t *C0, *C1; unsigned long long* dcount; float* bias;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2);
    hipMalloc(&C0, (size_t)M * N * 2); hipMalloc(&C1, (size_t)M * N * 2); hipMalloc(&dcount, 8);
    hipMalloc(&bias, N * 4); hipMemset(bias, 0, N * 4);
    k_fill<<<1024, 256>>>(A, (size_t)M * K, 17u);
    k_fill<<<1024, 256>>>(B, (size_t)N * K, 91u);
    GemmShape g;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.k_chunk = K; g.group_m = 8; g.stagger = 0;
    EpiBF16Plain e0{C0, bias, N, M, N};
    EpiBF16Plain e1{C1, bias, N, M, N};
    const double fl = 2.0 * M * N * K;
    printf("%-14s M=%5d N=%5d K=%5d  %6.2f GFLOP\n", name, M, N, K, fl / 1e9);
    const dim3 grid128((unsigned)(((M + 127) / 128) * ((N + 127) / 128)));
    constexpr size_t sh4 = (size_t)8 * G_TILE_BYTES;
    hipFuncSetAttribute((const void*)gemm_nt_glds4_kernel<EpiBF16Plain>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh4);
    hipFuncSetAttribute((const void*)gemm_nt_glds4x2_kernel<EpiBF16Plain>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh4);
    auto ref = [&](const EpiBF16Plain& e) { hipLaunchKernelGGL((gemm_nt_glds_kernel<EpiBF16Plain>), grid128, dim3(G_THREADS), 0, 0, A, B, g, e); };
    ref(e0);
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
        printf("   %-50s %8.1f us (%5.0f TF)   mismatches %llu %s\n", what, t, fl / t / 1e6, d, err == hipSuccess ? "" : hipGetErrorString(err));
    };
#ifdef WITH_VENDOR
    {
        float tv = vendor_us(A, B, C1, M, N, K, 20);
        tv = vendor_us(A, B, C1, M, N, K, 20);
        printf("   %-50s %8.1f us (%5.0f TF)\n", "VENDOR hipBLASLt bf16 -> bf16", tv, fl / tv / 1e6);
    }
#endif
    check("glds 128x128 (4 waves, 2 stages, 2 blocks / CU)", [&] { ref(e1); });
    check("glds4 128x128 (4 waves, 4 stages)", [&] { hipLaunchKernelGGL((gemm_nt_glds4_kernel<EpiBF16Plain>), grid128, dim3(G_THREADS), sh4, 0, A, B, g, e1); });
    check("glds4x2 128x128 (8 waves, 4 stages)  [product]", [&] { hipLaunchKernelGGL((gemm_nt_glds4x2_kernel<EpiBF16Plain>), grid128, dim3(2 * G_THREADS), sh4, 0, A, B, g, e1); });
    check("stag 160x256", [&] { launch_stag<5>(A, B, g, e1); });
    check("stag 256x128", [&] { (launch_stag<8, 2>(A, B, g, e1)); });
    check("stag 256x192", [&] { (launch_stag<8, 3>(A, B, g, e1)); });
    check("stag 256x256", [&] { launch_stag<8>(A, B, g, e1); });
    if (K % 192 == 0 && K >= 384) check("q4 160x256 NS3", [&] { (launch_q4<5, 3>(A, B, g, e1)); });
    hipFree(A); hipFree(B); hipFree(C0); hipFree(C1); hipFree(dcount); hipFree(bias);
}

int main() {
    run("warm-up", 4096, 4096, 1536);
    for (int K : {384, 768, 1536, 2304, 3072}) run("dgrad N=768", 5120, 768, K);
    run("fc1 fwd", 5120, 1536, 768);
    run("fc2 dgrad(dact)", 5120, 1536, 768);
    run("qkv fwd", 5120, 2304, 768);
    run("B=1024 N=768", 20480, 768, 768);
    return 0;
}
