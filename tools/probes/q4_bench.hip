// The persistent 4-wave NT kernel (gemm_q4.hip.h) against the 8-wave staggered kernel on the model's multi-round shapes: bit equality of
// the outputs (both accumulate one MFMA per 32-deep K-step in the same order) and GPU-side durations (HIP events around 20 launches,
// interleaved A / B / A / B rounds).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../clipcap_amd/csrc -o q4_bench q4_bench.hip
#define CC_Q4_VARIANTS
#include "gemm.hip.h"
#include <cstdio>
#include <cstring>
#include <vector>
#ifdef WITH_VENDOR      // measuring stick only (hipcc ... -DWITH_VENDOR -lhipblaslt): hipBLASLt's bf16 -> bf16 GEMM on the same buffers in the same process
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
        const float f = ((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f);      // uniform [-1, 1)
        p[i] = f2op(f);
    }
}
static __global__ void k_diff(const unsigned short* a, const unsigned short* b, size_t n, unsigned long long* out) {
    unsigned long long d = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d += a[i] != b[i];
    if (d) atomicAdd(out, d);
}

// what a functor without run-time switches costs: bias add + convert + one 16-B store per unit, nothing else
struct EpiLean {
    act_t* C; const float* bias; int ldc, M, Ns;
    static constexpr bool kPre = false;
    __device__ __forceinline__ void pre4(int, int, f32x4&) const {}
    __device__ __forceinline__ void bias8(int col, float (&b)[8]) const { load_bias8(bias, col, Ns, b); }
    __device__ __forceinline__ void fin(int row, int col, float (&v)[8], const float (&b)[8]) const {
        if (row >= M || col >= Ns) return;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += b[e];
        act_st8(C + (size_t)row * ldc + col, v);
    }
};
// EpiBF16 with its run-time switches as template parameters (what a specialised functor costs)
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
template <int NI, int VAR = 0, class E = EpiBF16>
static void launch_q4(const op16_t* A, const op16_t* B, const GemmShape& g, const E& e, int grid) {
    constexpr size_t sh = (size_t)4 * (32 * NI + 256) * 64;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)gemm_nt_q4_kernel<E, NI, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
    const int tiles = ((g.M + 32 * NI - 1) / (32 * NI)) * ((g.N + 255) / 256);
    hipLaunchKernelGGL((gemm_nt_q4_kernel<E, NI, VAR>), dim3(tiles < grid ? tiles : grid), dim3(256), sh, 0, A, B, g, e);
}
template <int VAR>
static void run_variant(const char* what, const op16_t* A, const op16_t* B, const GemmShape& g, const EpiBF16& e0, const EpiBF16& e1, int grid) {
    unsigned long long* dcount; hipMalloc(&dcount, 8); hipMemset(dcount, 0, 8);
    hipMemset(e1.C, 0x7f, (size_t)g.M * g.N * 2);
    launch_q4<8, VAR>(A, B, g, e1, grid);
    k_diff<<<1024, 256>>>(e0.C, e1.C, (size_t)g.M * g.N, dcount);
    unsigned long long d = 0;
    hipMemcpy(&d, dcount, 8, hipMemcpyDeviceToHost);
    float t[3];
    for (int r = 0; r < 3; r++) t[r] = time_us([&] { launch_q4<8, VAR>(A, B, g, e1, grid); }, 20);
    printf("   variant %d (%-34s): %7.1f %7.1f %7.1f us (%5.0f TF)  mismatches %llu\n", VAR, what, t[0], t[1], t[2], 2.0 * g.M * g.N * g.K / t[2] / 1e6, d);
    hipFree(dcount);
}
template <int NI, class E = EpiBF16>
static void launch_stag(const op16_t* A, const op16_t* B, const GemmShape& g, const E& e) {
    constexpr size_t sh = (size_t)H_NS * (32 * NI + H_BN) * H_BK * 2;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)gemm_nt_stag256_kernel<E, 4, false, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
    const dim3 gr((unsigned)(((g.M + 32 * NI - 1) / (32 * NI)) * ((g.N + 255) / 256)));
    hipLaunchKernelGGL((gemm_nt_stag256_kernel<E, 4, false, NI>), gr, dim3(512), sh, 0, A, B, g, e);
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
static void run(const char* name, int M, int N, int K, int grid, bool variants = false) {
    op16_t *A, *B; act_t *C0, *C1; unsigned long long* dcount;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2);
    hipMalloc(&C0, (size_t)M * N * 2); hipMalloc(&C1, (size_t)M * N * 2); hipMalloc(&dcount, 8);
    k_fill<<<1024, 256>>>(A, (size_t)M * K, 17u);
    k_fill<<<1024, 256>>>(B, (size_t)N * K, 91u);
    GemmShape g;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.k_chunk = K; g.group_m = 8; g.stagger = 0;
    EpiBF16 e0{C0, nullptr, nullptr, N, M, N, 0};
    EpiBF16 e1{C1, nullptr, nullptr, N, M, N, 0};
    printf("%-14s M=%5d N=%5d K=%5d  %6.2f GFLOP\n", name, M, N, K, 2.0 * M * N * K / 1e9);
    for (int ni : {8, 10}) {
        hipMemset(C0, 0xff, (size_t)M * N * 2); hipMemset(C1, 0x7f, (size_t)M * N * 2); hipMemset(dcount, 0, 8);
        if (ni == 8) { launch_stag<8>(A, B, g, e0); launch_q4<8>(A, B, g, e1, grid); }
        else { launch_stag<10>(A, B, g, e0); launch_q4<10>(A, B, g, e1, grid); }
        k_diff<<<1024, 256>>>(C0, C1, (size_t)M * N, dcount);
        unsigned long long d = 0;
        hipMemcpy(&d, dcount, 8, hipMemcpyDeviceToHost);
        const hipError_t err = hipGetLastError();
        float ts[3], tq[3];
        for (int r = 0; r < 3; r++) {
            if (ni == 8) { ts[r] = time_us([&] { launch_stag<8>(A, B, g, e0); }, 20); tq[r] = time_us([&] { launch_q4<8>(A, B, g, e1, grid); }, 20); }
            else { ts[r] = time_us([&] { launch_stag<10>(A, B, g, e0); }, 20); tq[r] = time_us([&] { launch_q4<10>(A, B, g, e1, grid); }, 20); }
        }
        const double fl = 2.0 * M * N * K;
        printf("   tile %3dx256: stag %7.1f %7.1f %7.1f us (%5.0f TF) | q4 %7.1f %7.1f %7.1f us (%5.0f TF)  ratio %.3f  mismatches %llu %s\n", 32 * ni, ts[0], ts[1], ts[2],
               fl / ts[2] / 1e6, tq[0], tq[1], tq[2], fl / tq[2] / 1e6, ts[2] / tq[2], d, err == hipSuccess ? "" : hipGetErrorString(err));
    }
#ifdef WITH_VENDOR
    {
        hipMemset(dcount, 0, 8);
        const float tv = vendor_us(A, B, C1, M, N, K, 20);
        launch_stag<8>(A, B, g, e0);
        k_diff<<<1024, 256>>>(C0, C1, (size_t)M * N, dcount);
        unsigned long long d = 0;
        hipMemcpy(&d, dcount, 8, hipMemcpyDeviceToHost);
        float ts = time_us([&] { launch_stag<8>(A, B, g, e0); }, 20), ts10 = time_us([&] { launch_stag<10>(A, B, g, e0); }, 20);
        ts = time_us([&] { launch_stag<8>(A, B, g, e0); }, 20); ts10 = time_us([&] { launch_stag<10>(A, B, g, e0); }, 20);
        const float tv2 = vendor_us(A, B, C1, M, N, K, 20);
        printf("   VENDOR hipBLASLt bf16 -> bf16, same buffers, same process: %7.1f / %7.1f us (%5.0f TF) | stag256 %7.1f | stag320 %7.1f us   (outputs differing from ours in the last bf16 bit or more: %llu of %zu)\n", tv, tv2,
               2.0 * M * N * K / tv2 / 1e6, ts, ts10, d, (size_t)M * N);
    }
#endif
    {   // no-store ablation (every row discarded by the functor) and the start stagger
        EpiBF16 en{C1, nullptr, nullptr, N, 0, N, 0};
        float a = time_us([&] { launch_stag<8>(A, B, g, en); }, 20), b = time_us([&] { launch_q4<8>(A, B, g, en, grid); }, 20);
        a = time_us([&] { launch_stag<8>(A, B, g, en); }, 20); b = time_us([&] { launch_q4<8>(A, B, g, en, grid); }, 20);
        printf("   NO STORES 256x256: stag %7.1f us | q4 %7.1f us\n", a, b);
        EpiLean l1{C1, nullptr, N, M, N}, l0{C1, nullptr, 0, M, N};
        for (int r = 0; r < 2; r++) { a = time_us([&] { (launch_stag<8, EpiLean>(A, B, g, l1)); }, 20); b = time_us([&] { (launch_q4<8, 0, EpiLean>(A, B, g, l1, grid)); }, 20); }
        printf("   LEAN functor 256x256: stag %7.1f us | q4 %7.1f us\n", a, b);
        for (int r = 0; r < 2; r++) { a = time_us([&] { (launch_stag<8, EpiLean>(A, B, g, l0)); }, 20); b = time_us([&] { (launch_q4<8, 0, EpiLean>(A, B, g, l0, grid)); }, 20); }
        printf("   LEAN functor, every row stored to row 0 (cache-resident): stag %7.1f us | q4 %7.1f us\n", a, b);
    }
    if (N == 3072) {      // c_fc forward as the product launches it: bias + gelu_new, gelu' stored beside
        float* bias; act_t* P; hipMalloc(&bias, N * 4); hipMemset(bias, 0, N * 4); hipMalloc(&P, (size_t)M * N * 2);
        EpiBF16 f{C1, P, bias, N, M, N, 3}; f.pre_nt = true;
        EpiBF16S<3, true> fs; static_cast<EpiBF16&>(fs) = f;
        float a = 0, b = 0, c = 0, d = 0;
        for (int r = 0; r < 2; r++) {
            a = time_us([&] { (launch_stag<10, EpiBF16>(A, B, g, f)); }, 20); b = time_us([&] { (launch_stag<10, EpiBF16S<3, true>>(A, B, g, fs)); }, 20);
            c = time_us([&] { (launch_q4<8, 0, EpiBF16S<3, true>>(A, B, g, fs, grid)); }, 20); d = time_us([&] { (launch_q4<10, 0, EpiBF16S<3, true>>(A, B, g, fs, grid)); }, 20);
        }
        printf("   c_fc functor (bias, gelu, gelu' store): stag320 EpiBF16 %7.1f | stag320 specialised %7.1f | q4-256 specialised %7.1f | q4-320 specialised %7.1f us\n", a, b, c, d);
        hipFree(bias); hipFree(P);
    }
    if (N == 3072) {      // the activation-gradient GEMM (mlp.c_proj input gradient): C = acc * gelu'(aux), aux stored by the forward
        act_t* P; hipMalloc(&P, (size_t)M * N * 2); hipMemset(P, 0x3f, (size_t)M * N * 2);
        EpiDAct f{C1, P, N, M, N, 3};
        float t128 = 0, a = 0, b = 0, c = 0, d = 0;
        for (int r = 0; r < 2; r++) {
            t128 = time_us([&] { launch_gemm(0, 0, A, K, B, K, M, N, K, 1, f, (hipStream_t)0, 0); }, 20);
            a = time_us([&] { (launch_stag<10, EpiDAct>(A, B, g, f)); }, 20); b = time_us([&] { (launch_stag<8, EpiDAct>(A, B, g, f)); }, 20);
            c = time_us([&] { (launch_q4<8, 0, EpiDAct>(A, B, g, f, grid)); }, 20); d = time_us([&] { (launch_q4<10, 0, EpiDAct>(A, B, g, f, grid)); }, 20);
        }
        printf("   activation-gradient functor (x gelu' from aux): 128x128 %7.1f | stag320 %7.1f | stag256 %7.1f | q4-256 %7.1f | q4-320 %7.1f us\n", t128, a, b, c, d);
        hipFree(P);
    }
    if (N == 2304) {      // c_attn forward as the product launches it (bias, no activation), the functor's switches as template parameters
        float* bias; hipMalloc(&bias, N * 4); hipMemset(bias, 0, N * 4);
        EpiBF16 f{C1, nullptr, bias, N, M, N, 0};
        EpiBF16S<0, false> fs; static_cast<EpiBF16&>(fs) = f;
        float a = 0, b = 0, c = 0, d = 0;
        for (int r = 0; r < 2; r++) {
            a = time_us([&] { (launch_stag<8, EpiBF16>(A, B, g, f)); }, 20); b = time_us([&] { (launch_stag<8, EpiBF16S<0, false>>(A, B, g, fs)); }, 20);
            c = time_us([&] { (launch_q4<8, 0, EpiBF16S<0, false>>(A, B, g, fs, grid)); }, 20); d = time_us([&] { (launch_q4<10, 0, EpiBF16S<0, false>>(A, B, g, fs, grid)); }, 20);
        }
        printf("   c_attn functor (bias): stag256 EpiBF16 %7.1f | stag256 specialised %7.1f | q4-256 specialised %7.1f | q4-320 specialised %7.1f us\n", a, b, c, d);
        hipFree(bias);
    }
    if (N == 50304) {     // lm_head forward as the product launches it: exponential form + row partials
        float *pm, *ps, *cref; const int np = N / 64;
        hipMalloc(&pm, (size_t)M * np * 4); hipMalloc(&ps, (size_t)M * np * 4); hipMalloc(&cref, M * 4); hipMemset(cref, 0, M * 4);
        EpiLMHeadExp le{C1, pm, ps, cref, N, M, 50257, np};
        float a = 0, b = 0, c = 0, d = 0;
        for (int r = 0; r < 2; r++) {
            a = time_us([&] { (launch_stag<10, EpiLMHeadExp>(A, B, g, le)); }, 10); b = time_us([&] { (launch_stag<8, EpiLMHeadExp>(A, B, g, le)); }, 10);
            c = time_us([&] { (launch_q4<8, 0, EpiLMHeadExp>(A, B, g, le, grid)); }, 10); d = time_us([&] { (launch_q4<10, 0, EpiLMHeadExp>(A, B, g, le, grid)); }, 10);
        }
        printf("   lm_head exponential-form functor: stag320 %7.1f | stag256 %7.1f | q4-256 %7.1f | q4-320 %7.1f us\n", a, b, c, d);
        hipFree(pm); hipFree(ps); hipFree(cref);
    }
    if (variants) {
        launch_stag<8>(A, B, g, e0);
        run_variant<1>("a read after every MFMA", A, B, g, e0, e1, grid);
        run_variant<2>("a read every 3rd MFMA, DMA gap 7", A, B, g, e0, e1, grid);
        run_variant<5>("all reads first, DMAs behind them", A, B, g, e0, e1, grid);
        run_variant<3>("ABLATION no LDS-DMA", A, B, g, e0, e1, grid);
        run_variant<4>("ABLATION no barrier", A, B, g, e0, e1, grid);
    }
    hipFree(A); hipFree(B); hipFree(C0); hipFree(C1); hipFree(dcount);
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    run("c_attn", 12800, 2304, 768, grid, true);
    run("c_fc", 12800, 3072, 768, grid);
    run("lm_head", 10240, 50304, 768, grid, true);
    run("c_proj", 12800, 768, 768, grid);
    run("mlp.c_proj", 12800, 768, 3072, grid);
    run("4096^3", 4096, 4096, 4096, grid);
    run("8192^3", 8192, 8192, 8192, grid, true);
    run("edge", 1000, 520, 256, grid);
    return 0;
}
