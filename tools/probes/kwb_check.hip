// gemm_nt_s64kwb_kernel (weight operand global -> VGPR from the fragment-ordered image) against gemm_nt_s64kw_kernel (both operands through
// LDS): same results, time per launch.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../clipcap_amd/csrc -o kwb_check kwb_check.hip
#include "gemm.hip.h"
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
using namespace CC_NS;
namespace cc_shared { int g_gemm_tile_mode = -1, g_gemm_s64 = -1, g_gemm_small_x2 = 1, g_decode_last_path = 0; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
// cold = 1: every launch reads a different copy of the weights (enough copies to exceed the 256 MB Infinity Cache), as the decode chain
// does (708 MB of weights per generated position); cold = 0: the same 6-8 MB every launch (L2 / MALL hits).
template <int RB> static void launch_b(dim3 gr, size_t sh, const op16_t* A, const op16_t* I, GemmShape g, EpiF32 e) {
    hipLaunchKernelGGL((gemm_nt_s64kwb_kernel<EpiF32, RB>), gr, dim3(320), sh, 0, A, I, g, e);
}
static void run(int M, int N, int K, int ks, int cold) {
    std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hB) v = f2bf(rnd());
    const size_t wb = hB.size() * 2;
    const int ncopy = cold ? (int)((640u << 20) / wb) : 1;
    op16_t *A, *B, *I; float *C0, *C1;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&B, wb * ncopy); hipMalloc(&I, wb * ncopy);
    hipMalloc(&C0, (size_t)ks * M * N * 4); hipMalloc(&C1, (size_t)ks * M * N * 4);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    for (int c = 0; c < ncopy; c++) {
        hipMemcpy((char*)B + c * wb, hB.data(), wb, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_skinny_image, dim3(1024), dim3(256), 0, 0, (const op16_t*)((char*)B + c * wb), (op16_t*)((char*)I + c * wb), N, K);
    }
    { hipError_t er = hipDeviceSynchronize(); if (er != hipSuccess) printf("image: %s\n", hipGetErrorString(er)); }
    GemmShape g;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.group_m = 8; g.stagger = 0;
    const int kt = K / 64, per = (kt + ks - 1) / ks;
    g.k_chunk = per * 64;
    const int kse = (kt + per - 1) / per;
    EpiF32 e0{C0, nullptr, N, M, N, ks > 1 ? 3 : 0, 1.0f}, e1{C1, nullptr, N, M, N, ks > 1 ? 3 : 0, 1.0f};
    e0.zstride = e1.zstride = (size_t)M * N;
    const size_t sh = 4 * 128 * 128;
    hipFuncSetAttribute((const void*)gemm_nt_s64kw_kernel<EpiF32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipFuncSetAttribute((const void*)gemm_nt_s64kwb_kernel<EpiF32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipFuncSetAttribute((const void*)gemm_nt_s64kwb_kernel<EpiF32, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipFuncSetAttribute((const void*)gemm_nt_s64kwb_kernel<EpiF32, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    const dim3 gr((unsigned)(((M + 63) / 64) * (N / 64)), 1, (unsigned)kse);
    float t[4];
    double md[4] = {0, 0, 0, 0}, mx = 0;
    std::vector<float> h0((size_t)kse * M * N), h1((size_t)kse * M * N);
    for (int v = 0; v < 4; v++) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        auto go = [&](int it) {
            const op16_t* Bc = (const op16_t*)((const char*)B + (size_t)(it % ncopy) * wb);
            const op16_t* Ic = (const op16_t*)((const char*)I + (size_t)(it % ncopy) * wb);
            if (v == 0) hipLaunchKernelGGL((gemm_nt_s64kw_kernel<EpiF32>), gr, dim3(256), sh, 0, A, Bc, g, e0);
            else if (v == 1) launch_b<4>(gr, sh, A, Ic, g, e1);
            else if (v == 2) launch_b<8>(gr, sh, A, Ic, g, e1);
            else launch_b<16>(gr, sh, A, Ic, g, e1);
        };
        hipMemset(C1, 0, h1.size() * 4);
        for (int it = 0; it < 3; it++) go(it);
        hipError_t er = hipDeviceSynchronize();
        if (er != hipSuccess) { printf("variant %d: %s\n", v, hipGetErrorString(er)); return; }
        hipEventRecord(a);
        for (int it = 0; it < 400; it++) go(it + 3);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&t[v], a, b);
        if (v == 0) hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost);
        else {
            hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < h0.size(); i++) { md[v] = std::fmax(md[v], std::fabs((double)h0[i] - h1[i])); mx = std::fmax(mx, std::fabs((double)h0[i])); }
        }
    }
    printf("%s M=%d N=%d K=%d slices %d (%d blocks): LDS-B %.2f us | image-B ring 4: %.2f, 8: %.2f, 16: %.2f us; max |diff| %.3g %.3g %.3g of max |C| %.3g\n", cold ? "COLD" : "hot ", M, N, K, kse,
           gr.x * gr.z, t[0] * 2.5, t[1] * 2.5, t[2] * 2.5, t[3] * 2.5, md[1], md[2], md[3], mx);
    hipFree(A); hipFree(B); hipFree(I); hipFree(C0); hipFree(C1);
}
int main() {
    for (int cold = 1; cold >= 0; cold--) {
        run(320, 3072, 1024, 1, cold);
        run(320, 1024, 4096, 3, cold);
        run(320, 1024, 1024, 1, cold);
        run(300, 1024, 1024, 3, cold);
        run(64, 4096, 1024, 1, cold);
        run(320, 4096, 1024, 1, cold);
    }
    return 0;
}
