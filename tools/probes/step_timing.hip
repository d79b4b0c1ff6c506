// Where does a K-step of the 64 x 64 skinny GEMM go?  Same loop as gemm_nt_s64_kernel<.,1,4,1> (4 waves, 4 stages of 16 KiB, 128-B LDS
// rows), s_memtime stamps around each phase, wave 0 of a few blocks.   hipcc --offload-arch=gfx950 -O3 -o step_timing step_timing.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1) ^ ((row >> 4) & 3)) & 7) << 4); }
constexpr int NS = 4, STAGE = 16384, NPH = 6;
__global__ __launch_bounds__(256, 2) void k_step(const char* __restrict__ A, const char* __restrict__ W, int K2, int tiles_m, float* out,
                                                 unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(1024))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int nk = K2 / 128;
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    auto issue = [&](int t, int slot) {
        char* st = sm + slot * STAGE;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int seg = wave + 4 * i;
            const int row = (seg & 7) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7) ^ ((row >> 4) & 3);
            const char* src = (seg < 8 ? A + (size_t)(tm * 64 + row) * K2 : W + (size_t)(tn * 64 + row) * K2) + t * 128 + chunk * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + seg * 1024), 16, 0, 0);
        }
    };
    for (int t = 0; t < NS - 1; t++) if (t < nk) issue(t, t);
    const int frow = lane & 31, fhalf = lane >> 5;
    int slot = 0, islot = NS - 1;
    unsigned long long ph[NPH] = {0, 0, 0, 0, 0, 0};
    for (int kt = 0; kt < nk; kt++) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        const int rem = min(nk - 1, kt + NS - 2) - kt;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        const unsigned long long t2 = __builtin_readcyclecounter();
        if (kt + NS - 1 < nk) issue(kt + NS - 1, islot);
        const unsigned long long t3 = __builtin_readcyclecounter();
        const char* cur = sm + slot * STAGE;
        bf16x8 a[4], b[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            a[kk] = *reinterpret_cast<const bf16x8*>(cur + lds_off(wm * 32 + frow, kk * 2 + fhalf));
            b[kk] = *reinterpret_cast<const bf16x8*>(cur + 8192 + lds_off(wn * 32 + frow, kk * 2 + fhalf));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t4 = __builtin_readcyclecounter();
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk], b[kk], acc, 0, 0, 0);
        const unsigned long long t5 = __builtin_readcyclecounter();
        asm volatile("s_nop 0" ::"v"(acc[0]));          // forces the MFMA results to be waited for here
        const unsigned long long t6 = __builtin_readcyclecounter();
        if (kt >= 2 && kt + 2 < nk) { ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += t6 - t5; }
        slot = slot + 1 == NS ? 0 : slot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc[r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (lane == 0)
        for (int p = 0; p < NPH; p++) stamps[((size_t)blockIdx.x * 4 + wave) * NPH + p] = ph[p];
}
// V1/V2: every wave computes the whole 64 x 64 tile for ONE 16-k quarter of each K-tile (4 ds_read_b128 + 4 MFMAs per step and wave: half
// the LDS fragment traffic of the 2 x 2 wave layout); V2 additionally issues the DMA after the fragment reads.  Whole-loop cycles only.
template <int V>
__global__ __launch_bounds__(256, 2) void k_step_v(const char* __restrict__ A, const char* __restrict__ W, int K2, int tiles_m, float* out,
                                                   unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(1024))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int nk = K2 / 128;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    auto issue = [&](int t, int slot) {
        char* st = sm + slot * STAGE;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int seg = wave + 4 * i;
            const int row = (seg & 7) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7) ^ ((row >> 4) & 3);
            const char* src = (seg < 8 ? A + (size_t)(tm * 64 + row) * K2 : W + (size_t)(tn * 64 + row) * K2) + t * 128 + chunk * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + seg * 1024), 16, 0, 0);
        }
    };
    for (int t = 0; t < NS - 1; t++) if (t < nk) issue(t, t);
    const int frow = lane & 31, fhalf = lane >> 5;
    int slot = 0, islot = NS - 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int kt = 0; kt < nk; kt++) {
        const int rem = min(nk - 1, kt + NS - 2) - kt;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (V == 1 && kt + NS - 1 < nk) issue(kt + NS - 1, islot);
        const char* cur = sm + slot * STAGE;
        bf16x8 a[2], b[2];
        if (V == 0) {
        } else {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                a[i] = *reinterpret_cast<const bf16x8*>(cur + lds_off(i * 32 + frow, wave * 2 + fhalf));
                b[i] = *reinterpret_cast<const bf16x8*>(cur + 8192 + lds_off(i * 32 + frow, wave * 2 + fhalf));
            }
        }
        if (V == 2 && kt + NS - 1 < nk) issue(kt + NS - 1, islot);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        slot = slot + 1 == NS ? 0 : slot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
    }
    float s = 0;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * 256 + tid] = s + wm + wn;
    if (lane == 0) stamps[(size_t)blockIdx.x * 4 + wave] = t1 - t0;
}
// V3: the 2 x 2 wave layout, software-pipelined: fragment reads of tile t+1 are issued before the MFMAs of tile t, DMA after them.
// V4: the same with the K split over the waves (whole tile per wave, a quarter of each K-tile).
template <int V>
__global__ __launch_bounds__(256, 2) void k_step_p(const char* __restrict__ A, const char* __restrict__ W, int K2, int tiles_m, float* out,
                                                   unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(1024))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int nk = K2 / 128;
    constexpr int NA = V == 3 ? 4 : 2;            // fragments per operand and step
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    auto issue = [&](int t, int slot) {
        char* st = sm + slot * STAGE;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int seg = wave + 4 * i;
            const int row = (seg & 7) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7) ^ ((row >> 4) & 3);
            const char* src = (seg < 8 ? A + (size_t)(tm * 64 + row) * K2 : W + (size_t)(tn * 64 + row) * K2) + t * 128 + chunk * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + seg * 1024), 16, 0, 0);
        }
    };
    const int frow = lane & 31, fhalf = lane >> 5;
    auto readf = [&](int t, bf16x8 (&a)[NA], bf16x8 (&b)[NA]) {
        const char* cur = sm + (t & 3) * STAGE;
#pragma unroll
        for (int q = 0; q < NA; q++) {
            if (V == 3) {
                a[q] = *reinterpret_cast<const bf16x8*>(cur + lds_off(wm * 32 + frow, q * 2 + fhalf));
                b[q] = *reinterpret_cast<const bf16x8*>(cur + 8192 + lds_off(wn * 32 + frow, q * 2 + fhalf));
            } else {
                a[q] = *reinterpret_cast<const bf16x8*>(cur + lds_off(q * 32 + frow, wave * 2 + fhalf));
                b[q] = *reinterpret_cast<const bf16x8*>(cur + 8192 + lds_off(q * 32 + frow, wave * 2 + fhalf));
            }
        }
    };
    auto mfma = [&](bf16x8 (&a)[NA], bf16x8 (&b)[NA]) {
        if (V == 3) {
#pragma unroll
            for (int q = 0; q < NA; q++) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b[q], acc[0][0], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    for (int t = 0; t < NS; t++) if (t < nk) issue(t, t);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // tiles 0 and 1 landed (nk >= 4 here)
    __builtin_amdgcn_s_barrier();
    bf16x8 a0[NA], b0[NA], a1[NA], b1[NA];
    readf(0, a0, b0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_readcyclecounter();
    auto step = [&](int kt, bf16x8 (&a)[NA], bf16x8 (&b)[NA], bf16x8 (&an)[NA], bf16x8 (&bn)[NA]) {
        if (kt + 1 < nk) readf(kt + 1, an, bn);
        mfma(a, b);
        if (kt + NS < nk) issue(kt + NS, kt & 3);
        const int rem = max(0, min(nk - 1, kt + NS) - (kt + 2));
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, a0, b0, a1, b1);
        if (kt + 1 < nk) step(kt + 1, a1, b1, a0, b0);
    }
    float s = 0;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * 256 + tid] = s + wm + wn;
    if (lane == 0) stamps[(size_t)blockIdx.x * 4 + wave] = t1 - t0;
}
int main() {
    const int M = 320, N = 1024;
    char *A, *W; float* out; unsigned long long* st;
    hipMalloc(&A, (size_t)M * 4096 * 2); hipMalloc(&W, (size_t)4096 * 4096 * 2);
    hipMemset(A, 0, (size_t)M * 4096 * 2); hipMemset(W, 0, (size_t)4096 * 4096 * 2);
    for (int cfg = 0; cfg < 3; cfg++) {
        const int Nn = cfg == 0 ? 1024 : cfg == 1 ? 3072 : 4096, K = cfg == 0 ? 4096 : 1024;
        const int blocks = 5 * (Nn / 64);
        hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&st, (size_t)blocks * 4 * NPH * 8);
        hipFuncSetAttribute((const void*)k_step, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE);
        for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), NS * STAGE, 0, A, W, K * 2, 5, out, st);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)blocks * 4 * NPH);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        const char* names[NPH] = {"vmcnt wait", "barrier", "dma issue", "lds reads", "mfma issue", "mfma drain"};
        const int steps = K / 64 - 4;
        printf("N=%d (%d blocks), K=%d: cycles per K-step (s_memtime, 100 MHz? see total), avg over blocks, wave 0 / wave 3\n", Nn, blocks, K);
        double tot0 = 0, tot3 = 0;
        for (int p = 0; p < NPH; p++) {
            double s0 = 0, s3 = 0;
            for (int b = 0; b < blocks; b++) { s0 += h[((size_t)b * 4 + 0) * NPH + p]; s3 += h[((size_t)b * 4 + 3) * NPH + p]; }
            s0 /= (double)blocks * steps; s3 /= (double)blocks * steps;
            tot0 += s0; tot3 += s3;
            printf("  %-12s %8.1f %8.1f\n", names[p], s0, s3);
        }
        printf("  %-12s %8.1f %8.1f\n", "total", tot0, tot3);
        {
            auto go = [&](const char* name, auto kern) {
                hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE);
                for (int it = 0; it < 3; it++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), NS * STAGE, 0, A, W, K * 2, 5, out, st);
                hipDeviceSynchronize();
                std::vector<unsigned long long> hh((size_t)blocks * 4);
                hipMemcpy(hh.data(), st, hh.size() * 8, hipMemcpyDeviceToHost);
                double sum = 0;
                for (auto v : hh) sum += v;
                printf("  %-40s %8.1f cycles per K-step\n", name, sum / hh.size() / (K / 64));
            };
            go("k-split over waves, dma before reads", k_step_v<1>);
            go("k-split over waves, dma after reads", k_step_v<2>);
            go("2x2 waves, software-pipelined", k_step_p<3>);
            go("k-split over waves, software-pipelined", k_step_p<4>);
        }
        hipFree(out); hipFree(st);
    }
    return 0;
}
