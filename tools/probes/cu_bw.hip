// Per-CU L2 -> CU bandwidth probe: plain 16-B loads to VGPRs vs global_load_lds (LDS-DMA), 1 or 2 blocks of 256 threads per CU.
// hipcc --offload-arch=gfx950 -O3 -o cu_bw cu_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int REGION = 96 * 1024;   // bytes per block, L2-resident across repeats

template <int UNROLL>
__global__ __launch_bounds__(256) void k_plain(const uint4* __restrict__ buf, int reps, unsigned* out) {
    const uint4* p = buf + (size_t)blockIdx.x * (REGION / 16);
    unsigned acc = 0;
    for (int r = 0; r < reps; r++)
        for (int i = threadIdx.x; i < REGION / 16; i += 256 * UNROLL) {
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) { const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + i + u * 256)); v[u] = make_uint4(t.x, t.y, t.z, t.w); }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_plain_cached(const uint4* __restrict__ buf, int reps, unsigned* out) {
    const uint4* p = buf + (size_t)blockIdx.x * (REGION / 16);
    unsigned acc = 0;
    for (int r = 0; r < reps; r++)
        for (int i = threadIdx.x; i < REGION / 16; i += 256 * UNROLL) {
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) v[u] = p[i + u * 256];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_dma(const uint4* __restrict__ buf, int reps, unsigned* out) {
    __shared__ __attribute__((aligned(1024))) char lds[4 * UNROLL * 1024 * 2];
    const uint4* p = buf + (size_t)blockIdx.x * (REGION / 16);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ring = 0;
    for (int r = 0; r < reps; r++)
        for (int i = threadIdx.x; i < REGION / 16; i += 256 * UNROLL) {
            char* dst = lds + ring * (4 * UNROLL * 1024) + wave * (UNROLL * 1024);
#pragma unroll
            for (int u = 0; u < UNROLL; u++) __builtin_amdgcn_global_load_lds((gptr_t)(p + i + u * 256), (lptr_t)(dst + u * 1024), 16, 0, 0);
            ring ^= 1;
            if (UNROLL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (UNROLL == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (reinterpret_cast<unsigned*>(lds)[lane] == 0x12345678u) out[0] = 1;
}
// GEMM-shaped DMA: block (tm, tn) streams A rows [64 tm, +64) and W rows [64 tn, +64) of row-major [rows][K] bf16 matrices, one
// 64-k (128-B) column block per step, 8 rows x 128 B per wave instruction, NS-1 steps in flight (the gemm_nt_s64_kernel pattern)
template <int NS, bool SWZ>
__global__ __launch_bounds__(256) void k_dma_gemm(const char* __restrict__ A, const char* __restrict__ W, int K2, int tiles_m, int reps, unsigned* out) {
    __shared__ __attribute__((aligned(1024))) char lds[NS * 16384];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int nk = K2 / 128;
    for (int r = 0; r < reps; r++) {
        int slot = 0;
        for (int kt = 0; kt < nk; kt++) {
            char* st = lds + slot * 16384;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int seg = wave + 4 * i;                 // 0..7 -> A, 8..15 -> W
                const int row = (seg & 7) * 8 + (lane >> 3);
                int chunk = lane & 7;
                if (SWZ) chunk ^= (row >> 1) & 7;
                const char* src = (seg < 8 ? A + (size_t)(tm * 64 + row) * K2 : W + (size_t)(tn * 64 + row) * K2) + kt * 128 + chunk * 16;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + seg * 1024), 16, 0, 0);
            }
            slot = slot + 1 == NS ? 0 : slot + 1;
            if (NS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (NS == 8) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (reinterpret_cast<unsigned*>(lds)[lane] == 0x12345678u) out[0] = 1;
}
template <class F>
static void run(const char* name, F launch, int blocks, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(blocks, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch(blocks, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * REGION * reps;
    printf("%-28s blocks %4d: %8.3f ms  %7.2f TB/s total  %6.1f GB/s per CU\n", name, blocks, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}
int main() {
    const size_t n = (size_t)512 * REGION;
    uint4* buf; unsigned* out;
    hipMalloc(&buf, n); hipMalloc(&out, 4);
    hipMemset(buf, 1, n);
    const int reps = 200;
    for (int blocks : {64, 256, 512}) {
        run("plain nt x4", [&](int g, int r) { hipLaunchKernelGGL(k_plain<4>, dim3(g), dim3(256), 0, 0, buf, r, out); }, blocks, reps);
        run("plain nt x8", [&](int g, int r) { hipLaunchKernelGGL(k_plain<8>, dim3(g), dim3(256), 0, 0, buf, r, out); }, blocks, reps);
        run("plain cached x8", [&](int g, int r) { hipLaunchKernelGGL(k_plain_cached<8>, dim3(g), dim3(256), 0, 0, buf, r, out); }, blocks, reps);
        run("lds-dma x2", [&](int g, int r) { hipLaunchKernelGGL(k_dma<2>, dim3(g), dim3(256), 0, 0, buf, r, out); }, blocks, reps);
        run("lds-dma x4", [&](int g, int r) { hipLaunchKernelGGL(k_dma<4>, dim3(g), dim3(256), 0, 0, buf, r, out); }, blocks, reps);
        run("lds-dma x8", [&](int g, int r) { hipLaunchKernelGGL(k_dma<8>, dim3(g), dim3(256), 0, 0, buf, r, out); }, blocks, reps);
    }
    for (int K : {768, 1024, 1040, 4096}) {
        for (int N : {1024, 3072}) {
            const int tiles = 5 * (N / 64), reps = 20;
            auto go = [&](const char* name, auto kern) {
                hipEvent_t a, b;
                hipEventCreate(&a); hipEventCreate(&b);
                hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 0, 0, (const char*)buf, (const char*)buf + (8 << 20), K * 2, 5, 2, out);
                hipDeviceSynchronize();
                hipEventRecord(a);
                hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 0, 0, (const char*)buf, (const char*)buf + (8 << 20), K * 2, 5, reps, out);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                const double per_step_us = ms * 1e3 / reps / (K / 64);
                printf("gemm-dma %-10s K %5d N %5d blocks %4d: %6.3f us per K-step  %6.1f GB/s per block\n", name, K, N, tiles, per_step_us, 16384.0 / per_step_us / 1e3);
            };
            go("ns4", k_dma_gemm<4, false>);
            go("ns4 swz", k_dma_gemm<4, true>);
            go("ns8", k_dma_gemm<8, false>);
        }
    }
    return 0;
}
