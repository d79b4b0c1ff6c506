// Follow-up to xcd_team.hip: what bounds "every XCD streams the same 24 MB per layer"?  Variants: load policy (plain / nt), slice order
// staggered per XCD, number of XCDs that stream (1, 2, 4, 8: per-XCD fabric port vs shared Infinity-Cache rate), unique bytes instead of shared.
// hipcc --offload-arch=gfx950 -O3 -o xcd_stream xcd_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
constexpr int NL = 24;
constexpr size_t LAYER_BYTES = 24u << 20;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// every workgroup of XCD x (blockIdx % 8 == x observed; rank = blockIdx / 8) reads 1/32 of each layer; NT: nontemporal loads; STAG: XCD x starts
// its walk over the layer at slice rotation x*4; UNIQ: XCD x reads its own 1/8 of the layer only (no sharing: the HBM-bound yardstick, 24 MB per layer total)
template <bool NT, bool STAG, bool UNIQ>
__global__ __launch_bounds__(256, 1) void k_stream(const u32x4* __restrict__ W, unsigned* sink, int reps, int nx) {
    extern __shared__ char lds[];
    const int xcc = blockIdx.x & 7, rank = blockIdx.x >> 3;
    if (xcc >= nx) return;
    unsigned acc = 0;
    for (int rep = 0; rep < reps; rep++)
        for (int l = 0; l < NL; l++) {
            // 24 MB = 96 chunks of 256 KB; workgroup takes chunks rank, rank+32, rank+64 (3 x 256 KB = 768 KB)
            for (int c = 0; c < 3; c++) {
                int chunk = rank + 32 * c;
                if (STAG) chunk = (chunk + xcc * 12) % 96;
                if (UNIQ) chunk = (xcc * 12 + (rank + 32 * c) % 12);     // 12 chunks per XCD, each read by 8 workgroups of that XCD... keeps per-CU bytes equal
                const u32x4* p = W + ((size_t)l * LAYER_BYTES + (size_t)chunk * (256 << 10)) / 16;
                u32x4 v[64];
#pragma unroll
                for (int u = 0; u < 64; u++) v[u] = NT ? __builtin_nontemporal_load(p + threadIdx.x + u * 256) : p[threadIdx.x + u * 256];
#pragma unroll
                for (int u = 0; u < 64; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
            }
        }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <bool NT, bool STAG, bool UNIQ>
static void run(const char* name, const u32x4* W, unsigned* sink, int nx) {
    const int reps = 3;
    float best = 1e30f;
    CK(hipFuncSetAttribute((const void*)k_stream<NT, STAG, UNIQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int it = 0; it < 3; it++) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_stream<NT, STAG, UNIQ>), dim3(256), dim3(256), 100 * 1024, 0, W, sink, reps, nx);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    const double us = best * 1e3 / (reps * NL);
    const double l2_bytes = (double)nx * LAYER_BYTES;
    printf("%-52s %2d XCDs %7.2f us/layer  delivered to L2s %6.2f TB/s  per XCD %5.0f GB/s\n", name, nx, us, l2_bytes / us / 1e6, l2_bytes / nx / us / 1e3);
}
int main() {
    u32x4* W; unsigned* sink;
    CK(hipMalloc(&W, NL * LAYER_BYTES)); CK(hipMemset(W, 1, NL * LAYER_BYTES)); CK(hipMalloc(&sink, 16));
    for (int nx : {1, 2, 4, 8}) run<false, false, false>("shared bytes, plain loads", W, sink, nx);
    run<true, false, false>("shared bytes, nt loads", W, sink, 8);
    run<false, true, false>("shared bytes, plain, XCD-staggered order", W, sink, 8);
    run<true, true, false>("shared bytes, nt, XCD-staggered order", W, sink, 8);
    run<false, false, true>("unique bytes per XCD (24 MB per layer in total)", W, sink, 8);
    run<true, false, true>("unique bytes per XCD, nt", W, sink, 8);
    return 0;
}
