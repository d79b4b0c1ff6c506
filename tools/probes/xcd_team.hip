// XCD-team probe (round 5): what does a decode layer cost when each XCD runs the whole layer stack for its own
// rows and hand-offs never leave the XCD?  Three unknowns, measured here before the engine is built:
//   (a) an XCD-local barrier among the 32 workgroups of one XCD (plain stores kept in that XCD's L2, ONE non-sc1 atomic per workgroup
//       executed in the L2, poll by sc1 load or by a returning L2 atomic),
//   (b) the rate at which all 8 XCDs can stream the SAME weight bytes (every XCD needs every weight: 8x the L2 fills, served by the
//       Infinity Cache when the XCDs run in phase),
//   (c) whether a payload written with plain stores is read correctly by the other workgroups of the team through L1-bypassing loads.
// Teams are formed from HW_REG_XCC_ID at run time (never from blockIdx): rank = ticket on a per-XCD counter.
// hipcc --offload-arch=gfx950 -O3 -o xcd_team xcd_team.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int NL = 24, NPH = 5;
constexpr int PH_BYTES[NPH] = {6 << 20, 0, 2 << 20, 8 << 20, 8 << 20};      // c_attn, attention (no weights), attn.c_proj, c_fc, mlp.c_proj (GPT-2-medium, bf16)
constexpr size_t LAYER_BYTES = 24u << 20;
constexpr int PAY = 4096;                                                   // payload bytes per workgroup and phase

struct Ctl {
    unsigned team_cnt[8];
    unsigned arrived;
    unsigned err;
    unsigned bad;
    unsigned pad[21];
    unsigned bar[8 * 32];          // one monotonic counter per XCD, 128 B apart
    unsigned long long t_phase[NPH + 1];      // s_memrealtime ticks (100 MHz) summed by block 0
};

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define RLX_WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP

__device__ __forceinline__ uint4 ld16_sc1(const void* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, RLX_AGENT), b = __hip_atomic_load(q + 1, RLX_AGENT);
    return make_uint4((unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32));
}

// ATOM: 0 = agent-scope atomic (sc1: executes memory-side), 1 = workgroup-scope atomic (no sc bits: executes in this XCD's L2)
// POLL: 0 = sc1 load, 1 = returning L2 atomic (fetch_or 0, workgroup scope)
template <int ATOM, int POLL>
__device__ __forceinline__ bool team_barrier(unsigned* ctr, unsigned target, unsigned* err, int* sflag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (ATOM == 0) __hip_atomic_fetch_add(ctr, 1u, RLX_AGENT); else __hip_atomic_fetch_add(ctr, 1u, RLX_WG);
        int ok = 1;
        unsigned spins = 0;
        for (;;) {
            const unsigned v = POLL == 0 ? __hip_atomic_load(ctr, RLX_AGENT) : __hip_atomic_fetch_or(ctr, 0u, RLX_WG);
            if (v >= target) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) { __hip_atomic_store(err, 1u, RLX_AGENT); ok = 0; break; }
        }
        *sflag = ok;
    }
    __syncthreads();
    return *reinterpret_cast<volatile int*>(sflag) != 0;
}

// MODE bit 0: stream the weights, bit 1: barriers, bit 2: payload exchange + check
template <int MODE, int ATOM, int POLL>
__global__ __launch_bounds__(256, 1) void k_team(const uint4* __restrict__ W, unsigned char* pay, Ctl* ctl, unsigned* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int* sflag = reinterpret_cast<int*>(lds);
    __shared__ int s_rank, s_size, s_xcc;
    if (threadIdx.x == 0) {
        const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;      // HW_REG_XCC_ID[3:0]
        s_xcc = xcc;
        s_rank = (int)__hip_atomic_fetch_add(&ctl->team_cnt[xcc], 1u, RLX_AGENT);
        __hip_atomic_fetch_add(&ctl->arrived, 1u, RLX_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->arrived, RLX_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { __hip_atomic_store(&ctl->err, 2u, RLX_AGENT); break; }
        }
        s_size = (int)__hip_atomic_load(&ctl->team_cnt[xcc], RLX_AGENT);
    }
    __syncthreads();
    const int rank = s_rank, tsize = s_size, xcc = s_xcc;
    unsigned* ctr = ctl->bar + xcc * 32;
    unsigned acc = 0, epoch = 0, bad = 0;
    unsigned long long tph[NPH + 1] = {0, 0, 0, 0, 0, 0};
    for (int rep = 0; rep < reps; rep++)
        for (int l = 0; l < NL; l++) {
            size_t off = (size_t)l * LAYER_BYTES;
            for (int ph = 0; ph < NPH; ph++) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                if (MODE & 1) {
                    // this workgroup's slice of the phase's weights: every load issued before anything is consumed
                    const int per = PH_BYTES[ph] / tsize;                  // bytes per workgroup
                    const uint4* p = W + (off + (size_t)rank * per) / 16;
                    const int n16 = per / 16;
                    uint4 v[64];
#pragma unroll
                    for (int u = 0; u < 64; u++) {
                        const int i = threadIdx.x + u * 256;
                        v[u] = i < n16 ? p[i] : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 64; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
                }
                off += PH_BYTES[ph];
                const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
                if (MODE & 4) {
                    // publish 4 KB: plain 16-B stores (stay in this XCD's L2)
                    uint4* dst = reinterpret_cast<uint4*>(pay + ((size_t)(xcc * 32 + rank) * 2 + (epoch & 1)) * PAY);
                    const unsigned tag = (epoch << 8) | (unsigned)rank;
                    dst[threadIdx.x] = make_uint4(tag, tag ^ threadIdx.x, tag + 1, tag + 2);
                }
                if (MODE & 2) {
                    epoch++;
                    if (!team_barrier<ATOM, POLL>(ctr, epoch * (unsigned)tsize, &ctl->err, sflag)) return;
                }
                if (MODE & 4) {
                    // read two other members' payloads with L1-bypassing loads and check every word
                    for (int k = 1; k <= 2; k++) {
                        const int other = (rank + k * 7) % tsize;
                        const uint4* src = reinterpret_cast<const uint4*>(pay + ((size_t)(xcc * 32 + other) * 2 + ((epoch - 1) & 1)) * PAY);
                        const uint4 r = ld16_sc1(src + threadIdx.x);
                        const unsigned tag = ((epoch - 1) << 8) | (unsigned)other;
                        if (r.x != tag || r.y != (tag ^ threadIdx.x) || r.z != tag + 1 || r.w != tag + 2) bad++;
                    }
                }
                const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
                tph[ph] += t1 - t0;
                tph[NPH] += t2 - t1;
            }
        }
    if (bad) atomicAdd(&ctl->bad, bad);
    if (acc == 0x12345678u) sink[0] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i <= NPH; i++) ctl->t_phase[i] = tph[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE, int ATOM, int POLL>
static void run(const char* name, const uint4* W, unsigned char* pay, Ctl* ctl, unsigned* sink) {
    const int reps = 3;
    float best = 1e30f;
    CK(hipFuncSetAttribute((const void*)k_team<MODE, ATOM, POLL>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    Ctl h;
    for (int it = 0; it < 3; it++) {
        CK(hipMemset(ctl, 0, sizeof(Ctl)));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_team<MODE, ATOM, POLL>), dim3(256), dim3(256), 100 * 1024, 0, W, pay, ctl, sink, reps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    }
    const double per_layer = best * 1e3 / (reps * NL);
    printf("%-44s %8.2f us/layer  err=%u bad=%u teams=[", name, per_layer, h.err, h.bad);
    for (int i = 0; i < 8; i++) printf("%u%s", h.team_cnt[i], i == 7 ? "" : ",");
    printf("]  block0 per layer: stream");
    for (int i = 0; i < NPH; i++) printf(" %.2f", h.t_phase[i] * 0.01 / (reps * NL));
    printf(" | sync+exchange %.2f us\n", h.t_phase[NPH] * 0.01 / (reps * NL));
}

int main() {
    uint4* W;
    unsigned char* pay;
    Ctl* ctl;
    unsigned* sink;
    CK(hipMalloc(&W, NL * LAYER_BYTES));
    CK(hipMemset(W, 1, NL * LAYER_BYTES));
    CK(hipMalloc(&pay, 8 * 32 * 2 * PAY));
    CK(hipMemset(pay, 0, 8 * 32 * 2 * PAY));
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMalloc(&sink, 16));
    printf("per layer: 5 phases (c_attn 6 MB, attention 0, c_proj 2 MB, c_fc 8 MB, mlp.c_proj 8 MB), every XCD streams all 24 MB; 256 workgroups, 1 per CU\n");
    run<2, 1, 0>("barriers only (L2 atomic, sc1 poll)", W, pay, ctl, sink);
    run<2, 1, 1>("barriers only (L2 atomic, L2-atomic poll)", W, pay, ctl, sink);
    run<2, 0, 0>("barriers only (agent atomic, sc1 poll)", W, pay, ctl, sink);
    run<1, 1, 0>("stream only", W, pay, ctl, sink);
    run<3, 1, 0>("stream + barriers (L2 atomic, sc1 poll)", W, pay, ctl, sink);
    run<3, 1, 1>("stream + barriers (L2 atomic, atomic poll)", W, pay, ctl, sink);
    run<3, 0, 0>("stream + barriers (agent atomic, sc1 poll)", W, pay, ctl, sink);
    run<7, 1, 0>("stream + barriers + payload (L2, sc1 poll)", W, pay, ctl, sink);
    run<7, 1, 1>("stream + barriers + payload (L2, atomic poll)", W, pay, ctl, sink);
    run<6, 1, 0>("barriers + payload, no stream (L2, sc1 poll)", W, pay, ctl, sink);
    return 0;
}
