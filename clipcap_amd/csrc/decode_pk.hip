// Persistent form of one generated position's GPT-2 layer stack (KV-cached decode, Tn == 1): ONE launch runs all n_layer blocks —
// c_attn, beam-group attention, attn.c_proj (+ ln_2), c_fc (gelu), mlp.c_proj (+ the next ln_1 / ln_f) — that decode.hip otherwise
// issues as 7 launches per layer (reference: the whole re-forward of inference/base.py:80-121 per generated token).
//
// Structure (MI355X: 256 CUs, 8 XCDs with private L2s; /opt/skills/guides MI355X_MICROARCH.md "price list"):
//   * grid = one 512-thread workgroup per CU (96 KiB of LDS per workgroup keeps a second one off the CU), all resident; a workgroup walks
//     the phases of every layer in order and takes the tasks of a phase that its index names (static schedule, XCD-aware: the row tiles
//     that share a weight slice sit on one XCD's L2).
//   * a task's dependencies are the tasks of the PREVIOUS phase on the same 64-row tile (everything in a GPT-2 block is row-local
//     except attention, which is local to a beam group): per (phase, row tile) arrival counters in global memory.  Producer: activation
//     stores are write-through (sc1), every storing wave drains (vmcnt(0)), workgroup barrier, ONE relaxed agent-scope atomic add.
//     Consumer: ONE lane polls the counter (relaxed sc1 load + s_sleep), then activation loads bypass L1 (sc1 — both the LDS-DMA loads of
//     the GEMM A tiles and the vector loads).  No fences (the guide's R1 form); weights and the old KV rows are plain loads.
//   * a GEMM task requests its first weight tiles (they depend on nothing) BEFORE it polls, so the dependency wait hides their latency.
//   * every spin is bounded; a timeout raises an error word that makes every workgroup leave (cc_decode_ws_check reads it).
//
// GEMM tasks: 64 x 64 tiles, 8 waves = two groups that split each 64-k K-tile (gemm.hip.h::gemm_nt_s64_kernel, KG = 2), 6-stage
// global_load_lds pipeline with counted vmcnt.  c_proj GEMMs split K over tasks into fp32 slabs; a finish phase (one wave per row) sums
// them, adds bias + residual and applies the following LayerNorm.  Attention: decode.hip::k_decode_attn_group, two (group, head) items per
// workgroup (heads 2j, 2j + 1 of one group: same entry list, so the two 4-wave teams run in lockstep).
//
// bf16 / fp16 builds only (the split-bf16 build keeps the launch-per-op path: its operand images are built between the GEMMs).
#include "../../include/clipcap_hip.h"
#include "gemm.hip.h"
#include "kernels.h"
#include "decode_pk.h"

namespace CC_NS {
#if CC_OP != 2 && defined(CC_EXPERIMENTS)      // lab build only (make lab): measured 2x slower than the per-op launches
namespace {

constexpr int PK_THREADS = 512, PK_NS = 6, PK_STAGE = 16384, PK_LDS = PK_NS * PK_STAGE;     // + 64 B behind it for the wait flag
constexpr unsigned PK_SPIN_MAX = 1u << 21;          // polls (~1 us each with the sleep) before a wait gives up
enum { C1 = 0, C2, C3, C3F, C4, C5, C5F, C_PH };

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define PK_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// ---- write-through stores / L1-bypassing loads of activations that another workgroup of this launch produces or consumes
// (s_nop 1 behind an inline-asm store: its data VGPRs are still being read when the next instruction issues, and hipcc's hazard
// recogniser cannot see into the asm — without it the first 8 bytes of some lanes' 16-B stores were garbage)
__device__ __forceinline__ void st16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8_sc1(void* p, u32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void stf4_sc1(float* p, float a, float b, float c, float d) {
    u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
    st16_sc1(p, v);
}
__device__ __forceinline__ void st_act8_sc1(act_t* p, const float (&f)[8]) {
    const uint4 r = pack8(f);
    u32x4 v = {r.x, r.y, r.z, r.w};
    st16_sc1(p, v);
}
// 16 B as two 8-B agent-scope relaxed loads (global_load_dwordx2 sc1, waits tracked by the compiler)
__device__ __forceinline__ uint4 ld16_sc1(const void* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, PK_RLX), b = __hip_atomic_load(q + 1, PK_RLX);
    return make_uint4((unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32));
}
__device__ __forceinline__ float4 ldf4_sc1(const float* p) {
    const uint4 r = ld16_sc1(p);
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}

__device__ __forceinline__ void pk_lgkm0() { __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8)); asm volatile("" ::: "memory"); }
__device__ __forceinline__ void pk_vm(int n) {      // s_waitcnt vmcnt(n), n in 0..10 (wave-uniform)
    switch (n) {
        case 0: s_wait_vm<0>(); break;
        case 1: s_wait_vm<1>(); break;
        case 2: s_wait_vm<2>(); break;
        case 3: s_wait_vm<3>(); break;
        case 4: s_wait_vm<4>(); break;
        case 5: s_wait_vm<5>(); break;
        case 6: s_wait_vm<6>(); break;
        case 7: s_wait_vm<7>(); break;
        case 8: s_wait_vm<8>(); break;
        case 9: s_wait_vm<9>(); break;
        default: s_wait_vm<10>(); break;
    }
}

struct PkDep { unsigned* c; int r0, r1; unsigned target; };       // wait until c[r] >= target for r in [r0, r1]

// Block-wide wait for a dependency.  Leaves the DMA queue alone (raw barrier: prefetched weight tiles stay in flight).  false = give up.
__device__ __forceinline__ bool pk_wait(const PkDep& d, unsigned* err, int* sflag) {
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int r = d.r0; r <= d.r1 && ok; r++) {
            unsigned spins = 0;
            while (__hip_atomic_load(d.c + r, PK_RLX) < d.target) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 63u) == 0 && (spins > PK_SPIN_MAX || __hip_atomic_load(err, PK_RLX) != 0u)) {
                    __hip_atomic_store(err, 1u, PK_RLX);
                    ok = 0;
                    break;
                }
            }
        }
        *sflag = ok;
        *reinterpret_cast<unsigned long long*>(sflag + 2) += __builtin_amdgcn_s_memrealtime() - t0;      // time spent polling (profile)
    }
    pk_lgkm0();
    __builtin_amdgcn_s_barrier();
    const int ok = *reinterpret_cast<volatile int*>(sflag);
    pk_lgkm0();
#ifdef PK_FENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    return ok != 0;
}
// Every wave's write-through stores have left, then one arrival per row tile.
__device__ __forceinline__ void pk_signal(unsigned* c, int r0, int r1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PK_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0)
        for (int r = r0; r <= r1; r++) __hip_atomic_fetch_add(c + r, 1u, PK_RLX);
}

// ---- epilogues of the GEMM tasks: (row, 8 consecutive columns)
struct PkEpiAct {            // bias (+ gelu_new) -> 16-bit activation
    act_t* C;
    const float* bias;
    int ldc, M, act;
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8]) const {
        if (row >= M) return;
        const float4 b0 = *reinterpret_cast<const float4*>(bias + col), b1 = *reinterpret_cast<const float4*>(bias + col + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        if (act == 2) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = gelu_new_f(v[e]);
        }
        st_act8_sc1(C + (size_t)row * ldc + col, v);
    }
};
struct PkEpiSlab {           // fp32 partial of one K slice
    float* S;
    int ld, M;
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8]) const {
        if (row >= M) return;
        float* p = S + (size_t)row * ld + col;
        stf4_sc1(p, v[0], v[1], v[2], v[3]);
        stf4_sc1(p + 4, v[4], v[5], v[6], v[7]);
    }
};

// One 64 x 64 output tile, K-tiles [kt0, kt0 + nk) of 64: A = activations [M][lda] (produced in this launch: sc1 DMA), B = weights
// [N][ldb] (K-contiguous; N, K multiples of 64).  The weight tiles of the pipeline's first stages are requested before the dependency
// wait.  LDS: `lds` = PK_NS stages of 16 KiB ([64 A rows | 64 B rows] x 128 B, XOR-swizzled chunks), reused for the epilogue.
template <class Epi>
__device__ __forceinline__ bool pk_gemm_tile(const op16_t* __restrict__ A, int lda, int M, int m0, const op16_t* __restrict__ B, int ldb, int n0,
                                             int kt0, int nk, char* lds, const Epi& epi, const PkDep& dep, unsigned* err, int* sflag) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    // this wave's DMA piece of a stage: 8 rows x 128 B of A and of B
    const int drow = wave * 8 + (lane >> 3);
    const int dchunk = ((lane & 7) ^ ((drow >> 1) & 7) ^ ((drow >> 4) & 3)) * 8;
    const op16_t* asrc = A + (size_t)min(m0 + drow, M - 1) * lda + (size_t)kt0 * 64 + dchunk;
    const op16_t* bsrc = B + (size_t)(n0 + drow) * ldb + (size_t)kt0 * 64 + dchunk;
    const int drow0 = lane >> 3;                       // wave 0's rows (wave 1 requests them in the prologue, see below)
    const op16_t* b0src = B + (size_t)(n0 + drow0) * ldb + (size_t)kt0 * 64 + ((lane & 7) ^ ((drow0 >> 1) & 7)) * 8;
    char* adst = lds + wave * 1024;
    char* bdst = lds + 8192 + wave * 1024;
#define PK_ISSUE_A(T, SLOT) __builtin_amdgcn_global_load_lds((gptr_t)(asrc + (size_t)(T)*64), (lptr_t)(adst + (SLOT)*PK_STAGE), 16, 0, 16)
#define PK_ISSUE_B(T, SLOT) __builtin_amdgcn_global_load_lds((gptr_t)(bsrc + (size_t)(T)*64), (lptr_t)(bdst + (SLOT)*PK_STAGE), 16, 0, 0)
    // wave 0 polls (its poll loads would queue behind its own prefetch: in-order vmcnt), wave 1 requests wave 0's B pieces as well
    const int pro = min(nk, PK_NS - 1);
    if (wave != 0)
        for (int t = 0; t < pro; t++) {
            PK_ISSUE_B(t, t);
            if (wave == 1) __builtin_amdgcn_global_load_lds((gptr_t)(b0src + (size_t)t * 64), (lptr_t)(bdst - 1024 + t * PK_STAGE), 16, 0, 0);
        }
    if (!pk_wait(dep, err, sflag)) return false;
    for (int t = 0; t < pro; t++) PK_ISSUE_A(t, t);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int steady = max(0, nk - (PK_NS - 1));       // iterations that issue a new (A, B) pair
    int slot = 0, islot = PK_NS - 1;
    for (int kt = 0; kt < nk; kt++) {
        // loads of this wave issued after the last one stage kt needs (in-order completion): they may stay in flight
        const int after = kt < pro ? (pro - 1 - kt) + 2 * min(kt, steady) : 2 * (min(nk - 1, kt + PK_NS - 2) - kt);
        pk_vm(after);
        __builtin_amdgcn_s_barrier();
        if (kt + PK_NS - 1 < nk) { PK_ISSUE_A(kt + PK_NS - 1, islot); PK_ISSUE_B(kt + PK_NS - 1, islot); }      // that slot held stage kt-1
        const char* cur = lds + slot * PK_STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int ch = (grp * 2 + kk) * 2 + fhalf;
            const op16x8 a = *reinterpret_cast<const op16x8*>(cur + g_lds_off(wm * 32 + frow, ch));
            const op16x8 b = *reinterpret_cast<const op16x8*>(cur + 8192 + g_lds_off(wn * 32 + frow, ch));
            acc = CC_MFMA_32x32x16(a, b, acc);
        }
        slot = slot + 1 == PK_NS ? 0 : slot + 1;
        islot = islot + 1 == PK_NS ? 0 : islot + 1;
    }
#undef PK_ISSUE_A
#undef PK_ISSUE_B
    __syncthreads();                                   // no DMA outstanding; every wave is past its fragment reads
    // the two K groups swap halves: group 1 hands over registers 0-7 (tile rows 0-15), group 0 registers 8-15
    float* xch = reinterpret_cast<float*>(lds);        // [2][8][256]
    {
        float* xo = xch + grp * 2048 + (tid & 255);
#pragma unroll
        for (int r = 0; r < 8; r++) xo[r * 256] = grp ? acc[r] : acc[8 + r];
    }
    __syncthreads();
    float fin[8];
    {
        const float* xi = xch + (grp ^ 1) * 2048 + (tid & 255);
#pragma unroll
        for (int r = 0; r < 8; r++) fin[r] = (grp ? acc[8 + r] : acc[r]) + xi[r * 256];
    }
    // accumulator (row = 8 (r >> 2) + 4 (lane >> 5) + (r & 3), col = lane & 31) -> wave-private strip -> 8 consecutive columns per lane
    constexpr int SLD = 36;
    float* strip = reinterpret_cast<float*>(lds + 16384) + wave * (16 * SLD);      // this wave's 16 rows x 32 columns
#pragma unroll
    for (int rr = 0; rr < 8; rr++) strip[((rr >> 2) * 8 + fhalf * 4 + (rr & 3)) * SLD + frow] = fin[rr];
    {
        const int lr = lane >> 2, c8 = lane & 3;
        const float4 x = *reinterpret_cast<const float4*>(strip + lr * SLD + c8 * 8), y = *reinterpret_cast<const float4*>(strip + lr * SLD + c8 * 8 + 4);
        float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
        epi(m0 + wm * 32 + grp * 16 + lr, n0 + wn * 32 + c8 * 8, v);
    }
    return true;
}

// finish of a split-K c_proj for 8 rows (one wave each): out = sum of slabs + bias + res; y16 = LayerNorm(out) * gamma + beta
__device__ __forceinline__ void pk_finish_rows(const float* slabs, size_t slab_elems, int ks, int row, int M, int D, const float* bias, const float* res,
                                               float* out32, const float* gamma, const float* beta, act_t* y16) {
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    constexpr int MAXV = 8;                // D <= 2048
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXV; it++) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            const float* sp = slabs + (size_t)row * D + c;
            float4 a = ldf4_sc1(sp);
            for (int z = 1; z < ks; z++) {
                const float4 p = ldf4_sc1(sp + z * slab_elems);
                a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
            }
            const float4 bb = *reinterpret_cast<const float4*>(bias + c);
            const float4 rr = ldf4_sc1(res + (size_t)row * D + c);
            a.x += bb.x + rr.x; a.y += bb.y + rr.y; a.z += bb.z + rr.z; a.w += bb.w + rr.w;
            stf4_sc1(out32 + (size_t)row * D + c, a.x, a.y, a.z, a.w);
            v[it] = a;
            s += a.x + a.y + a.z + a.w;
        }
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < MAXV; it++) {
        const int c = lane * 4 + it * 256;
        if (c < D) { const float a = v[it].x - mu, b = v[it].y - mu, c2 = v[it].z - mu, d = v[it].w - mu; q += a * a + b * b + c2 * c2 + d * d; }
    }
    const float rs = rsqrtf(wave_sum(q) / D + 1e-5f);
#pragma unroll
    for (int it = 0; it < MAXV; it++) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
            u32x2 pk = {pack2op((v[it].x - mu) * rs * g.x + b.x, (v[it].y - mu) * rs * g.y + b.y),
                        pack2op((v[it].z - mu) * rs * g.z + b.z, (v[it].w - mu) * rs * g.w + b.w)};
            st8_sc1(y16 + (size_t)row * D + c, pk);
        }
    }
}

// Beam-group attention for ONE (group, head) by a 4-wave team (decode.hip::k_decode_attn_group with sc1 loads of what this launch
// produced — q and the new K / V in qkv — and write-through output; both teams of the workgroup call it in lockstep, the barriers are
// workgroup barriers).  LDS per team: p[cap][8] | wmx[4][8] | wsum[8][8] | red2[8][G][64].
template <int G>
__device__ __forceinline__ void pk_attn_team(const act_t* __restrict__ qkv, act_t* __restrict__ kc, act_t* __restrict__ vc, const int2* __restrict__ ent,
                                             int nU, act_t* __restrict__ out, int s, int h, int H, int pos0, int ctx_max, float scale, int cap,
                                             float* tsm, int ttid) {
    constexpr int HD = 64, KPW = 32;
    float* p = tsm;
    float* wmx = p + (size_t)cap * 8;
    float* wsm = wmx + 32;
    float* red2 = wsm + 64;
    const int lane = ttid & 63;
    const int w = __builtin_amdgcn_readfirstlane(ttid >> 6);
    const int r0 = s * G, D = H * HD;
    const act_t* kb = kc + h * HD;
    const act_t* vb = vc + h * HD;
    const act_t* qrow = qkv + (size_t)r0 * 3 * D + h * HD;
    const int skg = lane >> 3, sdc = lane & 7;
    const int hf = lane >> 5, dp = lane & 31;
    int2 ek[4];
#pragma unroll
    for (int i = 0; i < 4; i++) ek[i] = ent[w * KPW + i * 8 + skg];
    int2 ev = ent[w * KPW + (lane & 31)];
    if (ttid < G * 16) {                                   // append this head's new K / V to the cache (read by the NEXT launch: plain stores)
        const int b = ttid >> 4, wq = ttid & 15, which = wq >> 3, c = wq & 7;
        const uint4 v = ld16_sc1(qrow + (size_t)b * 3 * D + (which + 1) * D + c * 8);
        *reinterpret_cast<uint4*>((which ? vc : kc) + ((size_t)(r0 + b) * ctx_max + pos0) * D + h * HD + c * 8) = v;
    }
    float qf[G][8], mx[G];
#pragma unroll
    for (int b = 0; b < G; b++) { unpack8(ld16_sc1(qrow + (size_t)b * 3 * D + sdc * 8), qf[b]); mx[b] = -INFINITY; }
    const int npass = (nU + 4 * KPW - 1) / (4 * KPW);
    unsigned vreg[KPW / 2];
#define PK_VLOAD()                                                                                                                            \
    {                                                                                                                                         \
        const act_t* vrow = ev.x < 0 ? qrow + (size_t)(-1 - ev.x) * 3 * D + 2 * D : vb + (size_t)ev.x * D;                                    \
        const unsigned long long va = reinterpret_cast<unsigned long long>(vrow);                                                             \
        const int valo = (int)(unsigned)va, vahi = (int)(unsigned)(va >> 32);                                                                 \
        _Pragma("unroll") for (int k = 0; k < KPW / 2; k++) {                                                                                 \
            const unsigned lo0 = __builtin_amdgcn_readlane(valo, 2 * k), hi0 = __builtin_amdgcn_readlane(vahi, 2 * k);                        \
            const unsigned lo1 = __builtin_amdgcn_readlane(valo, 2 * k + 1), hi1 = __builtin_amdgcn_readlane(vahi, 2 * k + 1);                \
            const unsigned long long a = ((unsigned long long)(hf ? hi1 : hi0) << 32) | (hf ? lo1 : lo0);                                     \
            vreg[k] = __hip_atomic_load(reinterpret_cast<const unsigned*>(a) + dp, PK_RLX);                                                   \
        }                                                                                                                                     \
    }
    for (int pass = 0; pass < npass; pass++) {
        const int ub = pass * 4 * KPW + w * KPW;
        if (pass > 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) ek[i] = ent[ub + i * 8 + skg];
        }
        uint4 kv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const act_t* krow = ek[i].x < 0 ? qrow + (size_t)(-1 - ek[i].x) * 3 * D + D : kb + (size_t)ek[i].x * D;
            kv[i] = ld16_sc1(krow + sdc * 8);
        }
        if (pass == 0) PK_VLOAD()
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float kf[8], sc[8];
            unpack8(kv[i], kf);
#pragma unroll
            for (int b = 0; b < G; b++) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) a += qf[b][e] * kf[e];
                sc[b] = a;
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
#pragma unroll
                for (int b = 0; b < G; b++) sc[b] += __shfl_xor(sc[b], o);
            }
#pragma unroll
            for (int b = 0; b < 8; b++) {
                sc[b] = (b < G && ((ek[i].y >> b) & 1)) ? sc[b] * scale : -INFINITY;
                if (b < G) mx[b] = fmaxf(mx[b], sc[b]);
            }
            if (sdc == 0) {
                float* pu = p + (size_t)(ub + i * 8 + skg) * 8;
                *reinterpret_cast<float4*>(pu) = make_float4(sc[0], sc[1], sc[2], sc[3]);
                *reinterpret_cast<float4*>(pu + 4) = make_float4(sc[4], sc[5], sc[6], sc[7]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < G; b++) mx[b] = wave_max(mx[b]);
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < G; b++) wmx[w * 8 + b] = mx[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < G; b++) mx[b] = fmaxf(fmaxf(wmx[b], wmx[8 + b]), fmaxf(wmx[16 + b], wmx[24 + b]));
    float acc[G][2], lsum[G];
#pragma unroll
    for (int b = 0; b < G; b++) { acc[b][0] = acc[b][1] = 0.f; lsum[b] = 0.f; }
    for (int pass = 0; pass < npass; pass++) {
        const int ub = pass * 4 * KPW + w * KPW;
#pragma unroll
        for (int jj = 0; jj < KPW * 8 / 64; jj++) {
            const int idx = jj * 64 + lane, b = idx & 7;
            if (b < G) {
                float m = mx[0];
#pragma unroll
                for (int b2 = 1; b2 < G; b2++) m = b == b2 ? mx[b2] : m;
                p[(size_t)ub * 8 + idx] = __expf(p[(size_t)ub * 8 + idx] - m);
            }
        }
        if (pass > 0) {
            ev = ent[ub + (lane & 31)];
            PK_VLOAD()
        }
#pragma unroll
        for (int k = 0; k < KPW / 2; k++) {
            float v0, v1;
            unpack2(vreg[k], v0, v1);
            const float* pu = p + (size_t)(ub + 2 * k + hf) * 8;
            const float4 pa = *reinterpret_cast<const float4*>(pu), pb = *reinterpret_cast<const float4*>(pu + 4);
            const float pj[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
            for (int b = 0; b < G; b++) { acc[b][0] += pj[b] * v0; acc[b][1] += pj[b] * v1; lsum[b] += pj[b]; }
        }
    }
#undef PK_VLOAD
#pragma unroll
    for (int b = 0; b < G; b++) *reinterpret_cast<float2*>(red2 + ((w * 2 + hf) * G + b) * HD + 2 * dp) = make_float2(acc[b][0], acc[b][1]);
    if (dp == 0) {
#pragma unroll
        for (int b = 0; b < G; b++) wsm[(w * 2 + hf) * 8 + b] = lsum[b];
    }
    __syncthreads();
    if (ttid < G * 8) {                                    // 8 consecutive outputs per thread: one 16-B write-through store
        const int b = ttid >> 3, d0 = (ttid & 7) * 8;
        float sum = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int x = 0; x < 8; x++) {
            sum += wsm[x * 8 + b];
            const float4 a = *reinterpret_cast<const float4*>(red2 + (x * G + b) * HD + d0), c = *reinterpret_cast<const float4*>(red2 + (x * G + b) * HD + d0 + 4);
            o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w; o[4] += c.x; o[5] += c.y; o[6] += c.z; o[7] += c.w;
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] *= inv;
        st_act8_sc1(out + (size_t)(r0 + b) * D + h * HD + d0, o);
    }
}

// XCD-aware static schedule of a GEMM phase: `units` distinct weight slices (column tile x K slice), `nrt` row tiles each.  Unit u lives on
// XCD u % 8 (workgroup b runs on XCD b % 8 — a speed assumption only), so the nrt tasks that stream the same weights share one L2.
// Workgroup b = (x = b % 8, q = b / 8) takes the tasks i = q, q + per, ... of its XCD's list; i -> (unit = (i / nrt) * 8 + x, rt = i % nrt).
struct PkSched {
    int x, q, per, nrt, units;
    __device__ __forceinline__ PkSched(int nb, int nrt_, int units_) : x(blockIdx.x & 7), q(blockIdx.x >> 3), per(nb >> 3), nrt(nrt_), units(units_) {}
    __device__ __forceinline__ bool get(int i, int& unit, int& rt) const {
        unit = (i / nrt) * 8 + x;
        rt = i % nrt;
        return unit < units;
    }
    __device__ __forceinline__ int count() const { return ((units - x + 7) >> 3) * nrt; }     // tasks of this XCD
};

template <int G>
__global__ __launch_bounds__(PK_THREADS, 2) void k_decode_layers(PkArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char pk_sm[];
    int* const sflagp = reinterpret_cast<int*>(pk_sm + PK_LDS);      // [0] wait flag, [2..3] poll-time accumulator, then the per-phase profile
    unsigned long long* const pwait = reinterpret_cast<unsigned long long*>(sflagp + 2);
    unsigned long long* const pprof = reinterpret_cast<unsigned long long*>(sflagp + 4);   // [7 phases][wait, total, tasks]
    if (threadIdx.x == 0) { *pwait = 0; for (int i = 0; i < 21; i++) pprof[i] = 0; }
    __syncthreads();
    unsigned long long pk_t0 = 0;
#define PK_T0() if (a.prof && threadIdx.x == 0) { pk_t0 = __builtin_amdgcn_s_memrealtime(); *pwait = 0; }
#define PK_T1(PH) if (a.prof && threadIdx.x == 0) { pprof[(PH)*3] += *pwait; pprof[(PH)*3 + 1] += __builtin_amdgcn_s_memrealtime() - pk_t0; pprof[(PH)*3 + 2] += 1; }      // ALL LDS in the one dynamic array (a second __shared__ object makes hipcc drain the DMA queue before LDS reads)
    const int tid = threadIdx.x, nb = gridDim.x;
    const int D = a.D, H = a.H, M = a.M, nrt = a.nrt;
    unsigned* err = a.ctr + C_PH * PK_MAX_RT;
    const int tn_d = D / 64, tn_3d = 3 * D / 64, tn_4d = 4 * D / 64;
    const int kt_d = D / 64, kt_4d = 4 * D / 64;
    const int per3 = (kt_d + a.ks3 - 1) / a.ks3, per5 = (kt_4d + a.ks5 - 1) / a.ks5;
    const size_t slab = (size_t)M * D;
    const int n_fin = (M + 7) / 8;                        // finish tasks (8 rows each)
    float* xa = a.x;                                      // residual stream entering the layer; x1 after the attention half
    float* xb = a.x1;
    for (int l = 0; l < a.NL; l++) {
        const long long base = a.layer0 + (long long)l * a.layer_stride;
        const long long l1w = base, aw = l1w + 2LL * D, ab = aw + 3LL * D * D, pw = ab + 3LL * D, pb = pw + (long long)D * D, l2w = pb + D,
                        l2b = l2w + D, fw = l2b + D, fb = fw + 4LL * D * D, p2w = fb + 4LL * D, p2b = p2w + 4LL * D * D, nxt = p2b + D;
        act_t* kc = a.kv + (size_t)l * a.cache_layer;
        act_t* vc = kc + a.cache_layer / 2;
        const unsigned lu = (unsigned)l;
        // ---- P1: qkv = ln_1(x) W_attn + b     (needs this row tile's previous finish)
        {
            const PkSched sc(nb, nrt, tn_3d);
            for (int i = sc.q, n = sc.count(); i < n; i += sc.per) {
                int ct, rt;
                if (!sc.get(i, ct, rt)) continue;
                PK_T0()
                const PkDep dep{a.ctr + C5F * PK_MAX_RT, rt, rt, lu * (unsigned)a.nfin_rt[rt]};
                const PkEpiAct epi{a.qkv, a.w32 + ab, 3 * D, M, 0};
                if (!pk_gemm_tile(a.xn, D, M, rt * 64, a.w16t + aw, D, ct * 64, 0, kt_d, pk_sm, epi, dep, err, sflagp)) return;
                pk_signal(a.ctr + C1 * PK_MAX_RT, rt, rt);
                PK_T1(0)
            }
        }
        // ---- P2: beam-group attention, two heads of one group per workgroup task
        {
            const int hp = H >> 1, ntask = a.NG * hp;
            const int team = tid >> 8, ttid = tid & 255;
            float* tsm = reinterpret_cast<float*>(pk_sm) + (size_t)team * a.attn_floats;
            for (int t = blockIdx.x; t < ntask; t += nb) {
                const int s = t / hp, h = (t - s * hp) * 2 + team;
                const int r0 = (s * G) >> 6, r1 = (s * G + G - 1) >> 6;
                PK_T0()
                const PkDep dep{a.ctr + C1 * PK_MAX_RT, r0, r1, (lu + 1u) * (unsigned)tn_3d};
                if (!pk_wait(dep, err, sflagp)) return;
                pk_attn_team<G>(a.qkv, kc, vc, a.ent + (size_t)s * a.cap, a.cnt[s], a.att, s, h, H, a.pos0, a.ctx_max, a.scale, a.cap, tsm, ttid);
                pk_signal(a.ctr + C2 * PK_MAX_RT, r0, r1);
                PK_T1(1)
            }
        }
        // ---- P3: attn.c_proj partials (K slices)      P3f: + bias + residual, ln_2
        {
            const PkSched sc(nb, nrt, tn_d * a.ks3);
            for (int i = sc.q, n = sc.count(); i < n; i += sc.per) {
                int u, rt;
                if (!sc.get(i, u, rt)) continue;
                const int ct = u / a.ks3, z = u - ct * a.ks3, k0 = z * per3, nk = min(per3, kt_d - k0);
                PK_T0()
                const PkDep dep{a.ctr + C2 * PK_MAX_RT, rt, rt, (lu + 1u) * (unsigned)a.n2[rt]};
                const PkEpiSlab epi{a.slab + (size_t)z * slab, D, M};
                if (nk > 0) {
                    if (!pk_gemm_tile(a.att, D, M, rt * 64, a.w16t + pw, D, ct * 64, k0, nk, pk_sm, epi, dep, err, sflagp)) return;
                } else if (!pk_wait(dep, err, sflagp)) return;
                pk_signal(a.ctr + C3 * PK_MAX_RT, rt, rt);
                PK_T1(2)
            }
            for (int t = blockIdx.x; t < n_fin; t += nb) {
                const int rt = (t * 8) >> 6;
                PK_T0()
                const PkDep dep{a.ctr + C3 * PK_MAX_RT, rt, rt, (lu + 1u) * (unsigned)(tn_d * a.ks3)};
                if (!pk_wait(dep, err, sflagp)) return;
                pk_finish_rows(a.slab, slab, a.ks3_eff, t * 8 + (tid >> 6), M, D, a.w32 + pb, xa, xb, a.w32 + l2w, a.w32 + l2b, a.xn);
                pk_signal(a.ctr + C3F * PK_MAX_RT, rt, rt);
                PK_T1(3)
            }
        }
        // ---- P4: h = gelu(ln_2(x1) W_fc + b)
        {
            const PkSched sc(nb, nrt, tn_4d);
            for (int i = sc.q, n = sc.count(); i < n; i += sc.per) {
                int ct, rt;
                if (!sc.get(i, ct, rt)) continue;
                PK_T0()
                const PkDep dep{a.ctr + C3F * PK_MAX_RT, rt, rt, (lu + 1u) * (unsigned)a.nfin_rt[rt]};
                const PkEpiAct epi{a.hact, a.w32 + fb, 4 * D, M, 2};
                if (!pk_gemm_tile(a.xn, D, M, rt * 64, a.w16t + fw, D, ct * 64, 0, kt_d, pk_sm, epi, dep, err, sflagp)) return;
                pk_signal(a.ctr + C4 * PK_MAX_RT, rt, rt);
                PK_T1(4)
            }
        }
        // ---- P5: mlp.c_proj partials      P5f: + bias + residual, the next layer's ln_1 (after the last layer: ln_f -> hf)
        {
            const PkSched sc(nb, nrt, tn_d * a.ks5);
            for (int i = sc.q, n = sc.count(); i < n; i += sc.per) {
                int u, rt;
                if (!sc.get(i, u, rt)) continue;
                const int ct = u / a.ks5, z = u - ct * a.ks5, k0 = z * per5, nk = min(per5, kt_4d - k0);
                PK_T0()
                const PkDep dep{a.ctr + C4 * PK_MAX_RT, rt, rt, (lu + 1u) * (unsigned)tn_4d};
                const PkEpiSlab epi{a.slab + (size_t)z * slab, D, M};
                if (nk > 0) {
                    if (!pk_gemm_tile(a.hact, 4 * D, M, rt * 64, a.w16t + p2w, 4 * D, ct * 64, k0, nk, pk_sm, epi, dep, err, sflagp)) return;
                } else if (!pk_wait(dep, err, sflagp)) return;
                pk_signal(a.ctr + C5 * PK_MAX_RT, rt, rt);
                PK_T1(5)
            }
            const bool last = l + 1 == a.NL;
            for (int t = blockIdx.x; t < n_fin; t += nb) {
                const int rt = (t * 8) >> 6;
                PK_T0()
                const PkDep dep{a.ctr + C5 * PK_MAX_RT, rt, rt, (lu + 1u) * (unsigned)(tn_d * a.ks5)};
                if (!pk_wait(dep, err, sflagp)) return;
                pk_finish_rows(a.slab, slab, a.ks5_eff, t * 8 + (tid >> 6), M, D, a.w32 + p2b, xb, xa, a.w32 + nxt, a.w32 + nxt + D, last ? a.hf : a.xn);
                pk_signal(a.ctr + C5F * PK_MAX_RT, rt, rt);
                PK_T1(6)
            }
        }
    }
    if (a.prof && threadIdx.x == 0)
        for (int i = 0; i < 21; i++) a.prof[(size_t)blockIdx.x * 21 + i] = pprof[i];
#undef PK_T0
#undef PK_T1
}

}  // namespace

int decode_layers_persistent(const PkLaunch& L, hipStream_t st) {
    const int D = L.D, H = L.H, M = L.M, G = L.group;
    if (H * 64 != D || (H & 1) || D > 2048 || G < 2 || G > 8 || M % G || M > 64 * PK_MAX_RT || L.NL < 1) return CC_ERR_SHAPE;
    static int n_cu = -1;
    if (n_cu < 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return CC_ERR_LAUNCH;
        n_cu = p.multiProcessorCount;
    }
    const int nb = (std::min(n_cu, 256) / 8) * 8;        // one workgroup per CU; a multiple of 8 for the XCD-aware schedule
    if (nb < 8) return CC_ERR_SHAPE;
    PkArgs a{};
    a.w32 = L.w32; a.w16t = L.w16t; a.D = D; a.H = H; a.NL = L.NL; a.layer0 = L.layer0; a.layer_stride = 12LL * D * D + 13LL * D;
    a.M = M; a.NG = M / G; a.nrt = (M + 63) / 64; a.pos0 = L.pos0; a.ctx_max = L.ctx_max; a.scale = 0.125f;
    a.x = L.x; a.x1 = L.x1; a.xn = L.xn; a.qkv = L.qkv; a.att = L.att; a.hact = L.hact; a.hf = L.hf; a.slab = L.slab;
    a.prof = L.prof; a.kv = L.kv; a.cache_layer = L.cache_layer; a.ent = L.ent; a.cnt = L.cnt; a.cap = L.cap; a.ctr = L.ctr;
    const int tiles = a.nrt * (D / 64);
    auto pick = [&](int kt) { int ks = std::max(1, nb / std::max(1, tiles)); ks = std::min(ks, std::max(1, kt / 4)); return ks; };
    a.ks3 = pick(D / 64); a.ks5 = pick(4 * D / 64);
    if ((size_t)std::max(a.ks3, a.ks5) * M * D * sizeof(float) > L.slab_bytes) return CC_ERR_SHAPE;
    { const int per = (D / 64 + a.ks3 - 1) / a.ks3; a.ks3_eff = (D / 64 + per - 1) / per; }
    { const int per = (4 * D / 64 + a.ks5 - 1) / a.ks5; a.ks5_eff = (4 * D / 64 + per - 1) / per; }
    for (int rt = 0; rt < a.nrt; rt++) {
        const int lo = rt * 64, hi = std::min(M, lo + 64) - 1;
        a.nfin_rt[rt] = hi / 8 - lo / 8 + 1;
        a.n2[rt] = ((hi / G) - (lo / G) + 1) * (H / 2);
    }
    a.attn_floats = (L.cap * 8 + 96 + 8 * G * 64 + 255) & ~255;
    if ((size_t)2 * a.attn_floats * sizeof(float) > PK_LDS) return CC_ERR_SHAPE;
#define PK_GO(G_)                                                                                                                     \
    case G_: {                                                                                                                        \
        static bool attr = false;                                                                                                     \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_decode_layers<G_>, hipFuncAttributeMaxDynamicSharedMemorySize, PK_LDS + 256); attr = true; } \
        hipLaunchKernelGGL((k_decode_layers<G_>), dim3(nb), dim3(PK_THREADS), PK_LDS + 256, st, a);                                         \
    } break;
    switch (G) { PK_GO(2) PK_GO(3) PK_GO(4) PK_GO(5) PK_GO(6) PK_GO(7) PK_GO(8) default: return CC_ERR_SHAPE; }
#undef PK_GO
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}
#else
int decode_layers_persistent(const PkLaunch&, hipStream_t) { return CC_ERR_SHAPE; }
#endif
}  // namespace CC_NS
