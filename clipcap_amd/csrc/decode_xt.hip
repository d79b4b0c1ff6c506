// XCD-team decode engine: one launch runs the whole GPT-2 layer stack of a generated position (KV-cached decode, one new position per
// row), each XCD for its OWN rows (reference: the full re-forward per generated token, clipcap/inference/base.py:80-121).
//
// Why teams.  At M = 320 rows the launch-per-op path is a chain of 7 dependent launches per layer whose arithmetic is 1-5 us each
// (DESIGN 4.4); the persistent form with cross-XCD hand-offs (decode_pk.hip, round 4) lost 2x because a hand-off between XCDs has to
// go through the memory side.  Here nothing crosses an XCD inside a layer: the 32 workgroups of XCD t (found from HW_REG_XCC_ID at run
// time, never from blockIdx) own captions [t cpt, (t+1) cpt) = at most 48 rows, and run c_attn, beam-group attention, attn.c_proj,
// c_fc and mlp.c_proj for those rows back to back.  Hand-offs are plain stores (they stay in that XCD's L2), a drained vmcnt, one
// atomic arrival per wave on the team's counter, one relaxed sc1 poll, and L1-bypassing (sc1) loads on the consumer side
// (tools/probes/xcd_team.hip: 0.96 us per team barrier on an idle chip, payload check clean).  The price is that every XCD streams every
// weight: 8 x 24 MB per layer through the fabric, 28 us per layer at the measured 6.9 TB/s (tools/probes/xcd_stream.hip) — which is why
// the weight stream is what the design is built around:
//
//   * WEIGHTS NEVER TOUCH LDS.  cc_decode_xt_image lays every weight out in MFMA fragment order (a 16 column x 32 k block = 1 KiB: lane l
//     holds W[n0 + (l & 15)][k0 + 8 (l >> 4) .. + 7]) and concatenates, per (workgroup, wave), the fragments of ALL layers in the order
//     that wave consumes them.  A wave's weight stream is therefore ONE contiguous run of 1-KiB pieces; it keeps a ring of RING
//     fragments in registers (global_load_dwordx4, fully coalesced), refills a slot the moment it is consumed, and so prefetches
//     across phase and layer boundaries for free: the stream never waits for a hand-off, only for the fabric.
//   * two kinds of waves per workgroup (8 waves, 256 registers each).  Waves 0-3 own the rings and do nothing but fragment reads of the
//     activation panel (LDS) and MFMAs: they never store to global memory and never wait vmcnt(0), so a deep ring costs no stall (one
//     vmcnt counter per wave: a wave that must know its stores have landed cannot keep loads in flight).  Waves 4-7 do everything that
//     needs vmcnt(0): they normalise / load the activation panel into LDS, sum the MFMA waves' partial tiles, apply bias / residual /
//     gelu, store, arrive on the team counter and poll it; they also are the 4-wave team of the attention phase.
//   * GEMM phase inside a workgroup: columns of the layer's weight matrix are split over the 32 workgroups of the team (96 / 32 / 128 / 32
//     columns at D = 1024), K over the four MFMA waves (c_fc: columns over the waves), partial tiles meet in LDS.
//
// Covered: bf16 / fp16 builds, D = 512 or 1024 (head dim 64), beam groups of 2..8 rows, at most 48 rows per XCD, a device that gives
// every XCD exactly 32 resident workgroups.  Anything else: CC_ERR_SHAPE from the host function (the caller keeps the launch-per-op path);
// a launch whose teams do not form, or whose spins time out, sets the error word AND the sticky word (cc_decode_ws_check).
#include "../../include/clipcap_hip.h"
#include "kernels.h"
#include "decode_xt.h"
#include <algorithm>

namespace CC_NS {
#if CC_OP != 2 && defined(CC_EXPERIMENTS)      // lab build only (make lab): measured slower than the per-op launches (HISTORY.md 4.5)
namespace {

typedef const __attribute__((address_space(1))) void* xg_t;
typedef __attribute__((address_space(3))) void* xl_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define XT_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned XT_SPIN_MAX = 1u << 20;
constexpr int XT_TEAM = 32;                    // workgroups per team = CUs per XCD

struct XtArgs {
    const float* w32;
    const u32x4* wimg;
    int NL, M, NG, cpt, pos0, ctx_max, cap, attn_floats, rt_max, lds_main, nbuf, lab_aux;      // lds_main: bytes of the panel / partial / attention region; the flag words sit behind it
    float scale;
    long long layer0, layer_stride;
    float *x, *x1;
    act_t *qkv, *att, *hact, *hf;
    act_t* kv;
    size_t cache_layer;
    const int2* ent;
    const int* cnt;
    unsigned* ctl;
    unsigned* sticky;
    unsigned long long* prof;
};

template <int D>
struct XtGeo {
    static constexpr int H = D / 64;
    static constexpr int KB = D / 32;                 // 32-wide k blocks of K = D
    static constexpr int KBW = KB / 4;                // ... per MFMA wave (K split over the four)
    static constexpr int CW_A = 3 * D / 32, NJ_A = CW_A / 16;      // c_attn: columns per workgroup, 16-column fragments
    static constexpr int CW_P = D / 32, NJ_P = CW_P / 16;          // attn.c_proj / mlp.c_proj
    static constexpr int CW_F = 4 * D / 32, NJ_F = CW_F / 4 / 16;  // c_fc: columns split over the waves, D / 32 per wave
    static constexpr int NC = 4 * D / 512;            // mlp.c_proj: K = 4D in chunks of 512
    static constexpr int FR_A = NJ_A * KBW, FR_P = NJ_P * KBW, FR_F = NJ_F * KB, FR_M = NC * NJ_P * 4;
    static constexpr int FR = FR_A + FR_P + FR_F + FR_M;      // fragments per wave and layer
    static constexpr int SA = 2 * D + 32;             // LDS row stride (bytes) of an activation panel with K = D: 16-B slot = (2 row + chunk) mod 16, conflict-free for ds_read_b128
    static constexpr int SC = 1024 + 32;              // ... of a 512-wide chunk
    static constexpr int A_ROW = SA > 2 * SC ? SA : 2 * SC;   // panel region per row: one K = D panel or two chunk buffers
    static constexpr int P_ROW = 4 * CW_A * 4;        // partial-tile region per row: four waves x the widest phase, fp32
    static_assert(D % 512 == 0, "column split needs 16-column fragments per workgroup");
};

__device__ __forceinline__ uint4 ld16_sc1(const void* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, XT_RLX), b = __hip_atomic_load(q + 1, XT_RLX);
    return make_uint4((unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32));
}
__device__ __forceinline__ float4 ldf4_sc1(const float* p) {
    const uint4 r = ld16_sc1(p);
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}
__device__ __forceinline__ void xt_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void xt_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void xt_vm_le(int n) {      // s_waitcnt vmcnt(n), n wave-uniform; larger counts wait for 12 (conservative)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void xt_bar() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

// ---- I/O waves: team barrier.  arrive = every I/O wave, after its own stores have landed in the L2; wait = wave 4 polls the team counter,
// the other I/O waves watch an LDS word.  epoch e is complete when the counter reaches e * 4 * 32.
__device__ __forceinline__ void xt_arrive(unsigned* ctr, int lane) {
    xt_vm0();
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, XT_RLX);
}
__device__ __forceinline__ bool xt_wait(unsigned* ctr, unsigned epoch, unsigned* err, volatile unsigned* lflag, int iw, int lane, unsigned long long& t_poll) {
    if (iw == 0) {
        if (lane == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            const unsigned target = epoch * 4u * XT_TEAM;
            unsigned spins = 0, v = epoch;
            while (__hip_atomic_load(ctr, XT_RLX) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 63u) == 0 && (spins > XT_SPIN_MAX || __hip_atomic_load(err, XT_RLX) != 0u)) { v = 0xffffffffu; break; }
            }
            *lflag = v;
            t_poll += __builtin_amdgcn_s_memrealtime() - t0;
        }
        xt_lgkm0();
    }
    unsigned v;
    while ((v = *lflag) < epoch) __builtin_amdgcn_s_sleep(1);
    return v != 0xffffffffu;
}

// ---- I/O waves: activation panel = LayerNorm(rows of the fp32 residual stream) as 16-bit operands, LDS rows SA bytes apart.
// (two-pass statistics like k_splitk_finish_row; the summation order differs: fp32 rounding only)
template <int D>
__device__ __forceinline__ void xt_fill_ln(const float* __restrict__ src, int Rt, const float* __restrict__ gamma, const float* __restrict__ beta,
                                           char* panel, int iw, int lane, int xt_lab_aux = 16) {
    constexpr int NV = D / 256, RB = 6, SA = XtGeo<D>::SA;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    float4 g[NV], b[NV];
#pragma unroll
    for (int it = 0; it < NV; it++) {
        g[it] = *reinterpret_cast<const float4*>(gamma + lane * 4 + it * 256);
        b[it] = *reinterpret_cast<const float4*>(beta + lane * 4 + it * 256);
    }
    // rows written by the other workgroups of the team in this launch: 16-B loads that bypass this CU's L1 (sc1), RB rows in flight per wave
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, Rt * D * 4, 0x00020000);
    for (int r0 = iw; r0 < Rt; r0 += 4 * RB) {
        float4 v[RB][NV];
#pragma unroll
        for (int k = 0; k < RB; k++) {
            const int r = min(r0 + 4 * k, Rt - 1);
#pragma unroll
            for (int it = 0; it < NV; it++) {
#ifdef CC_EXPERIMENTS
                v4u t;
                const int vo_ = (r * D + lane * 4 + it * 256) * 4;
                switch (xt_lab_aux) {
                    case 0: t = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, vo_, 0, 0)); break;
                    case 2: t = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, vo_, 0, 2)); break;
                    case 17: t = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, vo_, 0, 17)); break;
                    case 1: t = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, vo_, 0, 1)); break;
                    default: t = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, vo_, 0, 16)); break;
                }
#else
                const v4u t = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, (r * D + lane * 4 + it * 256) * 4, 0, 16));
#endif
                v[k][it] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
            }
        }
#pragma unroll
        for (int k = 0; k < RB; k++) {
            const int r = r0 + 4 * k;
            float s = 0.f;
#pragma unroll
            for (int it = 0; it < NV; it++) s += v[k][it].x + v[k][it].y + v[k][it].z + v[k][it].w;
            const float mu = wave_sum(s) / D;
            float q = 0.f;
#pragma unroll
            for (int it = 0; it < NV; it++) {
                const float a = v[k][it].x - mu, bb = v[k][it].y - mu, c = v[k][it].z - mu, d = v[k][it].w - mu;
                q += a * a + bb * bb + c * c + d * d;
            }
            const float rs_ = rsqrtf(wave_sum(q) / D + 1e-5f);
            if (r < Rt) {
#pragma unroll
                for (int it = 0; it < NV; it++) {
                    const uint2 pk = make_uint2(pack2op((v[k][it].x - mu) * rs_ * g[it].x + b[it].x, (v[k][it].y - mu) * rs_ * g[it].y + b[it].y),
                                                pack2op((v[k][it].z - mu) * rs_ * g[it].z + b[it].z, (v[k][it].w - mu) * rs_ * g[it].w + b[it].w));
                    *reinterpret_cast<uint2*>(panel + (size_t)r * SA + (lane * 4 + it * 256) * 2) = pk;
                }
            }
        }
    }
}

// ---- I/O waves: activation panel by LDS-DMA (rows produced in this launch: sc1).  `pieces` 1-KiB pieces per row (row = pieces * 512 elements
// at src + r * ld), LDS rows `stride` bytes apart.
__device__ __forceinline__ void xt_dma_rows(const act_t* __restrict__ src, size_t ld, int Rt, int pieces, char* panel, int stride, int iw, int lane) {
    const int n = Rt * pieces;
    for (int p = iw; p < n; p += 4) {
        const int r = p / pieces, h = p - r * pieces;
        __builtin_amdgcn_global_load_lds((xg_t)(src + (size_t)r * ld + h * 512 + lane * 8), (xl_t)(panel + (size_t)r * stride + h * 1024), 16, 0, 16);
    }
}

// ---- beam-group attention for ONE (caption, head) by ONE wave (decode.hip::k_decode_attn_group's arithmetic; q and the new K / V come from qkv,
// written in this launch: sc1).  The launch-per-op kernel spreads an item over four waves and two workgroup barriers; here each of the four
// I/O waves of a workgroup owns an item, so a workgroup's items run at once, and the chain per layer is ONE round trip: the union list of the
// item is the same in every layer (layer 0 leaves it in LDS, `ecache`), and every K row and V row of a 128-entry group is fetched by LDS-DMA
// (global_load_lds: per-lane source address, so one instruction gathers 8 arbitrary 128-B rows; no registers are held while the rows
// travel).  Unions wider than 128 entries take further groups with the running-maximum rescale.
// LDS of the wave (XT_ATTN_WAVE_BYTES): K rows [128][128 B] | V rows [128][128 B] | p[128][8] fp32 | red[G][64] fp32.
constexpr int XT_ATTN_WAVE_BYTES = 16384 + 16384 + 4096 + 8 * 64 * 4;
template <int G>
__device__ __forceinline__ void xt_attn_wave(const act_t* __restrict__ qkv, act_t* __restrict__ kc, act_t* __restrict__ vc, const int2* __restrict__ ent,
                                             int nU, act_t* __restrict__ out, int r0, int h, int D, int pos0, int ctx_max, float scale, char* wl,
                                             int2* ecache, bool cached, int lane) {
    constexpr int HD = 64;
    char* const Kt = wl;
    char* const Vt = wl + 16384;
    float* const p = reinterpret_cast<float*>(wl + 32768);
    float* const red = p + 1024;
    const int ngrp = (nU + 127) >> 7;
    const act_t* kb = kc + h * HD;
    const act_t* vb = vc + h * HD;
    const act_t* qrow = qkv + (size_t)r0 * 3 * D + h * HD;
    const int skg = lane >> 3, sdc = lane & 7, hf = lane >> 5, dp = lane & 31;
    for (int t = lane; t < G * 16; t += 64) {              // append this head's new K / V to the cache (read by the NEXT launch)
        const int b = t >> 4, wq = t & 15, which = wq >> 3, c = wq & 7;
        const uint4 v = ld16_sc1(qrow + (size_t)b * 3 * D + (which + 1) * D + c * 8);
        *reinterpret_cast<uint4*>((which ? vc : kc) + ((size_t)(r0 + b) * ctx_max + pos0) * D + h * HD + c * 8) = v;
    }
    uint4 qp[G];
#pragma unroll
    for (int b = 0; b < G; b++) qp[b] = ld16_sc1(qrow + (size_t)b * 3 * D + sdc * 8);
    float mrun[G], lsum[G], acc[G][2];
#pragma unroll
    for (int b = 0; b < G; b++) { mrun[b] = -INFINITY; lsum[b] = 0.f; acc[b][0] = acc[b][1] = 0.f; }
    for (int g = 0; g < ngrp; g++) {
        if (!(g == 0 && cached)) {                         // the group's 128 entries -> LDS (the list is padded to whole groups with owner-less entries)
            ecache[lane] = ent[g * 128 + lane];
            ecache[64 + lane] = ent[g * 128 + 64 + lane];
        }
        // gather the group's K and V rows: instruction j moves entries 8 j .. 8 j + 7, lane -> (entry 8 j + (lane >> 3), 16-B chunk lane & 7)
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int ex = ecache[8 * j + skg].x;
            const act_t* krow = ex < 0 ? qrow + (size_t)(-1 - ex) * 3 * D + D : kb + (size_t)ex * D;
            const act_t* vrow = ex < 0 ? qrow + (size_t)(-1 - ex) * 3 * D + 2 * D : vb + (size_t)ex * D;
            __builtin_amdgcn_global_load_lds((xg_t)(krow + sdc * 8), (xl_t)(Kt + j * 1024), 16, 0, 16);
            __builtin_amdgcn_global_load_lds((xg_t)(vrow + sdc * 8), (xl_t)(Vt + j * 1024), 16, 0, 16);
        }
        xt_vm0();
        float gmx[G];
#pragma unroll
        for (int b = 0; b < G; b++) gmx[b] = -INFINITY;
#pragma unroll 2
        for (int u0 = 0; u0 < 128; u0 += 8) {
            const int u = u0 + skg;
            float kf[8], sc[8];
            unpack8(*reinterpret_cast<const uint4*>(Kt + u * 128 + sdc * 16), kf);
            const int msk = ecache[u].y;
#pragma unroll
            for (int b = 0; b < G; b++) {
                float qf[8];
                unpack8(qp[b], qf);
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) a += qf[e] * kf[e];
                sc[b] = a;
            }
#pragma unroll
            for (int b = 0; b < G; b++) sc[b] = sum8(sc[b]);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                sc[b] = (b < G && ((msk >> b) & 1)) ? sc[b] * scale : -INFINITY;
                if (b < G) gmx[b] = fmaxf(gmx[b], sc[b]);
            }
            if (sdc == 0) {
                float* pu = p + (size_t)u * 8;
                *reinterpret_cast<float4*>(pu) = make_float4(sc[0], sc[1], sc[2], sc[3]);
                *reinterpret_cast<float4*>(pu + 4) = make_float4(sc[4], sc[5], sc[6], sc[7]);
            }
        }
        // running maximum over the groups (one group — the usual case — is exactly exp(s - max)); every beam owns its new key, so the
        // maximum is finite from the group that holds it on; before that group a beam's terms are all exp(-inf) = 0
#pragma unroll
        for (int b = 0; b < G; b++) {
            const float mn = fmaxf(mrun[b], wave_max(gmx[b]));
            const float f = mrun[b] == mn ? 1.f : __expf(mrun[b] - mn);      // (-inf) - (-inf) never evaluated
            acc[b][0] *= f; acc[b][1] *= f; lsum[b] *= f;
            mrun[b] = mn;
        }
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const int idx = jj * 64 + lane, b = idx & 7;
            if (b < G) {
                float m = mrun[0];
#pragma unroll
                for (int b2 = 1; b2 < G; b2++) m = b == b2 ? mrun[b2] : m;
                p[idx] = m == -INFINITY ? 0.f : __expf(p[idx] - m);
            }
        }
#pragma unroll 4
        for (int k = 0; k < 64; k++) {
            const int u = 2 * k + hf;
            float v0, v1;
            unpack2(*reinterpret_cast<const unsigned*>(Vt + u * 128 + dp * 4), v0, v1);
            const float4 pa = *reinterpret_cast<const float4*>(p + u * 8), pb = *reinterpret_cast<const float4*>(p + u * 8 + 4);
            const float pj[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
            for (int b = 0; b < G; b++) { acc[b][0] += pj[b] * v0; acc[b][1] += pj[b] * v1; lsum[b] += pj[b]; }
        }
    }
    // the two halves of the wave hold the even / odd entries' sums: combine through the wave's LDS tile, lanes 0-31 write the G rows
    if (hf) {
#pragma unroll
        for (int b = 0; b < G; b++) *reinterpret_cast<float2*>(red + b * HD + 2 * dp) = make_float2(acc[b][0], acc[b][1]);
    }
#pragma unroll
    for (int b = 0; b < G; b++) lsum[b] = lane_bcast(lsum[b], 0) + lane_bcast(lsum[b], 32);      // uniform inside each half: the halves' sums
    if (!hf) {
#pragma unroll
        for (int b = 0; b < G; b++) {
            const float2 o = *reinterpret_cast<const float2*>(red + b * HD + 2 * dp);
            const float inv = 1.f / lsum[b];
            *reinterpret_cast<unsigned*>(out + (size_t)(r0 + b) * D + h * HD + 2 * dp) = pack2op((acc[b][0] + o.x) * inv, (acc[b][1] + o.y) * inv);
        }
    }
}

// ---- I/O waves: sum the MFMA waves' partial tiles (NP of them, PS floats apart, rows CW floats apart), 8 consecutive columns of one row
__device__ __forceinline__ void xt_gather8(const float* P, int NP, int PS, int CW, int row, int c8, float (&v)[8]) {
    const float* s = P + (size_t)row * CW + c8 * 8;
    float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
    for (int w = 1; w < NP; w++) {
        const float4 c = *reinterpret_cast<const float4*>(s + (size_t)w * PS), d = *reinterpret_cast<const float4*>(s + (size_t)w * PS + 4);
        a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w; b.x += d.x; b.y += d.y; b.z += d.z; b.w += d.w;
    }
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void xt_bias8(const float* bias, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(bias), b = *reinterpret_cast<const float4*>(bias + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
}

template <int D, int G, int RING>
__global__ __launch_bounds__(512, 2) void k_decode_xt(XtArgs a) {
    typedef XtGeo<D> Geo;
    static_assert(Geo::FR % RING == 0 && RING <= XT_RING_MAX, "a layer must start at ring slot 0");
    extern __shared__ __attribute__((aligned(1024))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RtMax = a.rt_max;
    char* const panel = sm;                                                   // activation panel / chunk buffers
    float* const part = reinterpret_cast<float*>(sm + (size_t)RtMax * Geo::A_ROW);     // partial tiles
    float* const part_m = reinterpret_cast<float*>(sm + (size_t)a.nbuf * RtMax * Geo::SC);       // mlp.c_proj's partial tiles sit behind its chunk buffers
    unsigned* const lflag = reinterpret_cast<unsigned*>(sm + a.lds_main);
    int2* const ecache = reinterpret_cast<int2*>(sm + a.lds_main + 64);      // [4 I/O waves][128] union-list entries of the wave's attention item (the same in every layer)   // [0] epoch word, [1] xcc, [2] rank, [3] ok
    unsigned* const err = a.ctl + 9;
    // ---- teams: XCD from the hardware register, rank = ticket; every workgroup sees the same eight counts after the arrival spin
    if (tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;          // HW_REG_XCC_ID[3:0]
        const unsigned rank = __hip_atomic_fetch_add(a.ctl + xcc, 1u, XT_RLX);
        __hip_atomic_fetch_add(a.ctl + 8, 1u, XT_RLX);
        unsigned spins = 0, ok = 1;
        while (__hip_atomic_load(a.ctl + 8, XT_RLX) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > XT_SPIN_MAX) { ok = 0; break; }
        }
        for (int t = 0; t < 8 && ok; t++)
            if (__hip_atomic_load(a.ctl + t, XT_RLX) != (unsigned)XT_TEAM) ok = 0;
        if (!ok) { __hip_atomic_store(err, 1u, XT_RLX); __hip_atomic_store(a.sticky, 1u, XT_RLX); }
        lflag[0] = 0; lflag[1] = xcc; lflag[2] = rank; lflag[3] = ok;
    }
    __syncthreads();
    if (!lflag[3]) return;
    const int team = (int)lflag[1], rank = (int)lflag[2];
    const int cap0 = team * a.cpt, ncap = max(0, min(a.NG, cap0 + a.cpt) - cap0);
    const int row0 = cap0 * G, Rt = ncap * G;
    if (Rt == 0) return;                                                      // a team without captions has nothing to wait for
    unsigned* const ctr = a.ctl + 16 + 32 * team;
    const int n_items = ncap * Geo::H, n_it = (n_items + 4 * XT_TEAM - 1) / (4 * XT_TEAM);      // attention: one (caption, head) item per I/O wave and round
    const int q = lane >> 4, rl = lane & 15;

    if (wave < 4) {
        // =========================== MFMA waves: weight ring + fragment reads + MFMAs, nothing else ===========================
        const int w = wave;
        const u32x4* gp = a.wimg + ((size_t)(rank * 4 + w) * a.NL * Geo::FR) * 64 + lane;
        u32x4 ring[RING];
#pragma unroll
        for (int s = 0; s < RING; s++) ring[s] = gp[(size_t)s * 64];
        // LDS byte offsets of this lane's three row fragments (rows beyond the team's are clamped: their results are dropped)
        int aoff[3], coff[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int r = min(i * 16 + rl, Rt - 1);
            aoff[i] = r * Geo::SA + q * 16;
            coff[i] = r * Geo::SC + q * 16;
        }
        for (int l = 0; l < a.NL; l++) {
#define XT_TAKE(FI)                                                                                                      \
    const u32x4 wf_ = ring[(FI) % RING];                                                                                 \
    ring[(FI) % RING] = gp[(size_t)((FI) + RING) * 64];                                                                  \
    const op16x8 wfr = __builtin_bit_cast(op16x8, wf_);
            // ---- P1 c_attn: K split over the waves
            xt_bar();                                                         // B1: panel = ln_1(x)
#pragma unroll
            for (int j = 0; j < Geo::NJ_A; j++) {
                f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
                for (int kb = 0; kb < Geo::KBW; kb++) {
                    XT_TAKE(j * Geo::KBW + kb)
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const op16x8 af = *reinterpret_cast<const op16x8*>(panel + aoff[i] + (w * Geo::KBW + kb) * 64);
                        acc[i] = CC_MFMA_16x16x32(wfr, af, acc[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if (i * 16 + rl < Rt) *reinterpret_cast<f32x4*>(part + ((size_t)(w * RtMax + i * 16 + rl) * Geo::CW_A + j * 16 + q * 4)) = acc[i];
            }
            xt_lgkm0();
            xt_bar();                                                         // B2
            // ---- P2 attention: the I/O waves' phase (one item per wave: no workgroup barrier)
            // ---- P3 attn.c_proj
            xt_bar();
#pragma unroll
            for (int j = 0; j < Geo::NJ_P; j++) {
                f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
                for (int kb = 0; kb < Geo::KBW; kb++) {
                    XT_TAKE(Geo::FR_A + j * Geo::KBW + kb)
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const op16x8 af = *reinterpret_cast<const op16x8*>(panel + aoff[i] + (w * Geo::KBW + kb) * 64);
                        acc[i] = CC_MFMA_16x16x32(wfr, af, acc[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if (i * 16 + rl < Rt) *reinterpret_cast<f32x4*>(part + ((size_t)(w * RtMax + i * 16 + rl) * Geo::CW_P + j * 16 + q * 4)) = acc[i];
            }
            xt_lgkm0();
            xt_bar();
            // ---- P4 c_fc: columns split over the waves (no partial sums), all of K per wave
            xt_bar();
#pragma unroll
            for (int j = 0; j < Geo::NJ_F; j++) {
                f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
                for (int kb = 0; kb < Geo::KB; kb++) {
                    XT_TAKE(Geo::FR_A + Geo::FR_P + j * Geo::KB + kb)
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const op16x8 af = *reinterpret_cast<const op16x8*>(panel + aoff[i] + kb * 64);
                        acc[i] = CC_MFMA_16x16x32(wfr, af, acc[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if (i * 16 + rl < Rt) *reinterpret_cast<f32x4*>(part + ((size_t)(i * 16 + rl) * Geo::CW_F + w * (Geo::CW_F / 4) + j * 16 + q * 4)) = acc[i];
            }
            xt_lgkm0();
            xt_bar();
            // ---- P5 mlp.c_proj: K = 4D arrives in 512-wide chunks (two LDS buffers), four k blocks per wave and chunk
            {
                f32x4 acc[Geo::NJ_P][3];
#pragma unroll
                for (int j = 0; j < Geo::NJ_P; j++)
#pragma unroll
                    for (int i = 0; i < 3; i++) acc[j][i] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < Geo::NC; c++) {
                    xt_bar();                                                 // chunk c landed; every wave is past chunk c - 1
                    const char* cb = panel + (size_t)(c % a.nbuf) * RtMax * Geo::SC;
#pragma unroll
                    for (int j = 0; j < Geo::NJ_P; j++)
#pragma unroll
                        for (int kb = 0; kb < 4; kb++) {
                            XT_TAKE(Geo::FR_A + Geo::FR_P + Geo::FR_F + (c * Geo::NJ_P + j) * 4 + kb)
#pragma unroll
                            for (int i = 0; i < 3; i++) {
                                const op16x8 af = *reinterpret_cast<const op16x8*>(cb + coff[i] + (w * 4 + kb) * 64);
                                acc[j][i] = CC_MFMA_16x16x32(wfr, af, acc[j][i]);
                            }
                        }
                }
#pragma unroll
                for (int j = 0; j < Geo::NJ_P; j++)
#pragma unroll
                    for (int i = 0; i < 3; i++)
                        if (i * 16 + rl < Rt) *reinterpret_cast<f32x4*>(part_m + ((size_t)(w * RtMax + i * 16 + rl) * Geo::CW_P + j * 16 + q * 4)) = acc[j][i];
            }
            xt_lgkm0();
            xt_bar();
#undef XT_TAKE
            gp += (size_t)Geo::FR * 64;
        }
        return;
    }

    // =========================== I/O waves: panels, epilogues, hand-offs, attention ===========================
    const int iw = wave - 4, itid = tid - 256;
    unsigned ep = 0;
    unsigned long long t_poll = 0, t_prev = __builtin_amdgcn_s_memrealtime();
    unsigned long long* const tph = reinterpret_cast<unsigned long long*>(sm + a.lds_main + 64 + 4 * 128 * 8);      // [24] profile sums (LDS: no registers held across the kernel)
    if (a.prof && itid == 0)
        for (int i = 0; i < 24; i++) tph[i] = 0;
#define XT_PH(I) if (a.prof && itid == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tph[I] += n_ - t_prev; t_prev = n_; }
    float* xa = a.x + (size_t)row0 * D;
    float* xb = a.x1 + (size_t)row0 * D;
    act_t* const qkv = a.qkv + (size_t)row0 * 3 * D;
    act_t* const att = a.att + (size_t)row0 * D;
    act_t* const hact = a.hact + (size_t)row0 * 4 * D;
    const int PS = RtMax * Geo::CW_A;      // (set per phase below)
    (void)PS;
    for (int l = 0; l < a.NL; l++) {
        const long long base = a.layer0 + (long long)l * a.layer_stride;
        const long long l1w = base, ab = l1w + 2LL * D + 3LL * D * D, pb = ab + 3LL * D + (long long)D * D, l2w = pb + D, fb = l2w + 2LL * D + 4LL * D * D,
                        p2b = fb + 4LL * D + 4LL * D * D, nxt = p2b + D;
        act_t* kc = a.kv + (size_t)l * a.cache_layer;
        act_t* vc = kc + a.cache_layer / 2;
        // ---- P1 c_attn: panel = ln_1(x)
        if (l > 0 && !xt_wait(ctr, ep, err, lflag, iw, lane, t_poll)) return;
        XT_PH(0)
        xt_fill_ln<D>(xa, Rt, a.w32 + l1w, a.w32 + l1w + D, panel, iw, lane, a.lab_aux);
        xt_lgkm0();
        xt_bar();                                                             // B1
        XT_PH(1)
        xt_bar();                                                             // B2: partial tiles complete
        XT_PH(2)
        for (int o = itid; o < Rt * (Geo::CW_A / 8); o += 256) {
            const int row = o / (Geo::CW_A / 8), c8 = o - row * (Geo::CW_A / 8);
            float v[8];
            xt_gather8(part, 4, RtMax * Geo::CW_A, Geo::CW_A, row, c8, v);
            const int col = rank * Geo::CW_A + c8 * 8;
            xt_bias8(a.w32 + ab + col, v);
            act_st8(qkv + (size_t)row * 3 * D + col, v);
        }
        ep++;
        xt_arrive(ctr, lane);
        XT_PH(3)
        // ---- P2 attention: (caption, head) items of this team, one per workgroup and round
        if (!xt_wait(ctr, ep, err, lflag, iw, lane, t_poll)) return;
        XT_PH(4)
        for (int it = 0; it < n_it; it++) {
            const int item = (it * XT_TEAM + rank) * 4 + iw;
            if (item < n_items) {
                const int s_ = item / Geo::H, h = item - s_ * Geo::H, sg = cap0 + s_;
                const int nU = a.cnt[sg];
                const bool cached = it == 0 && nU <= 128 && l > 0;      // layer 0 left this item's (only) group in the wave's entry cache
                xt_attn_wave<G>(a.qkv, kc, vc, a.ent + (size_t)sg * a.cap, nU, a.att, sg * G, h, D, a.pos0, a.ctx_max, a.scale,
                                sm + (size_t)iw * XT_ATTN_WAVE_BYTES, ecache + iw * 128, cached, lane);
            }
        }
        XT_PH(5)
        ep++;
        xt_arrive(ctr, lane);
        XT_PH(7)
        // ---- P3 attn.c_proj + bias + residual -> x1
        if (!xt_wait(ctr, ep, err, lflag, iw, lane, t_poll)) return;
        XT_PH(8)
        xt_dma_rows(att, D, Rt, D / 512, panel, Geo::SA, iw, lane);
        xt_vm0();
        xt_bar();
        XT_PH(9)
        xt_bar();
        XT_PH(10)
        for (int o = itid; o < Rt * (Geo::CW_P / 8); o += 256) {
            const int row = o / (Geo::CW_P / 8), c8 = o - row * (Geo::CW_P / 8);
            float v[8];
            xt_gather8(part, 4, RtMax * Geo::CW_P, Geo::CW_P, row, c8, v);
            const int col = rank * Geo::CW_P + c8 * 8;
            xt_bias8(a.w32 + pb + col, v);
            const float4 r0 = ldf4_sc1(xa + (size_t)row * D + col), r1 = ldf4_sc1(xa + (size_t)row * D + col + 4);
            *reinterpret_cast<float4*>(xb + (size_t)row * D + col) = make_float4(v[0] + r0.x, v[1] + r0.y, v[2] + r0.z, v[3] + r0.w);
            *reinterpret_cast<float4*>(xb + (size_t)row * D + col + 4) = make_float4(v[4] + r1.x, v[5] + r1.y, v[6] + r1.z, v[7] + r1.w);
        }
        ep++;
        xt_arrive(ctr, lane);
        XT_PH(11)
        // ---- P4 c_fc: panel = ln_2(x1); gelu epilogue
        if (!xt_wait(ctr, ep, err, lflag, iw, lane, t_poll)) return;
        XT_PH(12)
        xt_fill_ln<D>(xb, Rt, a.w32 + l2w, a.w32 + l2w + D, panel, iw, lane, a.lab_aux);
        xt_lgkm0();
        xt_bar();
        XT_PH(13)
        xt_bar();
        XT_PH(14)
        for (int o = itid; o < Rt * (Geo::CW_F / 8); o += 256) {
            const int row = o / (Geo::CW_F / 8), c8 = o - row * (Geo::CW_F / 8);
            float v[8];
            xt_gather8(part, 1, 0, Geo::CW_F, row, c8, v);
            const int col = rank * Geo::CW_F + c8 * 8;
            xt_bias8(a.w32 + fb + col, v);
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = gelu_new_f(v[e]);
            act_st8(hact + (size_t)row * 4 * D + col, v);
        }
        ep++;
        xt_arrive(ctr, lane);
        XT_PH(15)
        // ---- P5 mlp.c_proj + bias + residual -> x
        if (!xt_wait(ctr, ep, err, lflag, iw, lane, t_poll)) return;
        XT_PH(16)
        {
            // chunk c + nbuf - 1 is requested as soon as every wave is past chunk c - 1 (its buffer); nbuf - 1 chunks in flight
            const int per = (Rt - iw + 3) / 4;             // DMA instructions of this wave per chunk
            for (int c = 0; c < a.nbuf - 1 && c < Geo::NC; c++) xt_dma_rows(hact + (size_t)c * 512, 4 * D, Rt, 1, panel + (size_t)c * RtMax * Geo::SC, Geo::SC, iw, lane);
            for (int c = 0; c < Geo::NC; c++) {
                const int later = min(Geo::NC - 1, c + a.nbuf - 2) - c;       // chunks requested after chunk c that may stay in flight
                xt_vm_le(later * per);
                xt_bar();
                const int cn = c + a.nbuf - 1;
                if (cn < Geo::NC) xt_dma_rows(hact + (size_t)cn * 512, 4 * D, Rt, 1, panel + (size_t)(cn % a.nbuf) * RtMax * Geo::SC, Geo::SC, iw, lane);
            }
        }
        xt_bar();
        XT_PH(18)
        for (int o = itid; o < Rt * (Geo::CW_P / 8); o += 256) {
            const int row = o / (Geo::CW_P / 8), c8 = o - row * (Geo::CW_P / 8);
            float v[8];
            xt_gather8(part_m, 4, RtMax * Geo::CW_P, Geo::CW_P, row, c8, v);
            const int col = rank * Geo::CW_P + c8 * 8;
            xt_bias8(a.w32 + p2b + col, v);
            const float4 r0 = ldf4_sc1(xb + (size_t)row * D + col), r1 = ldf4_sc1(xb + (size_t)row * D + col + 4);
            *reinterpret_cast<float4*>(xa + (size_t)row * D + col) = make_float4(v[0] + r0.x, v[1] + r0.y, v[2] + r0.z, v[3] + r0.w);
            *reinterpret_cast<float4*>(xa + (size_t)row * D + col + 4) = make_float4(v[4] + r1.x, v[5] + r1.y, v[6] + r1.z, v[7] + r1.w);
        }
        ep++;
        xt_arrive(ctr, lane);
        XT_PH(19)
        if (l + 1 == a.NL) {
            // ---- ln_f of this team's rows -> hf (one row per I/O wave and round); read by the lm_head launch that follows
            if (!xt_wait(ctr, ep, err, lflag, iw, lane, t_poll)) return;
            constexpr int NV = D / 256;
            for (int r = rank * 4 + iw; r < Rt; r += 4 * XT_TEAM) {
                float4 v[NV];
                float s = 0.f;
#pragma unroll
                for (int it = 0; it < NV; it++) { v[it] = ldf4_sc1(xa + (size_t)r * D + lane * 4 + it * 256); s += v[it].x + v[it].y + v[it].z + v[it].w; }
                const float mu = wave_sum(s) / D;
                float qq = 0.f;
#pragma unroll
                for (int it = 0; it < NV; it++) { const float e0 = v[it].x - mu, e1 = v[it].y - mu, e2 = v[it].z - mu, e3 = v[it].w - mu; qq += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3; }
                const float rs = rsqrtf(wave_sum(qq) / D + 1e-5f);
#pragma unroll
                for (int it = 0; it < NV; it++) {
                    const int c = lane * 4 + it * 256;
                    const float4 g = *reinterpret_cast<const float4*>(a.w32 + nxt + c), b = *reinterpret_cast<const float4*>(a.w32 + nxt + D + c);
                    act_st4(a.hf + (size_t)(row0 + r) * D + c, (v[it].x - mu) * rs * g.x + b.x, (v[it].y - mu) * rs * g.y + b.y, (v[it].z - mu) * rs * g.z + b.z,
                            (v[it].w - mu) * rs * g.w + b.w);
                }
            }
            XT_PH(20)
        }
    }
    if (a.prof && itid == 0) {
        unsigned long long* pp = a.prof + (size_t)blockIdx.x * XT_PROF_WORDS;
        for (int i = 0; i < 24; i++) pp[i] = tph[i];
        pp[30] = (unsigned long long)team; pp[31] = (unsigned long long)rank;
    }
#undef XT_PH
}

// ---- fragment-ordered weight image.  One thread per 16-B piece: (workgroup rank, wave, layer, fragment, lane) -> 8 consecutive k of one weight column.
template <int D>
__global__ __launch_bounds__(256) void k_xt_image(const op16_t* __restrict__ w16t, long long layer0, long long layer_stride, int NL, u32x4* __restrict__ img) {
    typedef XtGeo<D> Geo;
    const size_t total = (size_t)XT_TEAM * 4 * NL * Geo::FR * 64;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int lane = (int)(e & 63);
        size_t f = e >> 6;
        const int fi = (int)(f % Geo::FR); f /= Geo::FR;
        const int l = (int)(f % NL); f /= NL;
        const int w = (int)(f & 3), rank = (int)(f >> 2);
        const long long base = layer0 + (long long)l * layer_stride;
        const long long aw = base + 2LL * D, pw = aw + 3LL * D * D + 3LL * D, fw = pw + (long long)D * D + 3LL * D, p2w = fw + 4LL * D * D + 4LL * D;
        long long mat;
        int ld, n0, kb;
        if (fi < Geo::FR_A) {
            const int j = fi / Geo::KBW, k = fi % Geo::KBW;
            mat = aw; ld = D; n0 = rank * Geo::CW_A + j * 16; kb = w * Geo::KBW + k;
        } else if (fi < Geo::FR_A + Geo::FR_P) {
            const int i = fi - Geo::FR_A, j = i / Geo::KBW, k = i % Geo::KBW;
            mat = pw; ld = D; n0 = rank * Geo::CW_P + j * 16; kb = w * Geo::KBW + k;
        } else if (fi < Geo::FR_A + Geo::FR_P + Geo::FR_F) {
            const int i = fi - Geo::FR_A - Geo::FR_P, j = i / Geo::KB, k = i % Geo::KB;
            mat = fw; ld = D; n0 = rank * Geo::CW_F + w * (Geo::CW_F / 4) + j * 16; kb = k;
        } else {
            const int i = fi - Geo::FR_A - Geo::FR_P - Geo::FR_F, c = i / (Geo::NJ_P * 4), r = i % (Geo::NJ_P * 4), j = r / 4, k = r % 4;
            mat = p2w; ld = 4 * D; n0 = rank * Geo::CW_P + j * 16; kb = c * 16 + w * 4 + k;
        }
        const op16_t* src = w16t + mat + (size_t)(n0 + (lane & 15)) * ld + kb * 32 + (lane >> 4) * 8;
        img[e] = *reinterpret_cast<const u32x4*>(src);
    }
}

template <int D, int RING>
int xt_launch_d(const XtArgs& a, int G, size_t lds, hipStream_t st) {
#define XT_GO(G_)                                                                                                                     \
    case G_: {                                                                                                                        \
        static bool attr = false;                                                                                                     \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_decode_xt<D, G_, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((k_decode_xt<D, G_, RING>), dim3(8 * XT_TEAM), dim3(512), lds, st, a);                                     \
    } break;
    switch (G) { XT_GO(2) XT_GO(3) XT_GO(4) XT_GO(5) XT_GO(6) XT_GO(7) XT_GO(8) default: return CC_ERR_SHAPE; }
#undef XT_GO
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

}  // namespace

int64_t xt_image_bytes(int D, int NL) {
    const int fr = D == 1024 ? XtGeo<1024>::FR : D == 512 ? XtGeo<512>::FR : 0;
    if (!fr || NL < 1) return 0;
    return ((int64_t)XT_TEAM * 4 * NL * fr + XT_RING_MAX) * 1024;
}

int xt_build_image(int D, int NL, long long layer0, long long total, const op16_t* w16, op16_t* img, hipStream_t st) {
    if (!xt_image_bytes(D, NL)) return CC_ERR_SHAPE;
    const op16_t* w16t = w16 + total;            // transposed Conv1D weights: [N][K], K contiguous (cc_gpt2_sync_weights)
    const long long stride = 12LL * D * D + 13LL * D;
    if (hipMemsetAsync(reinterpret_cast<char*>(img) + xt_image_bytes(D, NL) - (int64_t)XT_RING_MAX * 1024, 0, (size_t)XT_RING_MAX * 1024, st) != hipSuccess) return CC_ERR_LAUNCH;
    if (D == 1024) hipLaunchKernelGGL((k_xt_image<1024>), dim3(2048), dim3(256), 0, st, w16t, layer0, stride, NL, reinterpret_cast<u32x4*>(img));
    else hipLaunchKernelGGL((k_xt_image<512>), dim3(2048), dim3(256), 0, st, w16t, layer0, stride, NL, reinterpret_cast<u32x4*>(img));
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

static bool xt_prepare(const XtLaunch& L, XtArgs& a, size_t& lds) {
    const int D = L.D, G = L.group, M = L.M;
    if ((D != 512 && D != 1024) || L.H * 64 != D || G < 2 || G > 8 || M % G || L.NL < 1 || !L.wimg) return false;
    static int n_cu = -1;
    if (n_cu < 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
        n_cu = p.multiProcessorCount;
    }
    if (n_cu != 8 * XT_TEAM) return false;
    a = XtArgs{};
    a.w32 = L.w32; a.wimg = reinterpret_cast<const u32x4*>(L.wimg); a.NL = L.NL; a.M = M; a.NG = M / G; a.cpt = (a.NG + 7) / 8;
    a.pos0 = L.pos0; a.ctx_max = L.ctx_max; a.cap = L.cap; a.scale = 0.125f; a.layer0 = L.layer0; a.layer_stride = 12LL * D * D + 13LL * D;
    a.x = L.x; a.x1 = L.x1; a.qkv = L.qkv; a.att = L.att; a.hact = L.hact; a.hf = L.hf; a.kv = L.kv; a.cache_layer = L.cache_layer;
    a.ent = L.ent; a.cnt = L.cnt; a.ctl = L.ctl; a.sticky = L.sticky; a.prof = L.prof;
    a.rt_max = a.cpt * G;
    { static const int aux = []() { const char* e = cc_lab_env("CC_XT_AUX"); return e ? atoi(e) : 16; }(); a.lab_aux = aux; }
    if (a.rt_max > 48) return false;
    a.attn_floats = XT_ATTN_WAVE_BYTES / 4;            // per I/O wave (xt_attn_wave)
    const size_t row_bytes = D == 1024 ? (size_t)XtGeo<1024>::A_ROW + XtGeo<1024>::P_ROW : (size_t)XtGeo<512>::A_ROW + XtGeo<512>::P_ROW;
    const size_t sc = XtGeo<1024>::SC, pm = (size_t)4 * a.rt_max * (D / 32) * 4;          // chunk row stride; mlp.c_proj's partial tiles
    const size_t base = std::max((size_t)a.rt_max * row_bytes, (size_t)4 * a.attn_floats * 4);
    a.nbuf = 3;
    if (std::max(base, 3 * a.rt_max * sc + pm) + 64 + 4 * 128 * 8 + 24 * 8 > 160 * 1024) a.nbuf = 2;
    a.lds_main = (int)((std::max(base, a.nbuf * a.rt_max * sc + pm) + 63) & ~(size_t)63);
    lds = (size_t)a.lds_main + 64 + 4 * 128 * 8 + 24 * 8;
    return lds <= 160 * 1024;
}

bool xt_covers(const XtLaunch& L) {
    XtArgs a;
    size_t lds;
    return xt_prepare(L, a, lds);
}

int decode_layers_xt(const XtLaunch& L, hipStream_t st) {
    XtArgs a;
    size_t lds;
    if (!xt_prepare(L, a, lds) || !L.ctl || !L.sticky) return CC_ERR_SHAPE;
#ifdef CC_EXPERIMENTS
    // lab build: ring depth from the environment (A/B runs; tools/xt_prof.py)
    static const int ring = []() { const char* e = cc_lab_env("CC_XT_RING"); return e ? atoi(e) : 32; }();
    if (L.D == 1024 && L.group == 5 && ring == 8) return xt_launch_d<1024, 8>(a, L.group, lds, st);
    if (L.D == 1024 && L.group == 5 && ring == 16) return xt_launch_d<1024, 16>(a, L.group, lds, st);
    if (L.D == 1024 && L.group == 5 && ring == 24) return xt_launch_d<1024, 24>(a, L.group, lds, st);
#endif
    return L.D == 1024 ? xt_launch_d<1024, 32>(a, L.group, lds, st) : xt_launch_d<512, 16>(a, L.group, lds, st);
}
#else
int64_t xt_image_bytes(int, int) { return 0; }
int xt_build_image(int, int, long long, long long, const op16_t*, op16_t*, hipStream_t) { return CC_ERR_SHAPE; }
bool xt_covers(const XtLaunch&) { return false; }
int decode_layers_xt(const XtLaunch&, hipStream_t) { return CC_ERR_SHAPE; }
#endif
}  // namespace CC_NS
