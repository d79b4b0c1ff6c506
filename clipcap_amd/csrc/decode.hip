// KV-cached caption decode for gfx950: replaces the reference's full GPT-2 re-forward per generated token
// (clipcap/inference/base.py:81) with an O(ctx) step, plus the device-side beam-search update of base.py:84-119.
// Decode is HBM-bound (every weight byte is read once per step); the GEMMs reuse gemm.hip.h, attention over the
// cache is one wave per (row, head, new position).
#include "../../include/clipcap_hip.h"
#ifdef CC_EXPERIMENTS
#include "../../include/clipcap_hip_lab.h"
#endif
#include "gemm_api.h"
#include "kernels.h"
#include "decode_pk.h"
#include "decode_xt.h"

using namespace CC_NS;

#ifndef CC_DEC_SCU
#define CC_DEC_SCU 4   // K rows (score phase) / V rows (PV phase) a lane keeps in flight in k_decode_attn
#endif
#ifndef CC_DEC_PVU
#define CC_DEC_PVU 4
#endif

#define CC_TRY(expr)                 \
    do {                             \
        int _e = (expr);             \
        if (_e != CC_OK) return _e;  \
    } while (0)

namespace {

inline hipStream_t S_(void* s) { return static_cast<hipStream_t>(s); }

// x[r,t,:] += wpe[pos0+t,:]
__global__ void k_add_wpe(const float* __restrict__ xin, const float* __restrict__ wpe, float* __restrict__ x, int R, int Tn, int D, int pos0) {
    const int d4n = D >> 2;
    const size_t total = (size_t)R * Tn * d4n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % d4n), t = (int)((i / d4n) % Tn);
        const float4 a = reinterpret_cast<const float4*>(xin)[i];
        const float4 p = reinterpret_cast<const float4*>(wpe + (size_t)(pos0 + t) * D)[c];
        reinterpret_cast<float4*>(x)[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

// x[r,:] = xin[r,:] + wpe[pos,:] and xn = LayerNorm(x) (layer 0's ln_1) in one launch: the single-position decode step (one wave per row)
__global__ __launch_bounds__(256) void k_add_wpe_ln(const float* __restrict__ xin, const float* __restrict__ wpe, float* __restrict__ x,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, act_t* __restrict__ xn,
                                                    int R, int D, int pos) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    constexpr int MAXV = 8;                      // D <= 2048
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXV; it++) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            const float4 a = *reinterpret_cast<const float4*>(xin + (size_t)row * D + c);
            const float4 p = *reinterpret_cast<const float4*>(wpe + (size_t)pos * D + c);
            v[it] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
            *reinterpret_cast<float4*>(x + (size_t)row * D + c) = v[it];
            s += v[it].x + v[it].y + v[it].z + v[it].w;
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
            act_st4(xn + (size_t)row * D + c, (v[it].x - mu) * rs * g.x + b.x, (v[it].y - mu) * rs * g.y + b.y, (v[it].z - mu) * rs * g.z + b.z,
                    (v[it].w - mu) * rs * g.w + b.w);
        }
    }
}

// attention of the Tn new queries of every row against the cache (ctx = pos0 + Tn, causal): one wave per (r,h,t).
// scores: one key per lane (K row = hd contiguous bf16, 16-B loads).  PV: lane = (key group kg, 8-wide d chunk dc): every V load
// is a 16-B vector, the key loop is 64/(hd/8) times shorter than one-d-per-lane, partial sums meet in wave-private LDS.
// APPEND: the wave also writes its own (row, head, new position) K / V slice into the cache (instead of a separate append launch)
// and reads the keys / values of the NEW positions straight from qkv — other waves' cache writes are not ordered with its reads.
template <bool APPEND>
__global__ __launch_bounds__(256) void k_decode_attn(const act_t* __restrict__ qkv, act_t* __restrict__ kc,
                                                     act_t* __restrict__ vc, const int* __restrict__ row_map, act_t* __restrict__ out,
                                                     int R, int Tn, int H, int hd, int pos0, int ctx_max, float scale) {
    extern __shared__ float psm[];  // per wave: p[ctx_max] | srow[ctx_max] | red[8][hd]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gid = blockIdx.x * 4 + wave;
    if (gid >= R * H * Tn) return;
    const int t = gid % Tn, h = (gid / Tn) % H, r = gid / (Tn * H);
    const int D = H * hd, nkeys = pos0 + t + 1;
    const int per_wave = 2 * ctx_max + 8 * hd;
    float* p = psm + wave * per_wave;
    int* srow = reinterpret_cast<int*>(p + ctx_max);
    float* red = p + 2 * ctx_max;
    const act_t* q = qkv + ((size_t)r * Tn + t) * 3 * D + h * hd;
    // position j of row r lives in cache row row_map[r*ctx_max + j] (beam ancestry table; identity when null)
    const act_t* kb = kc + h * hd;
    const act_t* vb = vc + h * hd;
    if (APPEND) {
        const int c8 = hd >> 3;                            // 16-B chunks per head slice
        if (lane < 2 * c8) {
            const int which = lane / c8, c = lane - which * c8;
            const act_raw8 v = act_ldraw8(q + (which + 1) * D + c * 8);
            act_t* dst = (which ? vc : kc) + ((size_t)r * ctx_max + pos0 + t) * D + h * hd + c * 8;
            act_straw8(dst, v);
        }
    }
    const act_t* knew = qkv + (size_t)r * Tn * 3 * D + D + h * hd;        // K of new position u: knew + u * 3D  (V: + D)
    for (int j = lane; j < nkeys; j += 64) srow[j] = row_map ? row_map[(size_t)r * ctx_max + j] : r;
    float m = -INFINITY;
    const int nchunk = hd >> 3;
    if ((nchunk & (nchunk - 1)) == 0 && nchunk <= 16) {
        // lane = (key group, 16-B chunk of the head slice): one load instruction covers 64 / nchunk whole K rows (full 128-B lines for
        // hd = 64) instead of 16 B of 64 different rows, SC_U of them in flight; the chunk dot products meet by xor-shuffles
        constexpr int SC_U = CC_DEC_SCU;
        const int kgs = 64 / nchunk, skg = lane / nchunk, sdc = lane - skg * nchunk;
        float qf[8];
        act_ld8(q + sdc * 8, qf);
        for (int j0 = 0; j0 < nkeys; j0 += kgs * SC_U) {
            act_raw8 kv[SC_U];
#pragma unroll
            for (int u = 0; u < SC_U; u++) {
                const int j = min(j0 + u * kgs + skg, nkeys - 1);
                const act_t* krow = (APPEND && j >= pos0) ? knew + (size_t)(j - pos0) * 3 * D : kb + ((size_t)srow[j] * ctx_max + j) * D;
                kv[u] = act_ldraw8(krow + sdc * 8);
            }
#pragma unroll
            for (int u = 0; u < SC_U; u++) {
                float b[8], sc = 0.f;
                act_unpack8(kv[u], b);
#pragma unroll
                for (int e = 0; e < 8; e++) sc += qf[e] * b[e];
                for (int o = 1; o < nchunk; o <<= 1) sc += __shfl_xor(sc, o);
                sc *= scale;
                const int j = j0 + u * kgs + skg;
                if (j < nkeys) {
                    if (sdc == 0) p[j] = sc;
                    m = fmaxf(m, sc);
                }
            }
        }
    } else {
        for (int j = lane; j < nkeys; j += 64) {
            float s = 0.f;
            const act_t* krow = (APPEND && j >= pos0) ? knew + (size_t)(j - pos0) * 3 * D : kb + ((size_t)srow[j] * ctx_max + j) * D;
            for (int d = 0; d < hd; d += 8) {
                float a[8], b[8];
                act_ld8(q + d, a);
                act_ld8(krow + d, b);
#pragma unroll
                for (int e = 0; e < 8; e++) s += a[e] * b[e];
            }
            s *= scale;
            p[j] = s;
            m = fmaxf(m, s);
        }
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < nkeys; j += 64) {
        const float e = __expf(p[j] - m);
        p[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    // wave-private LDS: same-wave writes above are visible to the reads below (in-order DS queue)
    const int kgroups = min(8, 64 / nchunk);
    const int kg = lane / nchunk, dc = lane - kg * nchunk;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kg < kgroups) {
        constexpr int PV_U = CC_DEC_PVU;                       // V rows in flight per lane
        for (int j0 = kg; j0 < nkeys; j0 += kgroups * PV_U) {
            act_raw8 vv[PV_U];
            float pj[PV_U];
#pragma unroll
            for (int u = 0; u < PV_U; u++) {
                const int j = min(j0 + u * kgroups, nkeys - 1);
                const act_t* vrow = (APPEND && j >= pos0) ? knew + D + (size_t)(j - pos0) * 3 * D : vb + ((size_t)srow[j] * ctx_max + j) * D;
                vv[u] = act_ldraw8(vrow + dc * 8);
                pj[u] = j0 + u * kgroups < nkeys ? p[j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < PV_U; u++) {
                float v[8];
                act_unpack8(vv[u], v);
#pragma unroll
                for (int e = 0; e < 8; e++) acc[e] += pj[u] * v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) red[kg * hd + dc * 8 + e] = acc[e];
    }
    for (int d = lane; d < hd; d += 64) {
        float o = 0.f;
        for (int g = 0; g < kgroups; g++) o += red[g * hd + d];
        out[((size_t)r * Tn + t) * D + h * hd + d] = f2act(o * inv);
    }
}

// Beam-group form of the single-position attention step (Tn == 1, head dim 64 = every GPT-2 size).  The G beams of a caption share
// their prefix rows and most of their ancestors (tools/decode_union_stats.py: 56 distinct rows against 217 read per group on the
// configs[4] decode), so
//   k_group_union   (once per position, one wave per group) builds the UNION of the (cache row, position) pairs the group's ancestry
//                   tables name: ent = {cache row * ctx_max + position, bit b set when beam b's table names that row}; the G new keys
//                   (one per beam) are the last G entries;
//   k_decode_attn_group (per layer, one 4-wave block per (group, head)) loads every distinct K / V row ONCE and scores it against all G
//                   queries; a beam's softmax runs over the entries whose bit it owns (the others hold -inf), i.e. exactly the keys
//                   k_decode_attn reads for it.  A wave owns 32 entries per pass (128 per block: one pass for the usual union): K as
//                   (key group, 16-B chunk) lanes like k_decode_attn, V one element per lane (a wave instruction = one 128-B row) with
//                   scalar entry loads, and ALL of a pass's K and V loads are in flight together — the chain is list -> rows -> done.
// Any row_map is handled exactly (rows that share nothing give G entries per position); sharing only decides how many rows are read.
// LDS per block: p[cap][8] | wmx[4][8] | wsum[8][8] | red2[8][G][64].
template <int G>
__global__ __launch_bounds__(64) void k_group_union(const int* __restrict__ row_map, int2* __restrict__ ent_g, int* __restrict__ cnt_g, int pos0,
                                                    int ctx_max, int cap, int append, unsigned* __restrict__ zero, int nzero) {
    const int lane = threadIdx.x, s = blockIdx.x, r0 = s * G;
    if (s == 0)                                            // the persistent layer launch that follows starts from cleared arrival counters
        for (int i = lane; i < nzero; i += 64) zero[i] = 0u;
    int2* ent = ent_g + (size_t)s * cap;
    int nU = 0;
    for (int j0 = 0; j0 < pos0; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < pos0;
        int m[G];
#pragma unroll
        for (int b = 0; b < G; b++) m[b] = valid ? (row_map ? row_map[(size_t)(r0 + b) * ctx_max + j] : r0 + b) : -1 - b;
        unsigned lead = 0, mk[G];
#pragma unroll
        for (int b = 0; b < G; b++) {
            unsigned k = 0;
            bool l = valid;
#pragma unroll
            for (int b2 = 0; b2 < G; b2++)
                if (m[b2] == m[b]) { k |= 1u << b2; if (b2 < b) l = false; }
            mk[b] = k;
            if (l) lead |= 1u << b;
        }
        const int cnt = __popc(lead);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        int idx = nU + incl - cnt;
#pragma unroll
        for (int b = 0; b < G; b++)
            if ((lead >> b) & 1) { ent[idx] = make_int2(m[b] * ctx_max + j, (int)mk[b]); idx++; }
        nU += __shfl(incl, 63);
    }
    // the new position: one private key per beam; append: still in qkv (the attention kernel is the one that stores it), marked -1 - b
    if (lane < G) ent[nU + lane] = make_int2(append ? -1 - lane : (r0 + lane) * ctx_max + pos0, 1 << lane);
    nU += G;
    // pad to whole passes of 128 with entries nobody owns (mask 0 -> score -inf -> weight 0) that name a readable row
    const int dummy = append ? -1 : r0 * ctx_max + pos0;
    for (int u = nU + lane; u < ((nU + 127) & ~127); u += 64) ent[u] = make_int2(dummy, 0);
    if (lane == 0) cnt_g[s] = nU;
}

// two consecutive stored elements as one load (k_decode_attn_group's V lanes: 32 lanes x 2 elements = one 64-wide head row)
#if CC_OP == 2
typedef float act_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void act_v2f(const act_v2& v, float& a, float& b) { a = v.x; b = v.y; }
#else
typedef unsigned act_v2;
__device__ __forceinline__ void act_v2f(const act_v2& v, float& a, float& b) { unpack2(v, a, b); }
#endif
typedef const __attribute__((address_space(1))) act_v2* g_v2p;

template <int G>
__global__ __launch_bounds__(256, kX3 ? 3 : 4) void k_decode_attn_group(const act_t* __restrict__ qkv, act_t* __restrict__ kc, act_t* __restrict__ vc,
                                                           const int2* __restrict__ ent_g, const int* __restrict__ cnt_g, act_t* __restrict__ out,
                                                           int H, int pos0, int ctx_max, float scale, int cap, int append) {
    constexpr int HD = 64, KPW = 32;                       // head dim; entries per wave and pass
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* p = gsm;                                        // p[u * 8 + b], u < cap (a multiple of 128)
    float* wmx = p + (size_t)cap * 8;                      // [4][8]
    float* wsm = wmx + 32;                                 // [8][8]  (wave, half)
    float* red2 = wsm + 64;                                // [8][G][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = blockIdx.x / H, h = blockIdx.x - s * H, r0 = s * G;
    const int D = H * HD;
    const int2* __restrict__ ent = ent_g + (size_t)s * cap;
    const act_t* kb = kc + h * HD;
    const act_t* vb = vc + h * HD;
    const act_t* qrow = qkv + (size_t)r0 * 3 * D + h * HD;          // row r0 + b: + b * 3D;  K: + D, V: + 2D
    const int skg = lane >> 3, sdc = lane & 7;
    // pass 0 always exists: its entries, then all of its K and V rows, are requested before anything is waited for
    int2 ek[4];
#pragma unroll
    for (int i = 0; i < 4; i++) ek[i] = ent[w * KPW + i * 8 + skg];
    int2 ev = ent[w * KPW + (lane & 31)];
    const int nU = cnt_g[s];
    if (append && tid < G * 16) {
        const int b = tid >> 4, wq = tid & 15, which = wq >> 3, c = wq & 7;
        const act_raw8 v = act_ldraw8(qrow + (size_t)b * 3 * D + (which + 1) * D + c * 8);
        act_straw8((which ? vc : kc) + ((size_t)(r0 + b) * ctx_max + pos0) * D + h * HD + c * 8, v);
    }
    float qf[G][8], mx[G];
#pragma unroll
    for (int b = 0; b < G; b++) { act_ld8(qrow + (size_t)b * 3 * D + sdc * 8, qf[b]); mx[b] = -INFINITY; }
    const int npass = (nU + 4 * KPW - 1) / (4 * KPW);
    // V lanes: (half = lane >> 5, element pair = lane & 31): one load instruction fetches the rows of two entries (2 x 128 B)
    const int hf = lane >> 5, dp = lane & 31;
#define CC_GRP_VLOAD()                                                                                                                        \
    {                                                                                                                                         \
        const act_t* vrow = ev.x < 0 ? qrow + (size_t)(-1 - ev.x) * 3 * D + 2 * D : vb + (size_t)ev.x * D;                                    \
        const unsigned long long va = reinterpret_cast<unsigned long long>(vrow);                                                             \
        const int valo = (int)(unsigned)va, vahi = (int)(unsigned)(va >> 32);                                                                 \
        _Pragma("unroll") for (int k = 0; k < KPW / 2; k++) {                                                                                 \
            const unsigned lo0 = __builtin_amdgcn_readlane(valo, 2 * k), hi0 = __builtin_amdgcn_readlane(vahi, 2 * k);                        \
            const unsigned lo1 = __builtin_amdgcn_readlane(valo, 2 * k + 1), hi1 = __builtin_amdgcn_readlane(vahi, 2 * k + 1);                \
            const unsigned long long a = ((unsigned long long)(hf ? hi1 : hi0) << 32) | (hf ? lo1 : lo0);                                     \
            vreg[k] = reinterpret_cast<g_v2p>(a)[dp];                                                                                         \
        }                                                                                                                                     \
    }
#define CC_GRP_PV()                                                                                                                           \
    _Pragma("unroll") for (int k = 0; k < KPW / 2; k++) {                                                                                     \
        float v0, v1;                                                                                                                         \
        act_v2f(vreg[k], v0, v1);                                                                                                             \
        const float* pu = p + (size_t)(ub + 2 * k + hf) * 8;                                                                                  \
        const float4 pa = *reinterpret_cast<const float4*>(pu), pb = *reinterpret_cast<const float4*>(pu + 4);                                \
        const float pj[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};                                                                 \
        _Pragma("unroll") for (int b = 0; b < G; b++) { acc[b][0] += pj[b] * v0; acc[b][1] += pj[b] * v1; lsum[b] += pj[b]; }                 \
    }
    act_v2 vreg[KPW / 2];                                  // pass 0's V rows
    for (int pass = 0; pass < npass; pass++) {
        const int ub = pass * 4 * KPW + w * KPW;           // wave-uniform
        if (pass > 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) ek[i] = ent[ub + i * 8 + skg];
        }
        act_raw8 kv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const act_t* krow = ek[i].x < 0 ? qrow + (size_t)(-1 - ek[i].x) * 3 * D + D : kb + (size_t)ek[i].x * D;
            kv[i] = act_ldraw8(krow + sdc * 8);
        }
        if (pass == 0) CC_GRP_VLOAD()
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float kf[8], sc[8];
            act_unpack8(kv[i], kf);
#pragma unroll
            for (int b = 0; b < G; b++) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) a += qf[b][e] * kf[e];
                sc[b] = a;
            }
#pragma unroll
            for (int b = 0; b < G; b++) sc[b] = sum8(sc[b]);
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
    for (int b = 0; b < G; b++) mx[b] = fmaxf(fmaxf(wmx[b], wmx[8 + b]), fmaxf(wmx[16 + b], wmx[24 + b]));     // finite: every beam owns its new key
    // ---- exp of the wave's own entries (only this wave reads them again), then P V with two output elements per lane and half
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
            CC_GRP_VLOAD()
        }
        CC_GRP_PV()
    }
#undef CC_GRP_VLOAD
#undef CC_GRP_PV
#pragma unroll
    for (int b = 0; b < G; b++) *reinterpret_cast<float2*>(red2 + ((w * 2 + hf) * G + b) * HD + 2 * dp) = make_float2(acc[b][0], acc[b][1]);
    if (dp == 0) {
#pragma unroll
        for (int b = 0; b < G; b++) wsm[(w * 2 + hf) * 8 + b] = lsum[b];
    }
    __syncthreads();
    for (int o = tid; o < G * HD; o += 256) {
        const int b = o >> 6, d = o & 63;
        float sum = 0.f, v = 0.f;
#pragma unroll
        for (int x = 0; x < 8; x++) { sum += wsm[x * 8 + b]; v += red2[(x * G + b) * HD + d]; }
        out[(size_t)(r0 + b) * D + h * HD + d] = f2act(v / sum);
    }
}

__global__ void k_kv_reorder(const act_t* __restrict__ src, act_t* __restrict__ dst, const int* __restrict__ map, int R_src, int R_dst,
                             int ctx, int ctx_max, int D, int NL2) {
    const int d8n = D * (int)sizeof(act_t) / 16;      // 16-B vectors per cache row
    const size_t per_row = (size_t)ctx * d8n;
    const size_t total = (size_t)NL2 * R_dst * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i % per_row;
        const int r = (int)((i / per_row) % R_dst), l = (int)(i / (per_row * R_dst));
        const int sr = map[r];
        const uint4* s = reinterpret_cast<const uint4*>(src + ((size_t)l * R_src + sr) * ctx_max * D);
        uint4* d = reinterpret_cast<uint4*>(dst + ((size_t)l * R_dst + r) * ctx_max * D);
        d[e] = s[e];
    }
}

__global__ void k_last_rows(int* __restrict__ map, int R, int Tn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) map[i] = i * Tn + Tn - 1;
}

__global__ void k_embed_tokens(const float* __restrict__ wte, const int* __restrict__ tok, float* __restrict__ out, int R, int D, int rows) {
    const int d4n = D >> 2;
    const size_t total = (size_t)R * d4n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % d4n), r = (int)(i / d4n);
        const int id = min(max(tok[r], 0), rows - 1);       // never reads outside wte (callers validate ids; an out-of-range id must not fault the device)
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(wte + (size_t)id * D)[c];
    }
}

// Gradient of the gather: dwte[tok[r], :] += dout[r, :] (fp32 atomics; rows that share an id accumulate).  Ids clamped like the forward's.
__global__ void k_embed_tokens_bwd(const float* __restrict__ dout, const int* __restrict__ tok, float* __restrict__ dwte, int R, int D, int rows) {
    const size_t total = (size_t)R * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D), r = (int)(i / D);
        const int id = min(max(tok[r], 0), rows - 1);
        __hip_atomic_fetch_add(dwte + (size_t)id * D + d, dout[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Bookkeeping between two beam steps in ONE launch (base.py:104-117: tokens = cat(tokens[src], next), embed(next), cache
// ancestry): row r continues group row g = (r / beam) * beam + src[r].  Replaces ~10 small framework launches per generated token.
__global__ __launch_bounds__(256) void k_beam_advance(int beam, int D, const float* __restrict__ wte, const int* __restrict__ next_tok,
                                                      const int* __restrict__ src, int pos, int ctx_max, const int* __restrict__ map_in,
                                                      int* __restrict__ map_out, int step, int tok_ld, const int* __restrict__ tok_in,
                                                      int* __restrict__ tok_out, float* __restrict__ x_out) {
    const int r = blockIdx.x, tid = threadIdx.x;
    const int g = src ? (r / beam) * beam + src[r] : r;
    int id = next_tok[r];
    if (map_out)
        for (int j = tid; j < ctx_max; j += 256) map_out[(size_t)r * ctx_max + j] = j < pos ? map_in[(size_t)g * ctx_max + j] : r;
    if (tok_out) {
        for (int j = tid; j < step; j += 256) tok_out[(size_t)r * tok_ld + j] = tok_in[(size_t)g * tok_ld + j];
        if (tid == 0) tok_out[(size_t)r * tok_ld + step] = id;
    }
    if (id < 0) id = 0;
    const float4* w = reinterpret_cast<const float4*>(wte + (size_t)id * D);
    for (int c = tid; c < (D >> 2); c += 256) reinterpret_cast<float4*>(x_out + (size_t)r * D)[c] = w[c];
}

// ---- beam step (base.py:84-119) in three small kernels so that all CUs take part --------------------------------
//   k_beam_rowstats : one block per (sample, beam row): max and sum(exp) of logits/temperature
//   k_beam_partial  : grid (sample, chunk): top-`beam` of the length-normalised candidate scores inside one slice of the
//                     flattened beam*V candidate space (thread-local insertion lists + block-wide selection)
//   k_beam_final    : one block per sample merges the chunk winners, then gathers / updates scores, lengths, stopped flags
// Ties resolve to the lowest flat index b*V + v.
constexpr int BEAM_MAX = 16;
constexpr int BEAM_CHUNKS = 16;

__device__ __forceinline__ bool cand_better(float a, int ia, float b, int ib) { return a > b || (a == b && ia < ib); }

template <bool VEC>   // VEC: rows are 16-B aligned (ldl % 4 == 0, aligned base) -> float4 loads
__global__ __launch_bounds__(512) void k_beam_rowstats(const float* __restrict__ logits, size_t ldl, int V, float inv_temp, float* __restrict__ rs) {
    // one pass: every thread keeps a running (max, sum of exp(x - max)) pair, rescaling the sum when its max moves; pairs are merged
    // the same way across lanes and waves (the logits matrix, 64 MB at 320 x 50257, is read once instead of twice)
    __shared__ float redm[8], reds[8];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* lg = logits + (size_t)row * ldl;
    const int V4 = VEC ? (V >> 2) : 0;
    float m = -INFINITY, sum = 0.f;
    auto add4 = [&](float a, float b, float c, float d) {
        const float mx = fmaxf(fmaxf(a, b), fmaxf(c, d));
        if (mx > m) { sum *= expf(m - mx); m = mx; }              // expf(-inf) = 0 on the first group
        sum += expf(a - m) + expf(b - m) + expf(c - m) + expf(d - m);
    };
    for (int v = tid; v < V4; v += 512) {
        const float4 x = reinterpret_cast<const float4*>(lg)[v];
        add4(x.x * inv_temp, x.y * inv_temp, x.z * inv_temp, x.w * inv_temp);
    }
    for (int v = V4 * 4 + tid; v < V; v += 512) {
        const float x = lg[v] * inv_temp;
        if (x > m) { sum *= expf(m - x); m = x; }
        sum += expf(x - m);
    }
    auto merge = [&](float om, float os) {
        const float mx = fmaxf(m, om);
        if (mx == -INFINITY) return;                               // both empty
        sum = sum * expf(m - mx) + os * expf(om - mx);
        m = mx;
    };
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64), os = __shfl_xor(sum, o, 64);
        merge(om, os);
    }
    if ((tid & 63) == 0) { redm[tid >> 6] = m; reds[tid >> 6] = sum; }
    __syncthreads();
    if (tid == 0) {
        m = redm[0]; sum = reds[0];
        for (int i = 1; i < 8; i++) merge(redm[i], reds[i]);
        rs[2 * row] = m;
        rs[2 * row + 1] = sum;
    }
}

// block-wide selection of the `beam` best of ncand (value, index) candidates held in LDS; results in sel_v / sel_i
__device__ __forceinline__ void select_top(float* cval, int* cidx, int ncand, int beam, float* red, int* redi, float* sel_v, int* sel_i, int tid,
                                           int nthreads) {
    for (int k = 0; k < beam; k++) {
        float bv = -INFINITY;
        int bi = 0x7fffffff, bp = -1;
        for (int c = tid; c < ncand; c += nthreads)
            if (cidx[c] != 0x7fffffff && (bp < 0 || cand_better(cval[c], cidx[c], bv, bi))) { bv = cval[c]; bi = cidx[c]; bp = c; }
        red[tid] = bv; redi[tid] = bp;
        __syncthreads();
        for (int o = nthreads >> 1; o > 0; o >>= 1) {
            if (tid < o) {
                const int pa = redi[tid], pb = redi[tid + o];
                if (pb >= 0 && (pa < 0 || cand_better(red[tid + o], cidx[pb], red[tid], cidx[pa]))) { red[tid] = red[tid + o]; redi[tid] = pb; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            const int pos = redi[0];
            if (pos >= 0) { sel_v[k] = cval[pos]; sel_i[k] = cidx[pos]; cidx[pos] = 0x7fffffff; }
            else { sel_v[k] = -INFINITY; sel_i[k] = 0x7fffffff; }
        }
        __syncthreads();
    }
}

// TB = compile-time beam width (0: run-time width, lists in scratch memory — the slow fallback for beam > 8).
// Each candidate's exact value needs expf + divide + logf (the reference's softmax().log() arithmetic); almost all of the 250 k
// candidates lose against the thread's current worst kept value, so a cheap bound (x t - m - log(sum), equal up to ~1e-6) with a
// 1e-3 margin decides whether the exact value is evaluated at all.
template <int TB>
__global__ __launch_bounds__(256) void k_beam_partial(const float* __restrict__ logits, size_t ldl, int beam_rt, int V, float inv_temp, int first,
                                                      const float* __restrict__ rs, const float* __restrict__ scores,
                                                      const float* __restrict__ seq_len, const unsigned char* __restrict__ stopped,
                                                      float* __restrict__ pval, int* __restrict__ pidx) {
    __shared__ float red[256];
    __shared__ int redi[256];
    __shared__ float cval[TB > 0 ? 1 : 256 * BEAM_MAX];          // candidate arrays: run-time-width fallback only
    __shared__ int cidx[TB > 0 ? 1 : 256 * BEAM_MAX];
    __shared__ float sel_v[BEAM_MAX];
    __shared__ int sel_i[BEAM_MAX];
    constexpr int LB = TB > 0 ? TB : BEAM_MAX;
    const int beam = TB > 0 ? TB : beam_rt;
    const int s = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
    const int nrows = first ? 1 : beam;
    const int total = nrows * V;
    const int per = (total + BEAM_CHUNKS - 1) / BEAM_CHUNKS;
    const int lo = ch * per, hi = min(total, lo + per);
    const float* lg = logits + (size_t)s * beam * ldl;
    // a chunk (total / 16 candidates) spans at most two beam rows when V >= per; handled generally by walking row segments.
    // scan(f): f(idx, v, seg, raw logit) for every candidate of the chunk, SU logits fetched per thread before any is looked at
    struct Seg { bool st; float m, sum, lsum, sc, inv_len, len; };
    auto scan = [&](auto&& f) {
        for (int b = lo / V; b < nrows && b * V < hi; b++) {
            const int seg_lo = max(lo, b * V), seg_hi = min(hi, (b + 1) * V);
            Seg g;
            g.st = !first && stopped[s * beam + b];
            g.m = rs[2 * (s * beam + b)]; g.sum = rs[2 * (s * beam + b) + 1];
            g.lsum = logf(g.sum);
            g.sc = first ? 0.f : scores[s * beam + b];
            g.len = first ? 1.f : (seq_len[s * beam + b] + (g.st ? 0.f : 1.f));
            g.inv_len = 1.0f / g.len;
            const float* row = lg + (size_t)b * ldl - (size_t)b * V;      // row[idx] = lg[b * ldl + (idx - b V)]
            constexpr int SU = 8;
            for (int idx0 = seg_lo + tid; idx0 < seg_hi; idx0 += 256 * SU) {
                float xs[SU];
#pragma unroll
                for (int u = 0; u < SU; u++) xs[u] = g.st ? 0.f : row[min(idx0 + u * 256, seg_hi - 1)];
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const int idx = idx0 + u * 256;
                    if (idx < seg_hi) f(idx, idx - b * V, g, xs[u]);
                }
            }
        }
    };
    // cheap image of a candidate's value: exact for stopped beams (base.py:96-101: only token 0 continues a stopped beam), within
    // ~1e-6 of the reference's softmax().log() arithmetic otherwise (x t - m - log(sum) instead of log(exp(x t - m) / sum))
    auto key_of = [&](int v, const Seg& g, float xraw) -> float {
        if (g.st) return v == 0 ? (first ? 0.f : (g.sc + 0.f) / g.len) : -INFINITY;
        const float x = xraw * inv_temp - g.m;
        return first ? (x - g.lsum) : (g.sc + (x - g.lsum)) * g.inv_len;
    };
    auto val_of = [&](int v, const Seg& g, float xraw) -> float {
        if (g.st) return v == 0 ? (first ? 0.f : (g.sc + 0.f) / g.len) : -INFINITY;
        const float lp = logf(expf(xraw * inv_temp - g.m) / g.sum);                                           // softmax().log()
        return first ? lp : (g.sc + lp) / g.len;                                                               // base.py:99-101
    };
    const int lane = tid & 63, wv = tid >> 6;
    bool done = false;
    if constexpr (TB > 0) {
        // Two passes instead of a sorted list per thread (with per-thread lists some lane of a wave inserts in almost every iteration,
        // so every wave ran the ~100-instruction exact-value + insertion path for all of its candidates):
        //   1. thread maxima of the cheap key; the TB-th largest thread maximum is a lower bound of the chunk's TB-th best value;
        //   2. only candidates within 1e-3 of that bound get the exact value and go to a small LDS list (a handful per block);
        //   3. TB rounds of block arg-max over the list (ties -> lowest flat index).
        constexpr int FCAP = 1024;
        __shared__ float fcv[FCAP];
        __shared__ int fci[FCAP];
        __shared__ int fcount;
        float tmax = -INFINITY;
        scan([&](int, int v, const Seg& g, float xraw) { tmax = fmaxf(tmax, key_of(v, g, xraw)); });
        float thr = -INFINITY;
        if (tid == 0) fcount = 0;
        for (int k = 0; k < TB; k++) {
            float bv = tmax;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) bv = fmaxf(bv, __shfl_xor(bv, o, 64));
            if (lane == 0) red[(k & 1) * 4 + wv] = bv;
            __syncthreads();
            thr = fmaxf(fmaxf(red[(k & 1) * 4], red[(k & 1) * 4 + 1]), fmaxf(red[(k & 1) * 4 + 2], red[(k & 1) * 4 + 3]));
            if (tmax == thr) tmax = -INFINITY;            // equal maxima leave together: the bound only gets lower (still valid)
        }
        scan([&](int idx, int v, const Seg& g, float xraw) {
            const float key = key_of(v, g, xraw);
            if (key > -INFINITY && key + 1e-3f >= thr) {
                const int pos = atomicAdd(&fcount, 1);
                if (pos < FCAP) { fcv[pos] = val_of(v, g, xraw); fci[pos] = idx; }
            }
        });
        __syncthreads();
        const int n = fcount;
        if (n <= FCAP) {
            for (int k = 0; k < TB; k++) {
                float bv = -INFINITY;
                int bi = 0x7fffffff;
                for (int c = tid; c < n; c += 256)
                    if (fci[c] != 0x7fffffff && cand_better(fcv[c], fci[c], bv, bi)) { bv = fcv[c]; bi = fci[c]; }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ov = __shfl_xor(bv, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
                }
                if (lane == 0) { red[(k & 1) * 4 + wv] = bv; redi[(k & 1) * 4 + wv] = bi; }
                __syncthreads();
                bv = red[(k & 1) * 4]; bi = redi[(k & 1) * 4];
#pragma unroll
                for (int w = 1; w < 4; w++) {
                    const float ov = red[(k & 1) * 4 + w];
                    const int oi = redi[(k & 1) * 4 + w];
                    if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
                }
                for (int c = tid; c < n; c += 256)
                    if (fci[c] == bi) fci[c] = 0x7fffffff;                  // flat indices are unique: one owner; visible after the next barrier
                if (tid == 0) { sel_v[k] = bi != 0x7fffffff ? bv : -INFINITY; sel_i[k] = bi; }
                __syncthreads();
            }
            done = true;
        }
    }
    if (!done) {
        // sorted insertion list per thread: any width (TB = 0: run-time width), and the overflow path of the list above (e.g. all logits equal)
        float lv[LB];
        int li[LB];
#pragma unroll
        for (int k = 0; k < LB; k++) { lv[k] = -INFINITY; li[k] = 0x7fffffff; }
        scan([&](int idx, int v, const Seg& g, float xraw) {
            if (!g.st) {
                const float worst = TB > 0 ? lv[LB - 1] : lv[beam - 1];
                if (key_of(v, g, xraw) + 1e-3f < worst) return;
            }
            const float val = val_of(v, g, xraw);
            if constexpr (TB > 0) {
                if (cand_better(val, idx, lv[LB - 1], li[LB - 1])) {
                    lv[LB - 1] = val; li[LB - 1] = idx;
#pragma unroll
                    for (int k = LB - 1; k > 0; k--) {
                        if (cand_better(lv[k], li[k], lv[k - 1], li[k - 1])) {
                            const float tv = lv[k]; lv[k] = lv[k - 1]; lv[k - 1] = tv;
                            const int ti = li[k]; li[k] = li[k - 1]; li[k - 1] = ti;
                        }
                    }
                }
            } else {
                if (cand_better(val, idx, lv[beam - 1], li[beam - 1])) {
                    int k = beam - 1;
                    while (k > 0 && cand_better(val, idx, lv[k - 1], li[k - 1])) { lv[k] = lv[k - 1]; li[k] = li[k - 1]; k--; }
                    lv[k] = val; li[k] = idx;
                }
            }
        });
        if constexpr (TB > 0) {
            // every thread's list is sorted, so the block's next best is the best list HEAD: wave arg-max by shuffles, the four wave
            // winners meet in LDS (one barrier per round), the owner pops its list
            for (int k = 0; k < TB; k++) {
                float bv = lv[0];
                int bi = li[0];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ov = __shfl_xor(bv, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
                }
                if (lane == 0) { red[(k & 1) * 4 + wv] = bv; redi[(k & 1) * 4 + wv] = bi; }
                __syncthreads();
                bv = red[(k & 1) * 4]; bi = redi[(k & 1) * 4];
#pragma unroll
                for (int w = 1; w < 4; w++) {
                    const float ov = red[(k & 1) * 4 + w];
                    const int oi = redi[(k & 1) * 4 + w];
                    if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
                }
                if (li[0] == bi && bi != 0x7fffffff) {              // flat indices are unique: exactly one owner
#pragma unroll
                    for (int q = 0; q + 1 < LB; q++) { lv[q] = lv[q + 1]; li[q] = li[q + 1]; }
                    lv[LB - 1] = -INFINITY; li[LB - 1] = 0x7fffffff;
                }
                if (tid == 0) { sel_v[k] = bi != 0x7fffffff ? bv : -INFINITY; sel_i[k] = bi; }
            }
            __syncthreads();
        } else {
            for (int k = 0; k < beam; k++) { cval[tid * beam + k] = lv[k]; cidx[tid * beam + k] = li[k]; }
            __syncthreads();
            select_top(cval, cidx, 256 * beam, beam, red, redi, sel_v, sel_i, tid, 256);
        }
    }
    if (tid < beam) {
        pval[((size_t)s * BEAM_CHUNKS + ch) * beam + tid] = sel_v[tid];
        pidx[((size_t)s * BEAM_CHUNKS + ch) * beam + tid] = sel_i[tid];
    }
}

__global__ __launch_bounds__(64) void k_beam_final(int beam, int V, int first, int stop_token, const float* __restrict__ pval,
                                                   const int* __restrict__ pidx, float* __restrict__ scores, float* __restrict__ seq_len,
                                                   unsigned char* __restrict__ stopped, int* __restrict__ next_tok, int* __restrict__ src_row) {
    __shared__ float red[64];
    __shared__ int redi[64];
    __shared__ float cval[BEAM_CHUNKS * BEAM_MAX];
    __shared__ int cidx[BEAM_CHUNKS * BEAM_MAX];
    __shared__ float sel_v[BEAM_MAX];
    __shared__ int sel_i[BEAM_MAX];
    __shared__ float ns_[BEAM_MAX], nl_[BEAM_MAX];
    __shared__ int hs_[BEAM_MAX];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int ncand = BEAM_CHUNKS * beam;
    for (int c = tid; c < ncand; c += 64) { cval[c] = pval[(size_t)s * ncand + c]; cidx[c] = pidx[(size_t)s * ncand + c]; }
    __syncthreads();
    select_top(cval, cidx, ncand, beam, red, redi, sel_v, sel_i, tid, 64);
    if (tid < beam) {      // gather / update state (base.py:86-119)
        const int idx = sel_i[tid];
        const int b = idx / V, v = idx % V;
        if (first) { nl_[tid] = 1.f; ns_[tid] = sel_v[tid]; hs_[tid] = 0; }
        else {
            const bool st = stopped[s * beam + b];
            nl_[tid] = seq_len[s * beam + b] + (st ? 0.f : 1.f);
            ns_[tid] = sel_v[tid] * nl_[tid];      // scores = scores_sum_average * seq_lengths (base.py:114)
            hs_[tid] = st;
        }
        hs_[tid] |= (v == stop_token) ? 1 : 0;
        next_tok[s * beam + tid] = v;
        src_row[s * beam + tid] = b;
    }
    __syncthreads();
    if (tid < beam) {
        scores[s * beam + tid] = ns_[tid];
        seq_len[s * beam + tid] = nl_[tid];
        stopped[s * beam + tid] = (unsigned char)hs_[tid];
    }
}

// ---- beam step in ONE kernel, fed by the lm_head epilogue's partials (gemm.hip.h EpiLogits) ----------------------------------
// One block per sample.  The row statistics (max, sum of exp) come from the per-(row, 64-column block) partials — no pass over the
// logits; every 64-column block is bounded by the key of its maximum, the TB-th largest bound is a lower bound of the sample's TB-th
// best candidate, and only the blocks whose bound reaches it (a handful) are read from the logits matrix at all.  Same arithmetic
// (softmax().log() as the reference writes it), same tie rule (lowest flat index) and same state update as the three-kernel path,
// which stays as the fallback for temperature != 1, run-time beam widths and candidate-list overflow (e.g. all logits equal).
template <int TB, int PER>      // PER: (row, block) partials per thread = ceil(TB * ceil(V / 64) / 256) at most (checked by the host)
__global__ __launch_bounds__(256) void k_beam_fused(const float* __restrict__ logits, size_t ldl, int V, int npart, const float* __restrict__ pmax,
                                                    const float* __restrict__ psum, int first, int stop_token, float* __restrict__ scores,
                                                    float* __restrict__ seq_len, unsigned char* __restrict__ stopped, int* __restrict__ next_tok,
                                                    int* __restrict__ src_row) {
    constexpr int FCAP = 1024, SCAP = 512;
    __shared__ float red[8];
    __shared__ int redi[8];
    __shared__ float rowred[4][TB];
    __shared__ float s_m[TB], s_lsum[TB], s_sum[TB], s_sc[TB], s_len[TB];
    __shared__ int s_st[TB];
    __shared__ float fcv[FCAP];
    __shared__ int fci[FCAP];
    __shared__ int surv[SCAP];
    __shared__ int fcount, scount;
    __shared__ float sel_v[TB];
    __shared__ int sel_i[TB];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nrows = first ? 1 : TB;
    const int nblk = (V + 63) >> 6;
    // 1. every partial of the sample's rows is requested up front (entry i = tid + 256 e: row i / nblk, block i % nblk) — one round trip
    //    instead of two dependent ones per row — and the row statistics m = max_j pmax, sum = sum_j psum exp(pmax - m) come from registers
    const int total = nrows * nblk;
    float pm[PER], ps[PER];
#pragma unroll
    for (int e = 0; e < PER; e++) {
        const int i = tid + 256 * e;
        const int b = min(i, total - 1) / nblk, j = min(i, total - 1) - b * nblk;
        const size_t at = (size_t)(s * TB + b) * npart + j;
        const float a = pmax[at], c = psum[at];
        pm[e] = i < total ? a : -INFINITY;
        ps[e] = i < total ? c : 0.f;
    }
    float lm[TB];
#pragma unroll
    for (int b = 0; b < TB; b++) lm[b] = -INFINITY;
#pragma unroll
    for (int e = 0; e < PER; e++) {
        const int b = min(tid + 256 * e, total - 1) / nblk;
#pragma unroll
        for (int q = 0; q < TB; q++) lm[q] = (q == b) ? fmaxf(lm[q], pm[e]) : lm[q];
    }
#pragma unroll
    for (int b = 0; b < TB; b++) {
        const float m = wave_max(lm[b]);
        if (lane == 0) rowred[wv][b] = m;
    }
    __syncthreads();
    float ls[TB];
#pragma unroll
    for (int b = 0; b < TB; b++) {
        lm[b] = fmaxf(fmaxf(rowred[0][b], rowred[1][b]), fmaxf(rowred[2][b], rowred[3][b]));
        ls[b] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; e++) {
        const int b = min(tid + 256 * e, total - 1) / nblk;
        float mb = -INFINITY;
#pragma unroll
        for (int q = 0; q < TB; q++) mb = (q == b) ? lm[q] : mb;
        const float t = pm[e] != -INFINITY ? ps[e] * expf(pm[e] - mb) : 0.f;
#pragma unroll
        for (int q = 0; q < TB; q++) ls[q] += (q == b) ? t : 0.f;
    }
#pragma unroll
    for (int b = 0; b < TB; b++) {
        const float t = wave_sum(ls[b]);
        if (lane == 0) rowred[wv][b] = t;
    }
    __syncthreads();
    if (tid < nrows) {
        const int b = tid;
        const float t = (rowred[0][b] + rowred[1][b]) + (rowred[2][b] + rowred[3][b]);
        const bool st = !first && stopped[s * TB + b];
        float mb = -INFINITY;
#pragma unroll
        for (int q = 0; q < TB; q++) mb = (q == b) ? lm[q] : mb;
        s_m[b] = mb; s_sum[b] = t; s_lsum[b] = logf(t); s_st[b] = st;
        s_sc[b] = first ? 0.f : scores[s * TB + b];
        s_len[b] = first ? 1.f : (seq_len[s * TB + b] + (st ? 0.f : 1.f));
    }
    if (tid == 0) { fcount = 0; scount = 0; }
    __syncthreads();
    // cheap image of a candidate's value (monotone in the logit), and the exact value (base.py:96-101)
    auto key_of = [&](int b, int v, float x) -> float {
        if (s_st[b]) return v == 0 ? (first ? 0.f : (s_sc[b] + 0.f) / s_len[b]) : -INFINITY;
        const float d = (x - s_m[b]) - s_lsum[b];
        return first ? d : (s_sc[b] + d) * (1.0f / s_len[b]);
    };
    auto val_of = [&](int b, int v, float x) -> float {
        if (s_st[b]) return v == 0 ? (first ? 0.f : (s_sc[b] + 0.f) / s_len[b]) : -INFINITY;
        const float lp = logf(expf(x - s_m[b]) / s_sum[b]);
        return first ? lp : (s_sc[b] + lp) / s_len[b];
    };
    // 2. bound of every (row, block): key of the block maximum (a stopped row: only token 0, i.e. block 0)
    float bk[PER];
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < PER; e++) {
        const int i = tid + 256 * e;
        const int b = min(i, total - 1) / nblk, j = min(i, total - 1) - b * nblk;
        const float k = s_st[b] ? (j == 0 ? key_of(b, 0, 0.f) : -INFINITY) : key_of(b, 1, pm[e]);
        bk[e] = i < total ? k : -INFINITY;
        tmax = fmaxf(tmax, bk[e]);
    }
    float thr = -INFINITY;
    for (int k = 0; k < TB; k++) {
        float bv = wave_max(tmax);
        if (lane == 0) red[(k & 1) * 4 + wv] = bv;
        __syncthreads();
        thr = fmaxf(fmaxf(red[(k & 1) * 4], red[(k & 1) * 4 + 1]), fmaxf(red[(k & 1) * 4 + 2], red[(k & 1) * 4 + 3]));
        if (tmax == thr) tmax = -INFINITY;            // equal maxima leave together: the bound only gets lower (still valid)
    }
    // 3. surviving blocks -> LDS list
#pragma unroll
    for (int e = 0; e < PER; e++) {
        if (bk[e] > -INFINITY && bk[e] + 1e-3f >= thr) {
            const int pos = atomicAdd(&scount, 1);
            if (pos < SCAP) surv[pos] = tid + 256 * e;
        }
    }
    __syncthreads();
    const int ns = scount;
    bool ok = ns <= SCAP;
    if (ok) {       // 4. one wave per surviving block: its 64 logits, exact values of the candidates within 1e-3 of the bound
        for (int q = wv; q < ns; q += 4) {
            const int i = surv[q], b = i / nblk, j = i - b * nblk, v = j * 64 + lane;
            if (v < V) {
                const float x = s_st[b] ? 0.f : logits[(size_t)(s * TB + b) * ldl + v];
                const float key = key_of(b, v, x);
                if (key > -INFINITY && key + 1e-3f >= thr) {
                    const int pos = atomicAdd(&fcount, 1);
                    if (pos < FCAP) { fcv[pos] = val_of(b, v, x); fci[pos] = b * V + v; }
                }
            }
        }
        __syncthreads();
        ok = fcount <= FCAP;
    }
    if (!ok) {
        // block-uniform overflow path (degenerate inputs, e.g. all logits equal): TB rounds of a full scan, each picking the best candidate
        // that comes strictly after the previous pick in the (value descending, flat index ascending) order.  Slow and exact.
        float pv = INFINITY;
        int pi = -1;
        for (int k = 0; k < TB; k++) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int b = 0; b < nrows; b++)
                for (int v = tid; v < V; v += 256) {
                    const float x = s_st[b] ? 0.f : logits[(size_t)(s * TB + b) * ldl + v];
                    const float val = val_of(b, v, x);
                    const int idx = b * V + v;
                    if ((pi < 0 || cand_better(pv, pi, val, idx)) && cand_better(val, idx, bv, bi)) { bv = val; bi = idx; }
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) { red[(k & 1) * 4 + wv] = bv; redi[(k & 1) * 4 + wv] = bi; }
            __syncthreads();
            bv = red[(k & 1) * 4]; bi = redi[(k & 1) * 4];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                const float ov = red[(k & 1) * 4 + w];
                const int oi = redi[(k & 1) * 4 + w];
                if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
            }
            pv = bv; pi = bi;
            if (tid == 0) { sel_v[k] = bi != 0x7fffffff ? bv : -INFINITY; sel_i[k] = bi; }
        }
        __syncthreads();
    }
    const int n = ok ? fcount : 0;
    for (int k = 0; ok && k < TB; k++) {      // 5. TB rounds of block arg-max over the list (ties -> lowest flat index)
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = tid; c < n; c += 256)
            if (fci[c] != 0x7fffffff && cand_better(fcv[c], fci[c], bv, bi)) { bv = fcv[c]; bi = fci[c]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { red[(k & 1) * 4 + wv] = bv; redi[(k & 1) * 4 + wv] = bi; }
        __syncthreads();
        bv = red[(k & 1) * 4]; bi = redi[(k & 1) * 4];
#pragma unroll
        for (int w = 1; w < 4; w++) {
            const float ov = red[(k & 1) * 4 + w];
            const int oi = redi[(k & 1) * 4 + w];
            if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        for (int c = tid; c < n; c += 256)
            if (fci[c] == bi) fci[c] = 0x7fffffff;
        if (tid == 0) { sel_v[k] = bi != 0x7fffffff ? bv : -INFINITY; sel_i[k] = bi; }
        __syncthreads();
    }
    if (tid < TB) {      // 6. gather / update state (base.py:86-119), as k_beam_final
        const int idx = sel_i[tid];
        const int b = idx / V, v = idx % V;
        float nl, nsc;
        int hs;
        if (first) { nl = 1.f; nsc = sel_v[tid]; hs = 0; }
        else {
            nl = s_len[b];                          // = seq_len[b] + (stopped ? 0 : 1), read before any state was written
            nsc = sel_v[tid] * nl;                  // scores = scores_sum_average * seq_lengths (base.py:114)
            hs = s_st[b];
        }
        hs |= (v == stop_token) ? 1 : 0;
        next_tok[s * TB + tid] = v;
        src_row[s * TB + tid] = b;
        scores[s * TB + tid] = nsc;
        seq_len[s * TB + tid] = nl;
        stopped[s * TB + tid] = (unsigned char)hs;
    }
}

struct DecWS {
    float *x, *x1;
    act_t *xn, *qkv, *att, *hact, *hf;
    float *meanf, *rstdf;
    int* last;
    float* scratch;
    size_t scratch_bytes;
    unsigned long long* pk_prof;   // [256][21] profile of the persistent layer launch (CC_PK_PROF=1)
    unsigned* pk_ctr;      // persistent layer launch (decode_pk.hip): arrival counters + error word
    unsigned* xt_ctl;      // XCD-team engine (decode_xt.hip): control words (zeroed before every launch) + one sticky error word behind them
    unsigned long long* xt_prof;
    int2* grp_ent;         // beam-group attention: union list [R / group][group * (pos0 + 1)] + entry counts (k_group_union)
    int* grp_cnt;
    size_t grp_ents;
    char* x3;              // bf16x3 build: operand-image scratch (gemm_api.h)
    size_t x3_bytes;
    size_t bytes;
};
void dec_carve(const cc_gpt2_cfg* c, int R, int Tn, void* ws, DecWS& w) {
    char* base = static_cast<char*>(ws);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        off = (off + 255) & ~size_t(255);
        char* r = base ? base + off : nullptr;
        off += bytes;
        return r;
    };
    const size_t M = (size_t)R * Tn, D = c->D;
    w.x = (float*)take(M * D * 4);
    w.x1 = (float*)take(M * D * 4);
    w.xn = (act_t*)take(M * D * sizeof(act_t));
    w.qkv = (act_t*)take(M * 3 * D * sizeof(act_t));
    w.att = (act_t*)take(M * D * sizeof(act_t));
    w.hact = (act_t*)take(M * 4 * D * sizeof(act_t));
    w.hf = (act_t*)take((size_t)R * D * sizeof(act_t));
    w.meanf = (float*)take((size_t)R * 4);
    w.rstdf = (float*)take((size_t)R * 4);
    w.last = (int*)take((size_t)R * 4);
    w.scratch_bytes = (size_t)8 * M * 4 * D * 4;   // up to 8 K-slices of the widest (4D) output
    w.scratch = (float*)take(w.scratch_bytes);
    w.pk_ctr = (unsigned*)take((size_t)PK_CTR_WORDS * 4);
    w.pk_prof = (unsigned long long*)take((size_t)256 * 21 * 8);
    w.xt_ctl = (unsigned*)take((size_t)(XT_CTL_WORDS + 16) * 4);
    w.xt_prof = (unsigned long long*)take((size_t)256 * XT_PROF_WORDS * 8);
    w.grp_ents = Tn == 1 ? (size_t)R * c->NPOS + (size_t)R * 128 : 0;
    w.grp_ent = (int2*)take(w.grp_ents * sizeof(int2));
    w.grp_cnt = (int*)take((size_t)R * 4);
    w.x3_bytes = kX3 ? ((M * 3 * 4 * D * sizeof(op16_t) + 255) & ~size_t(255)) : 0;      // the deepest A image: mlp.c_proj, K = 4D
    w.x3 = kX3 ? take(w.x3_bytes) : nullptr;
    w.bytes = (off + 255) & ~size_t(255);
}

bool cfg_ok(const cc_gpt2_cfg* c) {
    return c && (c->op_dtype == CC_OP) && c->D > 0 && c->H > 0 && c->NL > 0 && c->V > 0 && c->Vp >= c->V && (c->Vp % 128) == 0 && (c->D % 8) == 0 && (c->D % c->H) == 0 &&
           ((c->D / c->H) % 8) == 0;
}

}  // namespace

extern "C" {

int64_t CC_API(cc_decode_ws_bytes)(const cc_gpt2_cfg* cfg, int32_t R, int32_t Tnew) {
    if (!cfg_ok(cfg) || R <= 0 || Tnew <= 0) return CC_ERR_SHAPE;
    DecWS w;
    dec_carve(cfg, R, Tnew, nullptr, w);
    return (int64_t)w.bytes;
}

int64_t CC_API(cc_decode_part_floats)(const cc_gpt2_cfg* cfg, int32_t R) {
    if (!cfg_ok(cfg) || R <= 0) return CC_ERR_SHAPE;
    const int Ns = std::min(cfg->Vp, (cfg->V + 7) / 8 * 8);
    return (int64_t)2 * R * ((Ns + 63) / 64);
}

// the decode step behind every entry point; wimg / wteam: lab build only (include/clipcap_hip_lab.h: cc_decode_fwd_x), NULL in the product
static int decode_fwd_impl(const cc_gpt2_cfg* c, int32_t R, int32_t Tn, int32_t pos0, int32_t ctx_max, const float* w32, const uint16_t* w16, const uint16_t* wimg,
                           const uint16_t* wteam, const float* x, uint16_t* kv, const int32_t* row_map, int32_t group, void* ws, float* logits, int64_t ldl, float* lpart,
                           void* stream);
int CC_API(cc_decode_fwd_g)(const cc_gpt2_cfg* c, int32_t R, int32_t Tn, int32_t pos0, int32_t ctx_max, const float* w32, const uint16_t* w16,
                    const float* x, uint16_t* kv, const int32_t* row_map, int32_t group, void* ws, float* logits, int64_t ldl, float* lpart, void* stream) {
    return decode_fwd_impl(c, R, Tn, pos0, ctx_max, w32, w16, nullptr, nullptr, x, kv, row_map, group, ws, logits, ldl, lpart, stream);
}

int CC_API(cc_decode_fwd)(const cc_gpt2_cfg* c, int32_t R, int32_t Tn, int32_t pos0, int32_t ctx_max, const float* w32, const uint16_t* w16,
                  const float* x, uint16_t* kv, const int32_t* row_map, void* ws, float* logits, int64_t ldl, void* stream) {
    return CC_API(cc_decode_fwd_g)(c, R, Tn, pos0, ctx_max, w32, w16, x, kv, row_map, 1, ws, logits, ldl, nullptr, stream);
}

int CC_API(cc_decode_fwd_p)(const cc_gpt2_cfg* c, int32_t R, int32_t Tn, int32_t pos0, int32_t ctx_max, const float* w32, const uint16_t* w16,
                    const float* x, uint16_t* kv, const int32_t* row_map, void* ws, float* logits, int64_t ldl, float* lpart, void* stream) {
    return CC_API(cc_decode_fwd_g)(c, R, Tn, pos0, ctx_max, w32, w16, x, kv, row_map, 1, ws, logits, ldl, lpart, stream);
}

#ifdef CC_EXPERIMENTS      // ---- lab build only: the entry points of include/clipcap_hip_lab.h ----
// fragment-ordered images of the four GEMM weights of every block, at their arena offsets
int64_t CC_API(cc_decode_image_bytes)(const cc_gpt2_cfg* c) {
    if (!cfg_ok(c) || kX3 || (c->D % 64)) return 0;
    const int64_t D = c->D;
    return 2 * ((int64_t)c->Vp * D + (int64_t)c->NPOS * D + (int64_t)c->NL * (12 * D * D + 13 * D) + 2 * D);
}

int CC_API(cc_decode_image)(const cc_gpt2_cfg* c, const uint16_t* w16, uint16_t* wimg, void* stream) {
    if (!cfg_ok(c) || !w16 || !wimg) return CC_ERR_ARG;
    if (kX3 || (c->D % 64)) return CC_ERR_SHAPE;
    const int64_t D = c->D;
    const int64_t layer0 = (int64_t)c->Vp * D + (int64_t)c->NPOS * D;
    const int64_t total = layer0 + (int64_t)c->NL * (12 * D * D + 13 * D) + 2 * D;
    const op16_t* w16t = reinterpret_cast<const op16_t*>(w16) + total;       // transposed Conv1D weights: [N][K], K contiguous
    op16_t* img = reinterpret_cast<op16_t*>(wimg);
    for (int l = 0; l < c->NL; l++) {
        const int64_t base = layer0 + (int64_t)l * (12 * D * D + 13 * D);
        const int64_t aw = base + 2 * D, pw = aw + 3 * D * D + 3 * D, fw = pw + D * D + 3 * D, p2w = fw + 4 * D * D + 4 * D;
        CC_TRY(skinny_image(w16t + aw, img + aw, 3 * c->D, c->D, S_(stream)));
        CC_TRY(skinny_image(w16t + pw, img + pw, c->D, c->D, S_(stream)));
        CC_TRY(skinny_image(w16t + fw, img + fw, 4 * c->D, c->D, S_(stream)));
        CC_TRY(skinny_image(w16t + p2w, img + p2w, c->D, 4 * c->D, S_(stream)));
    }
    return CC_OK;
}

int64_t CC_API(cc_decode_xt_image_bytes)(const cc_gpt2_cfg* c) {
    if (!cfg_ok(c) || c->H * 64 != c->D) return 0;
    return xt_image_bytes(c->D, c->NL);
}

int CC_API(cc_decode_xt_image)(const cc_gpt2_cfg* c, const uint16_t* w16, uint16_t* wimg, void* stream) {
    if (!cfg_ok(c) || !w16 || !wimg) return CC_ERR_ARG;
    if (c->H * 64 != c->D || !xt_image_bytes(c->D, c->NL)) return CC_ERR_SHAPE;
    const int64_t D = c->D;
    const int64_t layer0 = (int64_t)c->Vp * D + (int64_t)c->NPOS * D;
    const int64_t total = layer0 + (int64_t)c->NL * (12 * D * D + 13 * D) + 2 * D;
    return xt_build_image(c->D, c->NL, layer0, total, w16, wimg, S_(stream));
}

int CC_API(cc_decode_fwd_x)(const cc_gpt2_cfg* c, int32_t R, int32_t Tn, int32_t pos0, int32_t ctx_max, const float* w32, const uint16_t* w16, const uint16_t* wimg,
                    const uint16_t* wteam, const float* x, uint16_t* kv, const int32_t* row_map, int32_t group, void* ws, float* logits, int64_t ldl, float* lpart,
                    void* stream) {
    return decode_fwd_impl(c, R, Tn, pos0, ctx_max, w32, w16, wimg, wteam, x, kv, row_map, group, ws, logits, ldl, lpart, stream);
}
#endif  // CC_EXPERIMENTS

static int decode_fwd_impl(const cc_gpt2_cfg* c, int32_t R, int32_t Tn, int32_t pos0, int32_t ctx_max, const float* w32, const uint16_t* w16, const uint16_t* wimg,
                           const uint16_t* wteam, const float* x, uint16_t* kv, const int32_t* row_map, int32_t group, void* ws, float* logits, int64_t ldl, float* lpart,
                           void* stream) {
    if (group < 1 || (R > 0 && R % group)) return CC_ERR_ARG;
    if (!cfg_ok(c) || R <= 0 || Tn <= 0 || pos0 < 0 || !w32 || !w16 || !x || !kv || !ws || !logits) return CC_ERR_ARG;
    const int Ns = std::min(c->Vp, (c->V + 7) / 8 * 8);
    if (pos0 + Tn > ctx_max || pos0 + Tn > c->NPOS || ldl < Ns || (ldl & 3) || ldl > 0x7fffffff) return CC_ERR_SHAPE;
    if ((size_t)4 * (2 * ctx_max + 8 * (c->D / c->H)) * sizeof(float) > 64 * 1024) return CC_ERR_SHAPE;
    hipStream_t st = S_(stream);
    DecWS w;
    dec_carve(c, R, Tn, ws, w);
#if CC_OP == 2
    x3_set_scratch(w.x3, w.x3_bytes);
#endif
    constexpr int PL = kX3 ? 3 : 1;          // operand-arena addressing as in api.hip (W16): bf16x3 weights own 3x the elements at 3x the offset
    const int D = c->D, M = R * Tn, H = c->H, hd = D / H;
    // arena offsets (same order as api.hip::gpt2_offsets)
    int64_t p = 0;
    const int64_t wte = p; p += (int64_t)c->Vp * D;
    const int64_t wpe = p; p += (int64_t)c->NPOS * D;
    // single-position step: positional add + layer 0's ln_1 in one launch; the last layer's finishing pass applies ln_f (below)
    const bool one = Tn == 1 && D <= 2048 && (D & 3) == 0;
    if (one) {
        hipLaunchKernelGGL(k_add_wpe_ln, dim3((R + 3) / 4), dim3(256), 0, st, x, w32 + wpe, w.x, w32 + p, w32 + p + D, w.xn, R, D, pos0);   // p = layer 0's ln_1.weight
    } else {
        const size_t total = (size_t)M * (D >> 2);
        hipLaunchKernelGGL(k_add_wpe, dim3((int)std::min<size_t>((total + 255) / 256, 2048)), dim3(256), 0, st, x, w32 + wpe, w.x, R, Tn, D, pos0);
    }
    const size_t cache_layer = (size_t)2 * R * ctx_max * D;
    const int64_t total = (int64_t)c->Vp * D + (int64_t)c->NPOS * D + (int64_t)c->NL * (12 * (int64_t)D * D + 13 * (int64_t)D) + 2 * D;
    const uint16_t* w16t = w16 + (size_t)PL * total;   // transposed Conv1D weights (cc_gpt2_sync_weights): forward GEMMs are NT
    const float scale = 1.0f / sqrtf((float)hd);
    // beam-group attention (k_decode_attn_group): single-position steps of `group` consecutive rows that share ancestry (a perf hint only)
    const int grp_cap = (group * (pos0 + 1) + 127) & ~127;          // entries per group, whole passes of 128
    const size_t grp_shm = ((size_t)grp_cap * 8 + 96 + (size_t)8 * group * 64) * sizeof(float);
    const bool grp_attn = (cc_shared::g_decode_mode & 1) && Tn == 1 && group >= 2 && group <= 8 && hd == 64 && grp_shm <= 64 * 1024 &&
                          (size_t)(R / group) * grp_cap <= w.grp_ents;
    bool xn_ready = one;
    bool hf_ready = false;
    int l_first = 0;
    if (one && group >= 2) cc_shared::g_decode_last_path = 0;
    if (grp_attn && one && !kX3 && wteam && (cc_shared::g_decode_mode & 4)) {
        // XCD-team engine (decode_xt.hip): every XCD runs the whole stack for its own captions; CC_ERR_SHAPE = not covered -> the paths below
        XtLaunch L{};
        L.w32 = w32; L.wimg = reinterpret_cast<const op16_t*>(wteam); L.D = D; L.H = H; L.NL = c->NL; L.M = M; L.group = group; L.pos0 = pos0; L.ctx_max = ctx_max;
        L.layer0 = p; L.x = w.x; L.x1 = w.x1; L.qkv = w.qkv; L.att = w.att; L.hact = w.hact; L.hf = w.hf;
        L.kv = reinterpret_cast<act_t*>(kv); L.cache_layer = cache_layer; L.ent = w.grp_ent; L.cnt = w.grp_cnt; L.cap = grp_cap;
        L.ctl = w.xt_ctl; L.sticky = w.xt_ctl + XT_CTL_WORDS;
        static const bool xt_prof = cc_lab_env("CC_XT_PROF") != nullptr;
        L.prof = xt_prof ? w.xt_prof : nullptr;
        // probe the geometry first (no launch): the union kernel below also clears the control words
        XtLaunch probe = L;
        probe.ctl = nullptr;
        if (xt_covers(probe)) {
            switch (group) {
#define CC_GU(G_) case G_: hipLaunchKernelGGL((k_group_union<G_>), dim3(R / group), dim3(64), 0, st, row_map, w.grp_ent, w.grp_cnt, pos0, ctx_max, grp_cap, 1, w.xt_ctl, XT_CTL_WORDS); break;
                CC_GU(2) CC_GU(3) CC_GU(4) CC_GU(5) CC_GU(6) CC_GU(7) CC_GU(8)
#undef CC_GU
                default: break;
            }
            const int rc = decode_layers_xt(L, st);
            if (rc != CC_OK) return rc;
            cc_shared::g_decode_last_path = 2;
            l_first = c->NL;
            p += (int64_t)c->NL * (12 * (int64_t)D * D + 13 * (int64_t)D);
            hf_ready = true;
        }
    }
    if (l_first == 0 && grp_attn && one && !kX3 && (cc_shared::g_decode_mode & 2)) {
        // the whole layer stack as ONE persistent launch (decode_pk.hip); CC_ERR_SHAPE = geometry not covered -> the per-op launches below
        switch (group) {
#define CC_GU(G_) case G_: hipLaunchKernelGGL((k_group_union<G_>), dim3(R / group), dim3(64), 0, st, row_map, w.grp_ent, w.grp_cnt, pos0, ctx_max, grp_cap, 1, w.pk_ctr, PK_CTR_WORDS); break;
            CC_GU(2) CC_GU(3) CC_GU(4) CC_GU(5) CC_GU(6) CC_GU(7) CC_GU(8)
#undef CC_GU
            default: break;
        }
        PkLaunch L{};
        L.w32 = w32; L.w16t = reinterpret_cast<const op16_t*>(w16t); L.D = D; L.H = H; L.NL = c->NL; L.M = M; L.group = group; L.pos0 = pos0; L.ctx_max = ctx_max;
        L.layer0 = p; L.x = w.x; L.x1 = w.x1; L.xn = w.xn; L.qkv = w.qkv; L.att = w.att; L.hact = w.hact; L.hf = w.hf;
        L.slab = w.scratch; L.slab_bytes = w.scratch_bytes; L.kv = reinterpret_cast<act_t*>(kv); L.cache_layer = cache_layer;
        L.ent = w.grp_ent; L.cnt = w.grp_cnt; L.cap = grp_cap; L.ctr = w.pk_ctr;
        static const bool pk_prof = cc_lab_env("CC_PK_PROF") != nullptr;
        L.prof = pk_prof ? w.pk_prof : nullptr;
        const int rc = decode_layers_persistent(L, st);
        if (rc == CC_OK) {
            cc_shared::g_decode_last_path = 1;
            l_first = c->NL;
            p += (int64_t)c->NL * (12 * (int64_t)D * D + 13 * (int64_t)D);
            hf_ready = true;
        } else if (rc != CC_ERR_SHAPE) {
            return rc;
        }
    }
    // fragment-ordered weight image (cc_decode_image): the K-over-the-waves GEMMs then load the weight operand global -> VGPR
    const op16_t* bimg = (!kX3 && wimg && (cc_shared::g_decode_mode & 8)) ? reinterpret_cast<const op16_t*>(wimg) : nullptr;
    for (int l = l_first; l < c->NL; l++) {
        const int64_t l1w = p; p += D;
        const int64_t l1b = p; p += D;
        const int64_t aw = p; p += (int64_t)D * 3 * D;
        const int64_t ab = p; p += 3 * D;
        const int64_t pw = p; p += (int64_t)D * D;
        const int64_t pb = p; p += D;
        const int64_t l2w = p; p += D;
        const int64_t l2b = p; p += D;
        const int64_t fw = p; p += (int64_t)D * 4 * D;
        const int64_t fb = p; p += 4 * D;
        const int64_t p2w = p; p += (int64_t)4 * D * D;
        const int64_t p2b = p; p += D;
        act_t* kc = reinterpret_cast<act_t*>(kv) + (size_t)l * cache_layer;     // (bf16x3: the cache holds fp32, twice the bytes)
        act_t* vc = kc + (size_t)R * ctx_max * D;
        // xn = ln_1(x): produced by the previous layer's fused finish when possible
        if (!xn_ready) CC_TRY(ln_fwd(w.x, D, nullptr, w32 + l1w, w32 + l1b, w.xn, nullptr, nullptr, nullptr, M, D, st));
        // c_attn (+ fused KV append into the cache)
        const bool f_qkv = gemm_nt_skinny_can_fuse(M, 3 * D, D, w.scratch_bytes) &&
                           ((M + 127) / 128) * ((3 * D + 127) / 128) < skinny_single_min_tiles();
        SkinnyFuse fq;
        if (f_qkv) { fq.kcache = kc; fq.vcache = vc; fq.Tn = Tn; fq.pos0 = pos0; fq.ctx_max = ctx_max; }
        fq.bimg = bimg ? bimg + aw : nullptr;
        CC_TRY(gemm_nt_skinny(w.xn, D, w16t + (size_t)PL * aw, D, M, 3 * D, D, w32 + ab, 0, nullptr, nullptr, w.qkv, 3 * D, w.scratch, w.scratch_bytes, st, &fq));
        if (grp_attn) {
            const int ng = R / group, app = f_qkv ? 0 : 1;
#define CC_GRP(G_)                                                                                                                            \
    case G_:                                                                                                                                  \
        if (l == 0) hipLaunchKernelGGL((k_group_union<G_>), dim3(ng), dim3(64), 0, st, row_map, w.grp_ent, w.grp_cnt, pos0, ctx_max, grp_cap, app, \
                                       (unsigned*)nullptr, 0);                                                                               \
        hipLaunchKernelGGL((k_decode_attn_group<G_>), dim3(ng * H), dim3(256), grp_shm, st, w.qkv, kc, vc, w.grp_ent, w.grp_cnt, w.att, H, pos0,  \
                           ctx_max, scale, grp_cap, app);                                                                                     \
        break;
            switch (group) { CC_GRP(2) CC_GRP(3) CC_GRP(4) CC_GRP(5) CC_GRP(6) CC_GRP(7) CC_GRP(8) default: return CC_ERR_ARG; }
#undef CC_GRP
        } else {
            const int nw = R * H * Tn;
            const size_t shm = (size_t)4 * (2 * ctx_max + 8 * hd) * sizeof(float);
            if (f_qkv)
                hipLaunchKernelGGL(k_decode_attn<false>, dim3((nw + 3) / 4), dim3(256), shm, st, w.qkv, kc, vc, row_map, w.att, R, Tn, H, hd, pos0, ctx_max, scale);
            else
                hipLaunchKernelGGL(k_decode_attn<true>, dim3((nw + 3) / 4), dim3(256), shm, st, w.qkv, kc, vc, row_map, w.att, R, Tn, H, hd, pos0, ctx_max, scale);
        }
        // attn.c_proj + residual (+ fused ln_2)
        const bool f_d = gemm_nt_skinny_can_fuse(M, D, D, w.scratch_bytes) && gemm_nt_skinny_can_fuse(M, D, 4 * D, w.scratch_bytes);
        SkinnyFuse f2;
        if (f_d) { f2.ln_gamma = w32 + l2w; f2.ln_beta = w32 + l2b; f2.ln_out16 = w.xn; }
        f2.bimg = bimg ? bimg + pw : nullptr;
        CC_TRY(gemm_nt_skinny(w.att, D, w16t + (size_t)PL * pw, D, M, D, D, w32 + pb, 0, w.x, w.x1, nullptr, D, w.scratch, w.scratch_bytes, st, &f2));
        if (!f_d) CC_TRY(ln_fwd(w.x1, D, nullptr, w32 + l2w, w32 + l2b, w.xn, nullptr, nullptr, nullptr, M, D, st));
        SkinnyFuse f3;
        f3.bimg = bimg ? bimg + fw : nullptr;
        CC_TRY(gemm_nt_skinny(w.xn, D, w16t + (size_t)PL * fw, D, M, 4 * D, D, w32 + fb, 2, nullptr, nullptr, w.hact, 4 * D, w.scratch, w.scratch_bytes, st, &f3));
        // mlp.c_proj + residual (+ fused ln_1 of the next layer: its parameters sit right behind this layer's in the arena)
        // (after the LAST layer p points at ln_f: with one new position per row the finishing pass normalises straight into hf)
        const bool last = l + 1 == c->NL;
        const bool f_next = f_d && (!last || one);
        SkinnyFuse f1;
        if (f_next) { f1.ln_gamma = w32 + p; f1.ln_beta = w32 + p + D; f1.ln_out16 = last ? w.hf : w.xn; }      // p now points at layer l+1's ln_1.weight (or ln_f)
        f1.bimg = bimg ? bimg + p2w : nullptr;
        CC_TRY(gemm_nt_skinny(w.hact, 4 * D, w16t + (size_t)PL * p2w, 4 * D, M, D, 4 * D, w32 + p2b, 0, w.x1, w.x, nullptr, D, w.scratch, w.scratch_bytes, st, &f1));
        xn_ready = f_next && !last;
        hf_ready = f_next && last;
    }
    const int64_t lnf_w = p, lnf_b = p + D;
    if (!hf_ready) {
        hipLaunchKernelGGL(k_last_rows, dim3((R + 255) / 256), dim3(256), 0, st, w.last, R, Tn);
        CC_TRY(ln_fwd(w.x, D, w.last, w32 + lnf_w, w32 + lnf_b, w.hf, nullptr, w.meanf, w.rstdf, R, D, st));
    }
    if (lpart) {      // logits + per-(row, 64-column block) softmax partials for cc_beam_step_p
        const int npart = (Ns + 63) / 64;
        CC_TRY(gemm_logits_part(w.hf, D, w16 + (size_t)PL * wte, D, R, Ns, c->V, D, logits, (int)ldl, lpart, lpart + (size_t)R * npart, npart, st));
    } else {
        CC_TRY(gemm_f32out(0, 0, w.hf, D, w16 + (size_t)PL * wte, D, R, Ns, D, logits, (int)ldl, nullptr, 0, 1.0f, 1, st));
    }
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

#ifdef CC_EXPERIMENTS
int CC_API(cc_decode_ws_check)(const cc_gpt2_cfg* c, int32_t R, int32_t Tn, const void* ws, void* stream) {
    if (!cfg_ok(c) || R <= 0 || Tn <= 0 || !ws) return CC_ERR_ARG;
    DecWS w;
    dec_carve(c, R, Tn, const_cast<void*>(ws), w);
    unsigned e = 0;
    if (hipStreamSynchronize(S_(stream)) != hipSuccess) return CC_ERR_LAUNCH;
    if (hipMemcpy(&e, w.pk_ctr + 7 * PK_MAX_RT, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return CC_ERR_LAUNCH;
    unsigned e2 = 0;
    if (hipMemcpy(&e2, w.xt_ctl + XT_CTL_WORDS, sizeof(e2), hipMemcpyDeviceToHost) != hipSuccess) return CC_ERR_LAUNCH;
    return (e | e2) ? CC_ERR_STATE : CC_OK;
}
#endif

int CC_API(cc_decode_reorder)(const cc_gpt2_cfg* c, int32_t R_src, int32_t R_dst, int32_t ctx, int32_t ctx_max, const uint16_t* kv_src, uint16_t* kv_dst,
                      const int32_t* src, void* stream) {
    if (!cfg_ok(c) || R_src <= 0 || R_dst <= 0 || ctx < 0 || ctx > ctx_max || !kv_src || !kv_dst || !src || kv_src == kv_dst) return CC_ERR_ARG;
    if (ctx == 0) return CC_OK;
    const size_t total = (size_t)c->NL * 2 * R_dst * ctx * (c->D * sizeof(act_t) / 16);
    hipLaunchKernelGGL(k_kv_reorder, dim3((int)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, S_(stream),
                       reinterpret_cast<const act_t*>(kv_src), reinterpret_cast<act_t*>(kv_dst), src, R_src,
                       R_dst, ctx, ctx_max, c->D, c->NL * 2);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

int64_t CC_API(cc_beam_ws_bytes)(int32_t S, int32_t beam, int32_t V) {
    (void)V;
    if (S <= 0 || beam <= 0 || beam > BEAM_MAX) return CC_ERR_SHAPE;
    return (int64_t)S * beam * 2 * sizeof(float) + (int64_t)S * BEAM_CHUNKS * beam * (sizeof(float) + sizeof(int)) + 512;
}

int CC_API(cc_beam_step_p)(int32_t S, int32_t beam, int32_t V, const float* logits, int64_t ldl, const float* lpart, int32_t npart, float temperature,
                   int32_t first, int32_t stop_token, float* scores, float* seq_lengths, uint8_t* has_stopped, int32_t* next_tokens,
                   int32_t* src_rows, void* ws, void* stream);

int CC_API(cc_beam_step)(int32_t S, int32_t beam, int32_t V, const float* logits, int64_t ldl, float temperature, int32_t first, int32_t stop_token,
                 float* scores, float* seq_lengths, uint8_t* has_stopped, int32_t* next_tokens, int32_t* src_rows, void* ws, void* stream) {
    return CC_API(cc_beam_step_p)(S, beam, V, logits, ldl, nullptr, 0, temperature, first, stop_token, scores, seq_lengths, has_stopped, next_tokens,
                                  src_rows, ws, stream);
}

int CC_API(cc_beam_step_p)(int32_t S, int32_t beam, int32_t V, const float* logits, int64_t ldl, const float* lpart, int32_t npart, float temperature,
                   int32_t first, int32_t stop_token, float* scores, float* seq_lengths, uint8_t* has_stopped, int32_t* next_tokens,
                   int32_t* src_rows, void* ws, void* stream) {
    if (S <= 0 || beam <= 0 || beam > BEAM_MAX || V <= 0 || !logits || ldl < V || !scores || !seq_lengths || !has_stopped || !next_tokens ||
        !src_rows || !ws || (lpart && npart * 64 < V))
        return CC_ERR_ARG;
    hipStream_t st = S_(stream);
    // partials from the lm_head epilogue (cc_decode_fwd_p), temperature 1: the whole update in one launch (k_beam_fused)
    constexpr int FUSED_VMAX = 51200;      // the fused kernel's per-thread register image of the partials is sized for vocabularies up to this
    if (lpart && (temperature <= 0.f || temperature == 1.0f) && (beam <= 5 || beam == 8) && V <= FUSED_VMAX) {
        const float* pmax = lpart;
        const float* psum = lpart + (size_t)S * beam * npart;
#define BEAM_FUSED(TB) hipLaunchKernelGGL((k_beam_fused<TB, (TB * (FUSED_VMAX / 64) + 255) / 256>), dim3(S), dim3(256), 0, st, logits, (size_t)ldl, V, npart, pmax, psum, first, stop_token, scores, seq_lengths, has_stopped, next_tokens, src_rows)
        switch (beam) {
            case 1: BEAM_FUSED(1); break;
            case 2: BEAM_FUSED(2); break;
            case 3: BEAM_FUSED(3); break;
            case 4: BEAM_FUSED(4); break;
            case 5: BEAM_FUSED(5); break;
            default: BEAM_FUSED(8); break;
        }
#undef BEAM_FUSED
        return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
    }
    const float inv_temp = 1.0f / (temperature > 0.f ? temperature : 1.0f);   // base.py:83
    float* rs = static_cast<float*>(ws);
    float* pval = rs + (size_t)S * beam * 2;
    int* pidx = reinterpret_cast<int*>(pval + (size_t)S * BEAM_CHUNKS * beam);
    // step 0 reads only row 0 of every sample's block of `beam` rows, but computing all rows' statistics is harmless and uniform
    if ((ldl & 3) || ((uintptr_t)logits & 15))
        hipLaunchKernelGGL(k_beam_rowstats<false>, dim3(S * beam), dim3(512), 0, st, logits, (size_t)ldl, V, inv_temp, rs);
    else
        hipLaunchKernelGGL(k_beam_rowstats<true>, dim3(S * beam), dim3(512), 0, st, logits, (size_t)ldl, V, inv_temp, rs);
#define BEAM_PARTIAL(TB) hipLaunchKernelGGL(k_beam_partial<TB>, dim3(S, BEAM_CHUNKS), dim3(256), 0, st, logits, (size_t)ldl, beam, V, inv_temp, first, rs, scores, seq_lengths, has_stopped, pval, pidx)
    switch (beam) {
        case 1: BEAM_PARTIAL(1); break;
        case 2: BEAM_PARTIAL(2); break;
        case 3: BEAM_PARTIAL(3); break;
        case 4: BEAM_PARTIAL(4); break;
        case 5: BEAM_PARTIAL(5); break;
        case 8: BEAM_PARTIAL(8); break;
        default: BEAM_PARTIAL(0); break;
    }
#undef BEAM_PARTIAL
    hipLaunchKernelGGL(k_beam_final, dim3(S), dim3(64), 0, st, beam, V, first, stop_token, pval, pidx, scores, seq_lengths, has_stopped, next_tokens,
                       src_rows);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

int CC_API(cc_embed_tokens)(const cc_gpt2_cfg* c, int32_t R, const float* w32, const int32_t* tokens, float* out, void* stream) {
    if (!cfg_ok(c) || R <= 0 || !w32 || !tokens || !out) return CC_ERR_ARG;
    const size_t total = (size_t)R * (c->D >> 2);
    hipLaunchKernelGGL(k_embed_tokens, dim3((int)std::min<size_t>((total + 255) / 256, 2048)), dim3(256), 0, S_(stream), w32, tokens, out, R, c->D, c->Vp);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

int CC_API(cc_embed_tokens_bwd)(const cc_gpt2_cfg* c, int32_t R, const float* dout, const int32_t* tokens, float* dwte, void* stream) {
    if (!cfg_ok(c) || R <= 0 || !dout || !tokens || !dwte) return CC_ERR_ARG;
    const size_t total = (size_t)R * c->D;
    hipLaunchKernelGGL(k_embed_tokens_bwd, dim3((int)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, S_(stream), dout, tokens, dwte, R, c->D, c->Vp);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

int CC_API(cc_beam_advance)(const cc_gpt2_cfg* c, int32_t R, int32_t beam, const float* w32, const int32_t* next_tokens, const int32_t* src_rows,
                    int32_t pos, int32_t ctx_max, const int32_t* row_map_in, int32_t* row_map_out, int32_t step, int32_t tok_ld,
                    const int32_t* tokens_in, int32_t* tokens_out, float* x_out, void* stream) {
    if (!cfg_ok(c) || R <= 0 || beam <= 0 || (R % beam) || !w32 || !next_tokens || !x_out || pos < 0 || pos > ctx_max) return CC_ERR_ARG;
    if ((row_map_out && (!row_map_in || row_map_in == row_map_out)) || (tokens_out && (step < 0 || step >= tok_ld || (step > 0 && !tokens_in) ||
                                                                                      tokens_in == tokens_out)))
        return CC_ERR_ARG;
    hipLaunchKernelGGL(k_beam_advance, dim3(R), dim3(256), 0, S_(stream), beam, c->D, w32, next_tokens, src_rows, pos, ctx_max, row_map_in, row_map_out,
                       step, tok_ld, tokens_in, tokens_out, x_out);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

}  // extern "C"
