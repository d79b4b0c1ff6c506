// On-device sampling step for the KV-cached sampling decoders (reference: clipcap/inference/base.py:135-201 generate_nucleus_sampling,
// :204-279 generate_no_beam with utils.py:5-37 top_k_top_p_filtering / repetition_penalty_apply).  One workgroup per row:
// temperature, repetition penalty, top-k, top-p and the multinomial draw without sorting the V logits — the kept set of a
// descending-sorted prefix is found by radix-selecting its threshold KEY (3 passes of 11/11/10 bits over an order-preserving
// integer image of the logit), with the probability mass of every bucket accumulated in 2^-32 fixed point (deterministic: no
// float atomics).  Ties at a threshold are kept in index order.  The draw is the inverse CDF in index order at a caller-supplied
// uniform, so the same (logits, u) always gives the same token; the reference's torch.multinomial stream is not reproducible
// from outside torch, and parity is on the pre-sampling distribution (probs_out), as SURVEY.md 8(a13) says.
#include "kernels.h"

namespace CC_NS {

constexpr int SM_T = 1024;          // threads per row
constexpr int SM_NB = 2048;         // radix buckets (11 bits)

__device__ __forceinline__ unsigned sm_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // larger float <-> larger key
}

struct SmRow {
    const float* x;
    const unsigned* bitmap;   // LDS: history tokens (repetition penalty)
    float rep_pen, inv_temp;
    float xmin, xscale;       // linear pre-bucket of the value range: lb(v) in [0, SM_NB), monotone in v
    bool use_rep;
    __device__ __forceinline__ float xform(int i, float v) const {
        if (use_rep && ((bitmap[i >> 5] >> (i & 31)) & 1u)) v = v < 0.f ? v * rep_pen : v / rep_pen;   // utils.py:33-37, before temperature
        float t = v * inv_temp;
        asm volatile("" : "+v"(t));      // the ROUNDED product is the element's value everywhere: nothing downstream may fuse with this multiply (see lb)
        return t;
    }
    __device__ __forceinline__ float val(int i) const { return xform(i, x[i]); }
    // Every pass of a selection must put an element into the SAME bucket.  With `v = x * inv_temp` visible to the optimiser, it was free
    // to contract x * inv_temp - xmin into one fma at one inlined call site and not at another (-ffp-contract=fast; HIP's __fsub_rn /
    // __fmul_rn are plain operators and do not prevent it): an element within 1e-7 of a bucket boundary then left the crossing bucket
    // between the histogram pass and the radix passes, the crossing was not found, and the row fell back to "keep everything" — once
    // in ~10^4 rows (found by tools/fuzz_decode_steps.py).  xform() now hands out the product behind an optimisation barrier.
    __device__ __forceinline__ int lb(float v) const { return (int)fminf(fmaxf((v - xmin) * xscale, 0.f), (float)(SM_NB - 1)); }
};

// Every pass over the row batches its loads: a one-element-per-iteration loop exposes a full L2 round trip per element
// (measured 870 cycles per iteration: 20 us per pass per row).
constexpr int SM_U = 8;
template <class F>
__device__ __forceinline__ void sm_for_strided(const SmRow& r, int V, F f) {       // any order: block-strided, SM_U loads in flight
    for (int i0 = threadIdx.x; i0 < V; i0 += SM_T * SM_U) {
        float raw[SM_U];
#pragma unroll
        for (int k = 0; k < SM_U; k++) { const int i = i0 + k * SM_T; raw[k] = i < V ? r.x[i] : 0.f; }
#pragma unroll
        for (int k = 0; k < SM_U; k++) { const int i = i0 + k * SM_T; if (i < V) f(i, r.xform(i, raw[k])); }
    }
}
// index order inside a wave's range [lo, hi): 64 elements per step, 4 steps of loads in flight; f(i, v, valid) returns false to stop
template <class F>
__device__ __forceinline__ void sm_for_wave_range(const SmRow& r, int lo, int hi, int lane, F f) {
    for (int i0 = lo; i0 < hi; i0 += 256) {
        float raw[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int i = i0 + 64 * k + lane; raw[k] = i < hi ? r.x[i] : 0.f; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + 64 * k + lane;
            if (i0 + 64 * k >= hi) return;
            if (!f(i, i < hi ? r.xform(i, raw[k]) : 0.f, i < hi)) return;
        }
    }
}

__device__ __forceinline__ unsigned long long sm_block_sum(unsigned long long v, unsigned long long* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    unsigned long long t = 0;
    for (int i = 0; i < SM_T / 64; i++) t += red[i];
    return t;
}
// exclusive prefix sum over the block in thread order; *total gets the block sum
__device__ __forceinline__ unsigned long long sm_block_excl(unsigned long long v, unsigned long long* red, unsigned long long* total) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    unsigned long long inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long n = __shfl_up(inc, o, 64);
        if (l >= o) inc += n;
    }
    __syncthreads();
    if (l == 63) red[w] = inc;
    __syncthreads();
    unsigned long long base = 0, t = 0;
    for (int i = 0; i < SM_T / 64; i++) {
        if (i < w) base += red[i];
        t += red[i];
    }
    *total = t;
    return base + inc - v;
}

// Threshold of the minimal descending-key prefix of the candidate set whose weight reaches `target` (>= target, or > target when
// `strict`).  Candidates: all elements for which cand(i, key) holds.  Returns the threshold key T;
// *gt = weight of candidates with key > T, *ties_w / *ties_n = weight and count of candidates with key == T.  If the whole set
// does not reach the target, T = 0 (everything is kept).
template <class Cand, class Weight>
__device__ unsigned sm_select(const SmRow& r, int V, unsigned long long target, bool strict, Cand cand, Weight weight,
                              unsigned long long* hist_w, unsigned* hist_n, unsigned long long* red, unsigned* bcast,
                              unsigned long long* gt, unsigned long long* ties_w, unsigned* ties_n) {
    unsigned prefix = 0, mask = 0;
    unsigned long long acc = 0;
    bool none = false;
    auto reached = [&](unsigned long long c) { return strict ? c > target : c >= target; };
    // Pass 0 spreads the row over SM_NB LINEAR buckets of its value range (monotone in the key, so bucket order is key order):
    // the top 11 key bits alone are sign + exponent + 2 mantissa bits and put almost every logit of a row into ~20 buckets —
    // the 64-bit LDS atomics of one wave then serialise on a handful of addresses (measured: 500 us per step for 320 rows).
    // The exact radix passes below only see the crossing bucket's few dozen elements.
    int lbsel;
    {
        for (int b = threadIdx.x; b < SM_NB; b += SM_T) { hist_w[b] = 0; hist_n[b] = 0; }
        __syncthreads();
        sm_for_strided(r, V, [&](int i, float v) {
            if (cand(i, sm_key(v))) {
                const int b = r.lb(v);
                atomicAdd(&hist_w[b], weight(v));
                atomicAdd(&hist_n[b], 1u);
            }
        });
        __syncthreads();
        const int b0 = SM_NB - 1 - 2 * (int)threadIdx.x, b1 = b0 - 1;
        const unsigned long long w0 = hist_w[b0], w1 = hist_w[b1];
        unsigned long long tot;
        const unsigned long long before = sm_block_excl(w0 + w1, red, &tot);
        if (threadIdx.x == 0) bcast[0] = 0xffffffffu;
        __syncthreads();
        if (!reached(before) && reached(before + w0 + w1)) {
            const int b = reached(before + w0) ? b0 : b1;
            bcast[0] = (unsigned)b;
            red[SM_T / 64] = before + (b == b0 ? 0 : w0);
        }
        __syncthreads();
        if (bcast[0] == 0xffffffffu) { *gt = 0; *ties_w = 0; *ties_n = 0; __syncthreads(); return 0u; }
        lbsel = (int)bcast[0];
        acc = red[SM_T / 64];
        __syncthreads();
    }
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; pass++) {
        const int sh = shifts[pass], nb = 1 << bits[pass];
        for (int b = threadIdx.x; b < SM_NB; b += SM_T) { hist_w[b] = 0; hist_n[b] = 0; }
        __syncthreads();
        sm_for_strided(r, V, [&](int i, float v) {
            const unsigned k = sm_key(v);
            if (r.lb(v) == lbsel && (k & mask) == prefix && cand(i, k)) {
                const int b = (k >> sh) & (nb - 1);
                atomicAdd(&hist_w[b], weight(v));
                atomicAdd(&hist_n[b], 1u);
            }
        });
        __syncthreads();
        // suffix scan from the top bucket: thread t owns buckets nb-1-2t and nb-2-2t
        const int b0 = nb - 1 - 2 * (int)threadIdx.x, b1 = b0 - 1;
        const unsigned long long w0 = b0 >= 0 ? hist_w[b0] : 0, w1 = b1 >= 0 ? hist_w[b1] : 0;
        unsigned long long tot;
        const unsigned long long before = acc + sm_block_excl(w0 + w1, red, &tot);      // weight of keys above this thread's pair
        if (threadIdx.x == 0) bcast[0] = 0xffffffffu;
        __syncthreads();
        if (!reached(before) && reached(before + w0 + w1)) {
            const int b = reached(before + w0) ? b0 : b1;
            bcast[0] = (unsigned)b;
            red[SM_T / 64] = before + (b == b0 ? 0 : w0);          // weight strictly above bucket b
        }
        __syncthreads();
        if (bcast[0] == 0xffffffffu) { none = true; break; }
        acc = red[SM_T / 64];
        prefix |= bcast[0] << sh;
        mask |= (unsigned)(nb - 1) << sh;
        __syncthreads();
    }
    if (none) { *gt = 0; *ties_w = 0; *ties_n = 0; return 0u; }
    const int bl = prefix & 1023;                                  // last pass: the bucket is one exact key
    *gt = acc;
    *ties_w = hist_w[bl];
    *ties_n = hist_n[bl];
    __syncthreads();
    return prefix;
}

// mode 0: generate_nucleus_sampling — p = softmax(x/T); the top_k largest; minimal sorted prefix with cumulative p >= top_p (mass
//         relative to the FULL softmax), renormalised.    mode 1: top_k_top_p_filtering then softmax — keeps logits >= the k-th
//         largest (all ties), then the minimal sorted prefix with cumulative mass > top_p relative to the top-k set.
__global__ __launch_bounds__(SM_T) void k_sample_rows(const float* __restrict__ logits, int ld, int V, float inv_temp, int top_k, float top_p,
                                                      int mode, const long long* __restrict__ hist, int hist_len, int hist_ld, float rep_pen,
                                                      const float* __restrict__ u, int* __restrict__ next_token, float* __restrict__ probs_out,
                                                      int stop_tok, float len_pen) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    unsigned long long* hist_w = reinterpret_cast<unsigned long long*>(sm_raw);                 // [SM_NB]
    unsigned long long* red = hist_w + SM_NB;                                                   // [SM_T/64 + 2]
    unsigned* hist_n = reinterpret_cast<unsigned*>(red + SM_T / 64 + 2);                        // [SM_NB]
    unsigned* bcast = hist_n + SM_NB;                                                           // [4]
    float* fred = reinterpret_cast<float*>(bcast + 4);                                          // [SM_T/64]
    float* fred2 = fred + SM_T / 64;                                                            // [SM_T/64]
    unsigned* bitmap = reinterpret_cast<unsigned*>(fred2 + SM_T / 64);                          // [(V+31)/32]
    const int row = blockIdx.x;
    SmRow r;
    r.x = logits + (size_t)row * ld;
    r.bitmap = bitmap;
    r.rep_pen = rep_pen;
    r.inv_temp = inv_temp;
    r.use_rep = hist != nullptr && hist_len > 0 && rep_pen != 1.0f;
    // Sentence-length penalty of generate_no_beam (no_beam.py:55-60 -> utils.py:40-51): AFTER filtering, a history token whose value
    // EQUALS float(stop token id) is multiplied by len_pen (the reference compares the gathered logit VALUES with the id, and this does
    // the same).  stop_tok < 0 = off.
    const bool slp = stop_tok >= 0 && hist != nullptr && hist_len > 0;
    const float stopf = (float)stop_tok;
    if (r.use_rep || slp) {
        for (int i = threadIdx.x; i < (V + 31) / 32; i += SM_T) bitmap[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < hist_len; i += SM_T) {
            const long long t = hist[(size_t)row * hist_ld + i];
            if (t >= 0 && t < V) atomicOr(&bitmap[t >> 5], 1u << (t & 31));
        }
        __syncthreads();
    }
    // row maximum and smallest finite value
    float m = -INFINITY, mn = INFINITY;
    r.xmin = 0.f; r.xscale = 0.f;
    sm_for_strided(r, V, [&](int, float v) {
        m = fmaxf(m, v);
        if (v > -INFINITY) mn = fminf(mn, v);
    });
    for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
    if ((threadIdx.x & 63) == 0) { fred[threadIdx.x >> 6] = m; fred2[threadIdx.x >> 6] = mn; }
    __syncthreads();
    m = fred[0]; mn = fred2[0];
    for (int i = 1; i < SM_T / 64; i++) { m = fmaxf(m, fred[i]); mn = fminf(mn, fred2[i]); }
    r.xmin = mn;
    r.xscale = (m > mn && mn < INFINITY) ? ((float)SM_NB - 0.001f) / (m - mn) : 0.f;
    // 2^-32 fixed point in one v_exp_f32 + one v_cvt_u32_f32: exp(v - m) <= 1 scaled to just under 2^32 (4294967040 = 2^32 - 256,
    // the largest float below 2^32)
    auto wfix = [m](float v) { return (unsigned long long)(unsigned)(__expf(v - m) * 4294967040.0f); };
    auto one = [](float) { return 1ull; };
    // top-k threshold (by count)
    unsigned Tk = 0;
    unsigned long long k_gt = 0, k_tw = 0;
    unsigned k_tn = 0;
    const bool use_k = top_k > 0 && top_k < V;
    if (use_k) {
        unsigned long long cgt, ctw;
        Tk = sm_select(r, V, (unsigned long long)top_k, false, [](int, unsigned) { return true; }, one, hist_w, hist_n, red, bcast, &cgt, &ctw, &k_tn);
        k_gt = cgt;
        (void)ctw;
    }
    // number of threshold ties the top-k set keeps, in index order: mode 1 keeps every tie (logits >= kth), mode 0 exactly k
    const unsigned k_keep = use_k ? (mode == 1 ? k_tn : (unsigned)((unsigned long long)top_k - k_gt)) : 0;
    // Index-order work is done in WAVE-contiguous ranges walked 64 elements at a time (coalesced; ranks inside a step from a
    // ballot, prefixes across steps in a wave-uniform counter, across waves through 16 LDS totals).  A per-THREAD contiguous chunk
    // makes every load instruction touch 64 different cache lines: 5 such passes x 320 rows were 10 GB of L2 traffic, 500 us.
    constexpr int NW = SM_T / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int CW = ((V + NW - 1) / NW + 63) / 64 * 64, w_lo = wave * CW, w_hi = min(V, w_lo + CW);
    unsigned long long* wtot = red;                                     // [NW] per-wave totals (red is free between block sums)
    // index of the n-th (1-based) element, in index order, satisfying pred; 0xffffffff if n == 0 or there are fewer
    auto locate_nth = [&](unsigned n, auto pred) -> unsigned {
        unsigned cnt = 0;
        sm_for_wave_range(r, w_lo, w_hi, lane, [&](int i, float v, bool valid) {
            cnt += __popcll(__ballot(valid && pred(i, v)));
            return true;
        });
        __syncthreads();
        if (lane == 0) wtot[wave] = cnt;
        if (threadIdx.x == 0) bcast[3] = 0xffffffffu;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; w++) base += (unsigned)wtot[w];
        if (n > 0 && base < n && n <= base + cnt) {                     // the n-th lives in this wave's range
            unsigned run = base;
            sm_for_wave_range(r, w_lo, w_hi, lane, [&](int i, float v, bool valid) {
                const bool f = valid && pred(i, v);
                const unsigned long long bm = __ballot(f);
                const unsigned before = run + __popcll(bm & ((1ull << lane) - 1ull));
                if (f && before + 1 == n) bcast[3] = (unsigned)i;
                run += __popcll(bm);
                return run < n;
            });
        }
        __syncthreads();
        const unsigned res = bcast[3];
        __syncthreads();
        return res;
    };
    // top-k membership: a threshold tie is a member iff it is among the first k_keep ties in index order, i.e. index <= k_last
    // (k_last stays 0xffffffff = "every tie" when all of them are kept — the usual case of a single element at the threshold)
    unsigned k_last = 0xffffffffu;
    if (use_k && k_keep > 0 && k_keep < k_tn) k_last = locate_nth(k_keep, [&](int, float v) { return sm_key(v) == Tk; });
    auto in_topk = [=](int i, unsigned k) { return !use_k || k > Tk || (k == Tk && k_keep > 0 && (unsigned)i <= k_last); };
    // mass of the reference set: full softmax (mode 0) or the top-k set (mode 1)
    unsigned long long zl = 0, zk = 0;
    sm_for_strided(r, V, [&](int i, float v) {
        const unsigned long long w = wfix(v);
        zl += w;
        if (in_topk(i, sm_key(v))) zk += w;
    });
    const unsigned long long Zall = sm_block_sum(zl, red), Zk = sm_block_sum(zk, red);
    // top-p threshold among the top-k set
    unsigned Tp = 0;
    unsigned long long p_gt = 0, p_tw = 0;
    unsigned p_tn = 0, p_keep = 0;
    const bool use_p = top_p > 0.f;
    if (use_p) {
        const double ref = (double)(mode == 0 ? Zall : Zk);
        unsigned long long target = (unsigned long long)((double)top_p * ref);
        if (mode == 0 && target == 0) target = 1;              // searchsorted(cum, tiny) = 0: the first element is always kept
        Tp = sm_select(r, V, target, mode == 1, in_topk, wfix, hist_w, hist_n, red, bcast, &p_gt, &p_tw, &p_tn);
        if (p_tn > 0) {
            const unsigned long long each = p_tw / p_tn;       // ties have identical weights
            unsigned long long c = p_gt;
            while (p_keep < p_tn) {
                c += each;
                p_keep++;
                if (mode == 1 ? c > target : c >= target) break;
            }
        }
    }
    // nucleus membership of a threshold tie, same device: index <= p_last
    unsigned p_last = 0xffffffffu;
    const bool p_on = use_p && Tp != 0;
    if (p_on && p_keep > 0 && p_keep < p_tn) p_last = locate_nth(p_keep, [&](int i, float v) { const unsigned k = sm_key(v); return k == Tp && in_topk(i, k); });
    auto kept = [=](int i, unsigned k) {
        if (!in_topk(i, k)) return false;
        if (!p_on) return true;
        return k > Tp || (k == Tp && p_keep > 0 && (unsigned)i <= p_last);
    };
    // final values: the kept set with the sentence-length penalty applied; their maximum replaces m in the softmax weights (the
    // penalised value may exceed the row maximum, or the row maximum may be the value that shrinks)
    auto fin = [=](int i, float v) { return (slp && v == stopf && ((bitmap[i >> 5] >> (i & 31)) & 1u)) ? v * len_pen : v; };
    float m2 = m;
    if (slp) {
        float mm = -INFINITY;
        sm_for_strided(r, V, [&](int i, float v) { if (kept(i, sm_key(v))) mm = fmaxf(mm, fin(i, v)); });
        for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) fred[threadIdx.x >> 6] = mm;
        __syncthreads();
        mm = fred[0];
        for (int i = 1; i < SM_T / 64; i++) mm = fmaxf(mm, fred[i]);
        m2 = mm;
    }
    auto wfin = [=](int i, float v) { return (unsigned long long)(unsigned)(__expf(fin(i, v) - m2) * 4294967040.0f); };
    // kept mass per wave range (index order across waves), total, and the draw: inverse CDF in index order at u
    unsigned long long mine = 0;
    sm_for_wave_range(r, w_lo, w_hi, lane, [&](int i, float v, bool valid) {
        if (valid && kept(i, sm_key(v))) mine += wfin(i, v);
        return true;
    });
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    __syncthreads();
    if (lane == 0) wtot[wave] = mine;
    if (threadIdx.x == 0) bcast[2] = 0xffffffffu;
    __syncthreads();
    unsigned long long total = 0, base = 0;
    for (int w = 0; w < NW; w++) {
        if (w < wave) base += wtot[w];
        total += wtot[w];
    }
    const double uu = (double)fminf(fmaxf(u[row], 0.f), 0.99999994f);
    const unsigned long long pick = (unsigned long long)(uu * (double)total);       // in [0, total)
    if (mine > 0 && pick >= base && pick < base + mine) {                // the drawn token lies in this wave's range
        unsigned long long run = base;
        sm_for_wave_range(r, w_lo, w_hi, lane, [&](int i, float v, bool valid) {
            const unsigned long long w = (valid && kept(i, sm_key(v))) ? wfin(i, v) : 0;
            unsigned long long inc = w;                                  // inclusive prefix over the 64 lanes
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long n = __shfl_up(inc, o, 64);
                if (lane >= o) inc += n;
            }
            const unsigned long long lo = run + inc - w;
            if (w > 0 && pick >= lo && pick < lo + w) bcast[2] = (unsigned)i;
            run += __shfl(inc, 63, 64);
            return run <= pick;
        });
    }
    __syncthreads();
    if (probs_out) {
        const float inv_total = total > 0 ? 1.0f / (float)total : 0.f;
        sm_for_strided(r, V, [&](int i, float v) { probs_out[(size_t)row * V + i] = kept(i, sm_key(v)) ? (float)wfin(i, v) * inv_total : 0.f; });
    }
    if (threadIdx.x == 0) {
        unsigned t = bcast[2];
        if (t == 0xffffffffu) t = 0;                  // total == 0 (every logit -inf): nothing to draw from
        next_token[row] = (int)t;
    }
}

size_t sample_lds_bytes(int V) {
    return (size_t)SM_NB * 8 + (SM_T / 64 + 2) * 8 + (size_t)SM_NB * 4 + 16 + 2 * (SM_T / 64) * 4 + (size_t)((V + 31) / 32) * 4;
}

int sample_rows(const float* logits, int R, int V, int ld, float temperature, int top_k, float top_p, int mode, const long long* hist,
                int hist_len, int hist_ld, float rep_pen, const float* u, int* next_token, float* probs_out, hipStream_t st, int stop_tok,
                float len_pen) {
    if (R <= 0) return CC_OK;
    if (V <= 0 || ld < V || (mode != 0 && mode != 1)) return CC_ERR_ARG;
    const size_t sh = sample_lds_bytes(V);
    if (sh > 150 * 1024) return CC_ERR_SHAPE;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_sample_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; }
    const float inv_temp = temperature > 0.f ? 1.0f / temperature : 1.0f;          // base.py:163
    hipLaunchKernelGGL(k_sample_rows, dim3(R), dim3(SM_T), sh, st, logits, ld, V, inv_temp, top_k, top_p, mode, hist, hist_len, hist_ld,
                       rep_pen, u, next_token, probs_out, stop_tok, len_pen);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

}  // namespace CC_NS
