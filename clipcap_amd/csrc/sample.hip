// On-device sampling step for the KV-cached sampling decoders (reference: clipcap/inference/base.py:135-201 generate_nucleus_sampling,
// :204-279 generate_no_beam with utils.py:5-37 top_k_top_p_filtering / repetition_penalty_apply).  One workgroup per row:
// temperature, repetition penalty, top-k, top-p and the multinomial draw without sorting the V logits — the kept set of a
// descending-sorted prefix is found by radix-selecting its threshold KEY (3 passes of 11/11/10 bits over an order-preserving
// integer image of the logit), with the probability mass of every bucket accumulated in 2^-32 fixed point (deterministic: no
// float atomics).  Ties at a threshold are kept in index order.  The draw is the inverse CDF in index order at a caller-supplied
// uniform, so the same (logits, u) always gives the same token; the reference's torch.multinomial stream is not reproducible
// from outside torch, and parity is on the pre-sampling distribution (probs_out), as SURVEY.md 8(a13) says.
#include "kernels.h"

namespace cc {

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
    bool use_rep;
    __device__ __forceinline__ float val(int i) const {
        float v = x[i];
        if (use_rep && ((bitmap[i >> 5] >> (i & 31)) & 1u)) v = v < 0.f ? v * rep_pen : v / rep_pen;   // utils.py:33-37, before temperature
        return v * inv_temp;
    }
};

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
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; pass++) {
        const int sh = shifts[pass], nb = 1 << bits[pass];
        for (int b = threadIdx.x; b < SM_NB; b += SM_T) { hist_w[b] = 0; hist_n[b] = 0; }
        __syncthreads();
        for (int i = threadIdx.x; i < V; i += SM_T) {
            const float v = r.val(i);
            const unsigned k = sm_key(v);
            if ((k & mask) == prefix && cand(i, k)) {
                const int b = (k >> sh) & (nb - 1);
                atomicAdd(&hist_w[b], weight(v));
                atomicAdd(&hist_n[b], 1u);
            }
        }
        __syncthreads();
        // suffix scan from the top bucket: thread t owns buckets nb-1-2t and nb-2-2t
        const int b0 = nb - 1 - 2 * (int)threadIdx.x, b1 = b0 - 1;
        const unsigned long long w0 = b0 >= 0 ? hist_w[b0] : 0, w1 = b1 >= 0 ? hist_w[b1] : 0;
        unsigned long long tot;
        const unsigned long long before = acc + sm_block_excl(w0 + w1, red, &tot);      // weight of keys above this thread's pair
        if (threadIdx.x == 0) bcast[0] = 0xffffffffu;
        __syncthreads();
        auto reached = [&](unsigned long long c) { return strict ? c > target : c >= target; };
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
                                                      const float* __restrict__ u, int* __restrict__ next_token, float* __restrict__ probs_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    unsigned long long* hist_w = reinterpret_cast<unsigned long long*>(sm_raw);                 // [SM_NB]
    unsigned long long* red = hist_w + SM_NB;                                                   // [SM_T/64 + 2]
    unsigned* hist_n = reinterpret_cast<unsigned*>(red + SM_T / 64 + 2);                        // [SM_NB]
    unsigned* bcast = hist_n + SM_NB;                                                           // [4]
    float* fred = reinterpret_cast<float*>(bcast + 4);                                          // [SM_T/64]
    unsigned* bitmap = reinterpret_cast<unsigned*>(fred + SM_T / 64);                           // [(V+31)/32]
    const int row = blockIdx.x;
    SmRow r;
    r.x = logits + (size_t)row * ld;
    r.bitmap = bitmap;
    r.rep_pen = rep_pen;
    r.inv_temp = inv_temp;
    r.use_rep = hist != nullptr && hist_len > 0 && rep_pen != 1.0f;
    if (r.use_rep) {
        for (int i = threadIdx.x; i < (V + 31) / 32; i += SM_T) bitmap[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < hist_len; i += SM_T) {
            const long long t = hist[(size_t)row * hist_ld + i];
            if (t >= 0 && t < V) atomicOr(&bitmap[t >> 5], 1u << (t & 31));
        }
        __syncthreads();
    }
    // row maximum
    float m = -INFINITY;
    for (int i = threadIdx.x; i < V; i += SM_T) m = fmaxf(m, r.val(i));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) fred[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fred[0];
    for (int i = 1; i < SM_T / 64; i++) m = fmaxf(m, fred[i]);
    auto wfix = [m](float v) { return (unsigned long long)(expf(v - m) * 4294967296.0f); };      // 2^-32 fixed point, <= 2^32
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
    // index-order rank of the ties needs contiguous chunks: thread t owns [t*C, (t+1)*C)
    const int C = (V + SM_T - 1) / SM_T, i0 = threadIdx.x * C, i1 = min(V, i0 + C);
    unsigned long long tot;
    unsigned long long k_rank0 = 0;
    if (use_k) {
        unsigned long long c = 0;
        for (int i = i0; i < i1; i++) c += sm_key(r.val(i)) == Tk;
        k_rank0 = sm_block_excl(c, red, &tot);
    }
    // candidate test of the top-p selection: member of the top-k set.  A threshold tie is a member iff its index-order rank is
    // below k_keep; the strided selection passes cannot recompute ranks, so the index of the LAST kept tie is published instead:
    // ties with index <= k_last are members.
    if (threadIdx.x == 0) bcast[1] = 0xffffffffu;
    __syncthreads();
    if (use_k) {
        unsigned long long rk = k_rank0;
        for (int i = i0; i < i1; i++)
            if (sm_key(r.val(i)) == Tk) {
                if (rk + 1 == k_keep) bcast[1] = (unsigned)i;
                rk++;
            }
        __syncthreads();
    }
    const unsigned k_last = bcast[1];       // with k_keep == 0 no tie is kept (k_last stays 0xffffffff and is never consulted)
    auto in_topk = [=](int i, unsigned k) { return !use_k || k > Tk || (k == Tk && k_keep > 0 && (unsigned)i <= k_last); };
    // mass of the reference set: full softmax (mode 0) or the top-k set (mode 1)
    unsigned long long zl = 0, zk = 0;
    for (int i = threadIdx.x; i < V; i += SM_T) {
        const float v = r.val(i);
        const unsigned long long w = wfix(v);
        zl += w;
        if (in_topk(i, sm_key(v))) zk += w;
    }
    const unsigned long long Zall = sm_block_sum(zl, red), Zk = sm_block_sum(zk, red);
    // top-p threshold among the top-k set
    unsigned Tp = 0;
    unsigned long long p_gt = 0, p_tw = 0;
    unsigned p_tn = 0;
    const bool use_p = top_p > 0.f;
    unsigned p_keep = 0;
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
    // final: kept = top-k member and (key > Tp or one of the first p_keep ties in index order)
    unsigned long long pc = 0;
    if (use_p && p_tn > 0)
        for (int i = i0; i < i1; i++) {
            const unsigned k = sm_key(r.val(i));
            pc += (k == Tp && in_topk(i, k));
        }
    const unsigned long long p_rank0 = sm_block_excl(pc, red, &tot);
    unsigned long long mine = 0;
    {
        unsigned long long rk = p_rank0;
        for (int i = i0; i < i1; i++) {
            const float v = r.val(i);
            const unsigned k = sm_key(v);
            bool keep = in_topk(i, k);
            if (keep && use_p && Tp != 0) {
                if (k < Tp) keep = false;
                else if (k == Tp) { keep = rk < p_keep; rk++; }
            }
            if (keep) mine += wfix(v);
        }
    }
    unsigned long long total;
    const unsigned long long base = sm_block_excl(mine, red, &total);
    // inverse CDF in index order at u
    const double uu = (double)fminf(fmaxf(u[row], 0.f), 0.99999994f);
    const unsigned long long pick = (unsigned long long)(uu * (double)total);       // in [0, total)
    if (threadIdx.x == 0) bcast[2] = 0xffffffffu;
    __syncthreads();
    {
        unsigned long long rk = p_rank0, c = base;
        const float inv_total = total > 0 ? 1.0f / (float)total : 0.f;
        for (int i = i0; i < i1; i++) {
            const float v = r.val(i);
            const unsigned k = sm_key(v);
            bool keep = in_topk(i, k);
            if (keep && use_p && Tp != 0) {
                if (k < Tp) keep = false;
                else if (k == Tp) { keep = rk < p_keep; rk++; }
            }
            const unsigned long long w = keep ? wfix(v) : 0;
            if (w > 0 && pick >= c && pick < c + w) bcast[2] = (unsigned)i;
            c += w;
            if (probs_out) probs_out[(size_t)row * V + i] = (float)w * inv_total;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = bcast[2];
        if (t == 0xffffffffu) {                       // total == 0 (all -inf): fall back to the arg-max like torch would fail loudly; pick 0
            t = 0;
        }
        next_token[row] = (int)t;
    }
}

size_t sample_lds_bytes(int V) {
    return (size_t)SM_NB * 8 + (SM_T / 64 + 2) * 8 + (size_t)SM_NB * 4 + 16 + (SM_T / 64) * 4 + (size_t)((V + 31) / 32) * 4;
}

int sample_rows(const float* logits, int R, int V, int ld, float temperature, int top_k, float top_p, int mode, const long long* hist,
                int hist_len, int hist_ld, float rep_pen, const float* u, int* next_token, float* probs_out, hipStream_t st) {
    if (R <= 0) return CC_OK;
    if (V <= 0 || ld < V || (mode != 0 && mode != 1)) return CC_ERR_ARG;
    const size_t sh = sample_lds_bytes(V);
    if (sh > 150 * 1024) return CC_ERR_SHAPE;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_sample_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; }
    const float inv_temp = temperature > 0.f ? 1.0f / temperature : 1.0f;          // base.py:163
    hipLaunchKernelGGL(k_sample_rows, dim3(R), dim3(SM_T), sh, st, logits, ld, V, inv_temp, top_k, top_p, mode, hist, hist_len, hist_ld,
                       rep_pen, u, next_token, probs_out);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

}  // namespace cc
