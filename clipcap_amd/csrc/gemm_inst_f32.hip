// Built with `make ABLATION=1` this TU also carries the timing-only ablation variants of the NT kernel (CC_GEMM_ABL=1|2|4|5|6 at
// run time, tools/gemm_bench.py); the default build has none, so product code generation is not perturbed by sibling variants.
#include "gemm.hip.h"
#include <algorithm>
#include "gemm_api.h"
namespace CC_NS {
// both operands already in the 16-bit operand form (in the bf16x3 build: their hi / lo images, K = 3 x the logical depth)
static int gemm_f32out_op(int al, int bl, const op16_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, float* C, int ldc,
                          const float* bias, int mode, float alpha, int ksplit, hipStream_t st) {
    if ((ldc & 3) || (N & 7)) return CC_ERR_SHAPE;
    if (ksplit > 1 && mode != 2) return CC_ERR_ARG;  // slab mode (3) is reached through gemm_wgrad only
    EpiF32 e{C, bias, ldc, M, N, mode, alpha};
    return launch_gemm(al, bl, A, lda, B, ldb, M, N, K, ksplit, e, st);
}
int gemm_f32out(int al, int bl, const act_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, float* C, int ldc,
                const float* bias, int mode, float alpha, int ksplit, hipStream_t st) {
    cc_shared::ProfScope _all(cc_shared::SITE_ALL_GEMMS, st, 2.0 * M * N * (double)K);
    const op16_t* A16;
    CC_X3_NT(A, lda, ldb, M, K, A16, al, bl, st);
    return gemm_f32out_op(al, bl, A16, lda, B, ldb, M, N, K, C, ldc, bias, mode, alpha, ksplit, st);
}
__global__ __launch_bounds__(256) void k_slab_reduce(const float* __restrict__ slabs, size_t slab_elems, int ks, int Nw, float* __restrict__ dW,
                                                     int ldw, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(slabs)[i];
        for (int s = 1; s < ks; s++) {
            const float4 b = reinterpret_cast<const float4*>(slabs + s * slab_elems)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const size_t e = i * 4;
        float* d = dW + (e / Nw) * ldw + (e % Nw);
        const float4 c = *reinterpret_cast<const float4*>(d);
        *reinterpret_cast<float4*>(d) = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    }
}

struct SlabItems { WgradBatch::Item it[8]; };
__global__ __launch_bounds__(256) void k_slab_reduce_multi(SlabItems items) {
    const WgradBatch::Item& q = items.it[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q.n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(q.slabs)[i];
        for (int s = 1; s < q.ks; s++) {
            const float4 b = reinterpret_cast<const float4*>(q.slabs + s * q.slab)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const size_t e = i * 4;
        float* d = q.dW + (e / q.Nw) * q.ldw + (e % q.Nw);
        const float4 c = *reinterpret_cast<const float4*>(d);
        *reinterpret_cast<float4*>(d) = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    }
}
// the deferred problems of a batch as one grouped launch: a common slice count ks is chosen to minimise
// rounds x (K-steps per block + fixed per-block cost), rounds = ceil(blocks / CUs), then the slabs are parked for the batched reduce
static int wgrad_launch_group(WgradBatch& b, hipStream_t st) {
    if (b.nd == 0) return CC_OK;
    TTGroup grp;
    grp.n = b.nd;
    grp.mode = 0;
    const int K = b.d[0].K;
    if (b.direct) {
        // one K slice per tile, dW += tile in the epilogue: the 256 x 256 kernel when K allows it (a block walks the whole K: 160 steps at K = 5120)
        double fl = 0.0;
        for (int i = 0; i < b.nd; i++) fl += 2.0 * b.d[i].Mw * b.d[i].Nw * (double)K;
        cc_shared::ProfScope _alld(cc_shared::SITE_ALL_GEMMS, st, fl);
        const bool t256 = (K % H_BK) == 0;
        const int tile = t256 ? 256 : 128;
        const int kt64 = (K + G_BK - 1) / G_BK;
        int first = 0;
        for (int i = 0; i < b.nd; i++) {
            const auto& d = b.d[i];
            grp.A[i] = d.X; grp.B[i] = d.Y;
            grp.g[i] = GemmShape{d.Mw, d.Nw, d.K, d.ldx, d.ldy, kt64 * G_BK, 8, 0};
            grp.slab[i] = d.dW; grp.zstride[i] = 0; grp.ldc[i] = d.ldw;
            grp.first[i] = first;
            first += ((d.Mw + tile - 1) / tile) * ((d.Nw + tile - 1) / tile);
        }
        for (int i = b.nd; i <= TT_GROUP_MAX; i++) grp.first[i] = first;
        grp.mode = 1;
        if (t256) {
            // the tail beyond the last full round of 256 workgroups: K slices + atomics, so that it is a short round of every CU
            // (CC_WGRAD_TAIL=0 in the lab build: whole tiles only)
            static const bool tail_on = []() { const char* e = cc_lab_env("CC_WGRAD_TAIL"); return !e || atoi(e) != 0; }();
            const int full = (first / 256) * 256, rem = first - full;
            int sp = rem > 0 ? 256 / rem : 1;
            sp = std::min(sp, std::min(8, std::max(1, (K / H_BK) / 16)));      // at least 16 K-steps per slice
            // (the slices' slabs: rem x sp x 256 KiB <= 64 MiB of the scratch, which a direct batch does not use otherwise)
            if (tail_on && full > 0 && sp >= 2 && b.scratch && b.n == 0 && (size_t)rem * sp * H_BM * H_BN * sizeof(float) <= WGRAD_SCRATCH_BYTES) {
                grp.whole = full; grp.split = sp; grp.tail = b.scratch;
            }
        }
        b.nd = 0;
        return t256 ? launch_gemm_tt256_group(grp, st) : launch_gemm_tt128_group(grp, st);
    }
    double flops = 0.0;
    for (int i = 0; i < b.nd; i++) flops += 2.0 * b.d[i].Mw * b.d[i].Nw * (double)K;
    cc_shared::ProfScope _all(cc_shared::SITE_ALL_GEMMS, st, flops);
    size_t slab_sum = 0;
    for (int i = 0; i < b.nd; i++) slab_sum += (((size_t)b.d[i].Mw * b.d[i].Nw * sizeof(float)) + 255) & ~size_t(255);
    const size_t room = WGRAD_SCRATCH_BYTES - b.used;
    // tile kernel: 256 x 256 (8 waves, fewer / fatter blocks) or 128 x 128; slice count: minimise rounds x (K-steps + fixed per-block
    // cost), rounds = ceil(blocks / CUs).  CC_WGRAD_TILE / CC_WGRAD_KS override (tuning knobs).
    static const int force_tile = []() { const char* e = cc_lab_env("CC_WGRAD_TILE"); return e ? atoi(e) : 0; }();
    static const int force_ks = []() { const char* e = cc_lab_env("CC_WGRAD_KS"); return e ? atoi(e) : 0; }();
    int best_tile = 128, best_ks = 1;
    double best_cost = 1e30;
    for (int tile : {256, 128}) {
        if (force_tile && tile != force_tile) continue;
        if (tile == 256 && (K % H_BK)) continue;
        const int bk = tile == 256 ? H_BK : G_BK, ksteps = K / bk;
        int total = 0;
        for (int i = 0; i < b.nd; i++) total += ((b.d[i].Mw + tile - 1) / tile) * ((b.d[i].Nw + tile - 1) / tile);
        // per K-step cost in units of a 128-tile step (measured: 256-tile step of 32 ~ 0.8 us for 4x the tile area, 128-tile step of 64 ~ 0.65 us)
        const double step_cost = tile == 256 ? 1.25 : 1.0, fixed = tile == 256 ? 16.0 : 14.0;
        for (int ks = 1; ks <= 8 && (size_t)ks * slab_sum <= room && ksteps / ks >= 8; ks++) {
            if (force_ks && ks != force_ks) continue;
            const int per = (ksteps + ks - 1) / ks;
            const int rounds = (total * ks + 255) / 256;
            const double cost = rounds * (per * step_cost + fixed);
            if (cost < best_cost) { best_cost = cost; best_tile = tile; best_ks = ks; }
        }
    }
    if (best_cost >= 1e30) return CC_ERR_STATE;      // no (tile, slices) fits the scratch: gemm_wgrad's admission check keeps this from happening
    const int bk = best_tile == 256 ? H_BK : G_BK, ksteps = K / bk;
    // slices are multiples of 64 deep for both kernels (k_chunk convention of GemmShape)
    const int kt64 = (K + G_BK - 1) / G_BK;
    const int per64 = (kt64 + best_ks - 1) / best_ks, ks = (kt64 + per64 - 1) / per64;
    (void)ksteps;
    int first = 0;
    for (int i = 0; i < b.nd; i++) {
        const auto& d = b.d[i];
        const size_t slab = (size_t)d.Mw * d.Nw;
        float* sc = b.scratch + b.used / sizeof(float);
        const int tiles = ((d.Mw + best_tile - 1) / best_tile) * ((d.Nw + best_tile - 1) / best_tile);
        grp.A[i] = d.X; grp.B[i] = d.Y;
        grp.g[i] = GemmShape{d.Mw, d.Nw, d.K, d.ldx, d.ldy, per64 * G_BK, 8, 0};
        grp.slab[i] = sc; grp.zstride[i] = slab; grp.ldc[i] = d.Nw;
        grp.first[i] = first;
        first += tiles * ks;
        b.it[b.n++] = WgradBatch::Item{sc, slab, ks, d.Nw, d.dW, d.ldw, slab / 4};
        b.used += (size_t)ks * slab * sizeof(float);
        b.used = (b.used + 255) & ~size_t(255);
    }
    for (int i = b.nd; i <= TT_GROUP_MAX; i++) grp.first[i] = first;
    b.nd = 0;
    return best_tile == 256 ? launch_gemm_tt256_group(grp, st) : launch_gemm_tt128_group(grp, st);
}

int wgrad_flush(WgradBatch& b, hipStream_t st) {
    const int rcg = wgrad_launch_group(b, st);
    if (rcg != CC_OK) return rcg;
    if (b.n == 0) return CC_OK;
    SlabItems items;
    size_t nmax = 0;
    for (int i = 0; i < b.n; i++) { items.it[i] = b.it[i]; nmax = std::max(nmax, b.it[i].n4); }
    hipLaunchKernelGGL(k_slab_reduce_multi, dim3((int)std::min<size_t>((nmax + 255) / 256, 1024), b.n), dim3(256), 0, st, items);
    b.n = 0;
    b.used = 0;
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}
// folds `ks` slabs at `slabs` into dW: immediately, or parked in the batch
static int wgrad_reduce(const float* slabs, size_t slab, int ks, int Nw, float* dW, int ldw, WgradBatch* batch, hipStream_t st) {
    const size_t n4 = slab / 4;
    if (batch) {
        batch->it[batch->n++] = WgradBatch::Item{slabs, slab, ks, Nw, dW, ldw, n4};
        batch->used += (size_t)ks * slab * sizeof(float);
        batch->used = (batch->used + 255) & ~size_t(255);
        if (batch->n == 8) return wgrad_flush(*batch, st);
        return CC_OK;
    }
    hipLaunchKernelGGL(k_slab_reduce, dim3((int)std::min<size_t>((n4 + 255) / 256, 2048)), dim3(256), 0, st, slabs, slab, ks, Nw, dW, ldw, n4);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

int gemm_wgrad(const act_t* Xa, int ldx, const act_t* Ya, int ldy, int Mw, int Nw, int K, float* dW, int ldw, float* scratch,
               hipStream_t st, WgradBatch* batch) {
    if ((Nw & 7) || (ldw & 3)) return CC_ERR_SHAPE;
    const double flops = 2.0 * Mw * Nw * (double)K;
#if CC_OP == 2
    // bf16x3: dW = X^T Y = Xhi^T Yhi + Xhi^T Ylo + Xlo^T Yhi.  The [K][3 Mw] image [hi | hi | lo] of X, read as a [3K][Mw] matrix, has
    // rows (hi, hi, lo) per source row, the [hi | lo | hi] image of Y rows (hi, lo, hi): the unchanged kernels contract over K' = 3K.
    // The images live in the call's scratch, so nothing can be deferred to a grouped launch.
    if (Mw & 7) return CC_ERR_SHAPE;
    int rcx = CC_OK;
    // the first operand (the output gradient) has the [hi | hi | lo] form the input-gradient GEMM of the same tensor reads as its A operand:
    // a caller that has split it once for both passes it as an image (x3_expect_image)
    const bool ximg = x3_take_expected(Xa);
    const op16_t* X = ximg ? reinterpret_cast<const op16_t*>(Xa) : x3_operand(Xa, (size_t)ldx, K, Mw, 0, true, st, &rcx);
    if (!X) return rcx;
    const op16_t* Y = x3_operand(Ya, (size_t)ldy, K, Nw, 1, ximg, st, &rcx);
    if (!Y) return rcx;
    ldx = Mw; ldy = Nw; K *= 3;
    const bool may_defer = false;
#else
    const op16_t* X = Xa;
    const op16_t* Y = Ya;
    const bool may_defer = true;
#endif
    const size_t slab = (size_t)Mw * Nw;
    // the DMA-staged TT kernels need 16-B aligned operands and row strides; anything else takes the register-staged kernel
    const bool tt_ok = (Mw & 7) == 0 && (ldx & 7) == 0 && (ldy & 7) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 15) == 0;
    // scratch left behind the batch's parked slabs; a gradient that cannot get at least 2 slices there flushes the batch first
    if (batch && scratch && batch->n > 0 && WGRAD_SCRATCH_BYTES - batch->used < 2 * slab * sizeof(float)) {
        const int rcf = wgrad_flush(*batch, st);
        if (rcf != CC_OK) return rcf;
    }
    // grouped path: park the problem; wgrad_flush launches the layer's gradients together (one tail, one launch floor)
    // a deferred problem needs at least one slab behind the parked ones and the other deferred problems' slabs (wgrad_launch_group then
    // finds a slice count that fits); a gradient whose single slab exceeds the scratch (GPT-2-xl width: c_fc 4 D^2 fp32 = 41 MB is fine,
    // but 12 D^2 = 123 MB for the layer is not) makes the group flush early, one larger than the whole scratch is never deferred
    const size_t slab_bytes = ((slab * sizeof(float)) + 255) & ~size_t(255);
    if (may_defer && batch && batch->defer && g_gemm_tile_mode != 0 && tt_ok && scratch && (K % G_BK) == 0 && K >= 1024 && (slab & 3) == 0 &&
        slab_bytes <= WGRAD_SCRATCH_BYTES && (batch->nd == 0 || batch->d[0].K == K)) {
        size_t pending = 0;
        for (int i = 0; i < batch->nd; i++) pending += (((size_t)batch->d[i].Mw * batch->d[i].Nw * sizeof(float)) + 255) & ~size_t(255);
        const int cap = std::min(std::max(batch->cap, 1), 32);
        if (batch->nd == cap || (!batch->direct && (batch->n + batch->nd >= 8 || batch->used + pending + slab_bytes > WGRAD_SCRATCH_BYTES))) {
            const int rcf = wgrad_flush(*batch, st);
            if (rcf != CC_OK) return rcf;
        }
        batch->scratch = scratch;
        batch->d[batch->nd++] = WgradBatch::Deferred{X, Y, ldx, ldy, Mw, Nw, K, dW, ldw};
        return CC_OK;
    }
    cc_shared::ProfScope _all(cc_shared::SITE_ALL_GEMMS, st, flops);
    const size_t used = (batch && scratch) ? batch->used : 0;
    float* sc = scratch ? scratch + used / sizeof(float) : nullptr;
    const size_t fit = scratch ? (WGRAD_SCRATCH_BYTES - used) / (slab * sizeof(float)) : 1;
    // 256 x 256 kernel with DMA staging + transpose reads (one block per CU; slices sized to fill the CUs once).  It wins from
    // about 30 GFLOP per launch (GPT-2 weight gradients: +10...17 %); below that the 128 x 128 TT kernel's finer tiles and fewer,
    // smaller slabs win (mapper weight gradients, K = 5120) — tools/wgrad_bench.py.
    if (g_gemm_tile_mode != 0 && tt_ok && (K % H_BK) == 0 && (g_gemm_tile_mode == 4 || 2.0 * Mw * Nw * (double)K >= 3e10)) {
        const int tiles = ((Mw + H_BM - 1) / H_BM) * ((Nw + H_BN - 1) / H_BN);
        int ks = std::max(1, 256 / tiles);
        ks = std::min(ks, std::max(1, K / 512));          // at least 16 K-tiles per slice
        if ((size_t)ks > fit) ks = (int)std::max<size_t>(fit, 1);
        if (ks == 1) {                                    // wide outputs (lm_head): accumulate straight into dW
            EpiF32 e{dW, nullptr, ldw, Mw, Nw, 1, 1.0f};
            return launch_gemm_tt256(X, ldx, Y, ldy, Mw, Nw, K, 1, e, nullptr, st);
        }
        EpiF32 e{sc, nullptr, Nw, Mw, Nw, 3, 1.0f};
        e.zstride = slab;
        int ks_eff = 1;
        const int rc = launch_gemm_tt256(X, ldx, Y, ldy, Mw, Nw, K, ks, e, &ks_eff, st);
        if (rc != CC_OK) return rc;
        return wgrad_reduce(sc, slab, ks_eff, Nw, dW, ldw, batch, st);
    }
    const int tiles = ((Mw + G_BM - 1) / G_BM) * ((Nw + G_BN - 1) / G_BN);
    // small weight gradients (mapper, K = 5120): 128 x 128 TT kernel, 4-stage DMA pipeline + transpose reads, one block per CU
    if (g_gemm_tile_mode != 0 && tt_ok && (K % G_BK) == 0 && K >= 1024 && tiles <= 256 && scratch && fit >= 1) {

        int ks = std::max(1, 256 / tiles);
        ks = std::min(ks, std::max(1, K / (4 * G_BK)));
        if ((size_t)ks > fit) ks = (int)fit;
        EpiF32 e{sc, nullptr, Nw, Mw, Nw, 3, 1.0f};
        e.zstride = slab;
        int ks_eff = 1;
        const int rc = launch_gemm_tt128(X, ldx, Y, ldy, Mw, Nw, K, ks, e, &ks_eff, st);
        if (rc != CC_OK) return rc;
        return wgrad_reduce(sc, slab, ks_eff, Nw, dW, ldw, batch, st);
    }
    int ks = 512 / (tiles > 0 ? tiles : 1);
    const int kmax = (K + 255) / 256;  // at least 4 K-steps per slice
    if (ks > kmax) ks = kmax;
    if (scratch) { if ((size_t)ks > fit) ks = (int)fit; } else ks = 1;
    if (ks <= 1) return gemm_f32out_op(1, 1, X, ldx, Y, ldy, Mw, Nw, K, dW, ldw, nullptr, 1, 1.0f, 1, st);
    // slices z write slab z (EpiF32 store mode; C pointer advanced per z inside the kernel via blockIdx.z * slab)
    EpiF32 e{sc, nullptr, Nw, Mw, Nw, 3, 1.0f};
    e.zstride = slab;
    int rc = launch_gemm(1, 1, X, ldx, Y, ldy, Mw, Nw, K, ks, e, st);
    if (rc != CC_OK) return rc;
    // the launcher may have reduced the slice count (ceil division): recompute it the same way
    const int kt = (K + G_BK - 1) / G_BK, per = (kt + ks - 1) / ks;
    const int ks_eff = (kt + per - 1) / per;
    return wgrad_reduce(sc, slab, ks_eff, Nw, dW, ldw, batch, st);
}

__global__ __launch_bounds__(256) void k_splitk_finish(const float* __restrict__ slabs, size_t slab_elems, int ks, int M, int N,
                                                       const float* __restrict__ bias, int act, const float* __restrict__ res,
                                                       float* __restrict__ out32, act_t* __restrict__ out16, int ldo) {
    const size_t n8 = (size_t)M * (N >> 3);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / (N >> 3)), col = (int)(i % (N >> 3)) * 8;
        const size_t e = (size_t)row * N + col;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < ks; s++) {
            const float4 a = *reinterpret_cast<const float4*>(slabs + s * slab_elems + e);
            const float4 b = *reinterpret_cast<const float4*>(slabs + s * slab_elems + e + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        if (bias) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] += bias[col + k];
        }
        if (act == 1) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = fmaxf(v[k], 0.f);
        } else if (act == 2) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = gelu_new_f(v[k]);
        }
        const size_t o = (size_t)row * ldo + col;
        if (res) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] += res[o + k];
        }
        if (out32) {
            *reinterpret_cast<float4*>(out32 + o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(out32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (out16) act_st8(out16 + o, v);
    }
}

// one 256-thread block per finished row (M is small here, so rows are the only parallelism): slabs -> (+bias, act, +residual) ->
// fp32 / bf16 stores -> optional KV append -> optional LayerNorm (block reduction) -> bf16
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void k_splitk_finish_row(const float* __restrict__ slabs, size_t slab_elems, int ks, int M, int N,
                                                           const float* __restrict__ bias, int act, const float* __restrict__ res,
                                                           float* __restrict__ out32, act_t* __restrict__ out16, int ldo, SkinnyFuse f) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    constexpr int MAXV = 3;                // float4 per thread -> N <= 3072
    float4 v[MAXV];
    float s1 = 0.f;
    // LayerNorm parameters are fetched up front, beside the slab loads, not after the two block reductions
    float4 lg_[MAXV], lb_[MAXV];
    if (f.ln_out16) {
#pragma unroll
        for (int it = 0; it < MAXV; it++) {
            const int c = threadIdx.x * 4 + it * 1024;
            if (c < N) { lg_[it] = *reinterpret_cast<const float4*>(f.ln_gamma + c); lb_[it] = *reinterpret_cast<const float4*>(f.ln_beta + c); }
        }
    }
#pragma unroll
    for (int it = 0; it < MAXV; it++) {
        const int c = threadIdx.x * 4 + it * 1024;
        if (c < N) {
            float4 a = make_float4(0, 0, 0, 0);
            const float* sp = slabs + (size_t)row * N + c;
            float4 bb = make_float4(0, 0, 0, 0), rr = make_float4(0, 0, 0, 0);
            if (bias) bb = *reinterpret_cast<const float4*>(bias + c);
            if (res) rr = *reinterpret_cast<const float4*>(res + (size_t)row * ldo + c);
            int s = 0;
            for (; s + 4 <= ks; s += 4) {                      // four slices in flight
                const float4 p0 = *reinterpret_cast<const float4*>(sp + (s + 0) * slab_elems), p1 = *reinterpret_cast<const float4*>(sp + (s + 1) * slab_elems);
                const float4 p2 = *reinterpret_cast<const float4*>(sp + (s + 2) * slab_elems), p3 = *reinterpret_cast<const float4*>(sp + (s + 3) * slab_elems);
                a.x += (p0.x + p1.x) + (p2.x + p3.x); a.y += (p0.y + p1.y) + (p2.y + p3.y);
                a.z += (p0.z + p1.z) + (p2.z + p3.z); a.w += (p0.w + p1.w) + (p2.w + p3.w);
            }
            if (ks - s == 3) {
                const float4 p0 = *reinterpret_cast<const float4*>(sp + (s + 0) * slab_elems), p1 = *reinterpret_cast<const float4*>(sp + (s + 1) * slab_elems);
                const float4 p2 = *reinterpret_cast<const float4*>(sp + (s + 2) * slab_elems);
                a.x += (p0.x + p1.x) + p2.x; a.y += (p0.y + p1.y) + p2.y; a.z += (p0.z + p1.z) + p2.z; a.w += (p0.w + p1.w) + p2.w;
            } else {
                for (; s < ks; s++) {
                    const float4 p = *reinterpret_cast<const float4*>(sp + s * slab_elems);
                    a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
                }
            }
            a.x += bb.x; a.y += bb.y; a.z += bb.z; a.w += bb.w;
            if (act == 2) { a.x = gelu_new_f(a.x); a.y = gelu_new_f(a.y); a.z = gelu_new_f(a.z); a.w = gelu_new_f(a.w); }
            else if (act == 1) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
            const size_t o = (size_t)row * ldo + c;
            a.x += rr.x; a.y += rr.y; a.z += rr.z; a.w += rr.w;
            if (out32) *reinterpret_cast<float4*>(out32 + o) = a;
            const act_raw4 pk = act_pack4(a.x, a.y, a.z, a.w);
            if (out16) act_straw4(out16 + o, pk);
            if (f.kcache) {                   // N == 3*D
                const int D = N / 3, which = c / D, cc_ = c - which * D;
                if (which > 0) {
                    const int r = row / f.Tn, t = row - r * f.Tn;
                    act_t* dst = (which == 1 ? f.kcache : f.vcache) + ((size_t)r * f.ctx_max + f.pos0 + t) * D + cc_;
                    act_straw4(dst, pk);
                }
            }
            v[it] = a;
            s1 += a.x + a.y + a.z + a.w;
        }
    }
    if (!f.ln_out16) return;               // block-uniform
    const float mu = block_sum256(s1, red) / N;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < MAXV; it++) {
        const int c = threadIdx.x * 4 + it * 1024;
        if (c < N) { const float a = v[it].x - mu, b = v[it].y - mu, c2 = v[it].z - mu, d = v[it].w - mu; q += a * a + b * b + c2 * c2 + d * d; }
    }
    const float rs = rsqrtf(block_sum256(q, red) / N + 1e-5f);
#pragma unroll
    for (int it = 0; it < MAXV; it++) {
        const int c = threadIdx.x * 4 + it * 1024;
        if (c < N) {
            const float4 g = lg_[it], b = lb_[it];
            act_st4(f.ln_out16 + (size_t)row * N + c, (v[it].x - mu) * rs * g.x + b.x, (v[it].y - mu) * rs * g.y + b.y,
                    (v[it].z - mu) * rs * g.z + b.z, (v[it].w - mu) * rs * g.w + b.w);
        }
    }
}

int skinny_single_min_tiles() {
    static const int v = []() { const char* e = cc_lab_env("CC_SKINNY_SINGLE"); return e ? atoi(e) : 60; }();   // measured on config 5: 171 (never) / 165 (90) / 159 (60) / 160 (20) ms per decode batch
    return v;
}
bool gemm_nt_skinny_can_fuse(int M, int N, int K, size_t scratch_bytes) {
    return M > 0 && N > 0 && (N & 7) == 0 && N <= 3072 && (K % G_BK) == 0 && scratch_bytes >= (size_t)M * N * sizeof(float);
}

// slab sum of the K-sliced lm_head input gradient in the exponential form: out = r * sum(slabs) - w * wte[target]
__global__ __launch_bounds__(256) void k_deepk_finish_lm(const float* __restrict__ slabs, size_t slab_elems, int ks, int M, int N, LmFix f,
                                                         act_t* __restrict__ out16, int ldo) {
    const size_t n8 = (size_t)M * (N >> 3);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / (N >> 3)), col = (int)(i % (N >> 3)) * 8;
        const size_t e = (size_t)row * N + col;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8];
        for (int s = 0; s < ks; s++) {
            const float4 a = *reinterpret_cast<const float4*>(slabs + s * slab_elems + e);
            const float4 c = *reinterpret_cast<const float4*>(slabs + s * slab_elems + e + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += c.x; v[5] += c.y; v[6] += c.z; v[7] += c.w;
        }
        const float r = f.fac[2 * row], w = f.fac[2 * row + 1];
#if CC_OP == 2
        act_ld8(reinterpret_cast<const float*>(f.wte) + (size_t)f.target[row] * N + col, b);      // bf16x3: LmFix.wte is the fp32 master
#else
        unpack8(*reinterpret_cast<const uint4*>(f.wte + (size_t)f.target[row] * N + col), b);
#endif
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = r * v[k] - w * b[k];
        act_st8(out16 + (size_t)row * ldo + col, v);
    }
}

int gemm_nt_deepk(const act_t* Aa, int lda, const op16_t* B, int ldb, int M, int N, int K, act_t* out16, int ldo, float* scratch,
                  size_t scratch_bytes, hipStream_t st, const LmFix* fix) {
    static const int knob = []() { const char* e = cc_lab_env("CC_DEEPK"); return e ? atoi(e) : -1; }();   // 0 = off, n > 0 = force n slices
    if (knob == 0 || g_gemm_tile_mode == 0 || !scratch || (N & 7) || (ldo & 7) || (K % 64) || (lda & 7) || (ldb & 7)) return CC_ERR_SHAPE;
    const long tiles = (long)((M + H_BM - 1) / H_BM) * ((N + H_BN - 1) / H_BN);
    const size_t slab = (size_t)M * N;
    int ks = knob > 0 ? knob : (int)(256 / (tiles > 0 ? tiles : 1));
    const size_t fit = scratch_bytes / (slab * sizeof(float));
    if ((size_t)ks > fit) ks = (int)fit;
    if (ks > K / 2048) ks = K / 2048;                      // slices of at least 64 K-steps: below that the slab pass costs what the idle CUs gain
    if (ks < 2 || tiles > 128) return CC_ERR_SHAPE;
    cc_shared::ProfScope _all(cc_shared::SITE_ALL_GEMMS, st, 2.0 * M * N * (double)K);
    const op16_t* A;
    CC_X3_NT(Aa, lda, ldb, M, K, A, 0, 0, st);
    EpiF32 e{scratch, nullptr, N, M, N, 3, 1.0f};
    e.zstride = slab;
    // tile order: an XCD's share of the grid (tiles / 8 consecutive logical tiles per K slice) should hold whole row panels, so that the
    // column tiles of a panel run side by side on one L2 and the deep A panel is fetched once, not once per column tile
    int tile = 4, tiles_n = (N + H_BN - 1) / H_BN;
    long tl = tiles;
    if (kX3 && x3_fused_on() && (K % (3 * H_BK)) == 0) {
        // bf16x3: the fused two-stage form lives on the 256 x 192 kernel (gemm_stag256_body<X3F>): its tile count, and the slice count
        // that minimises rounds x chunks per slice (whole rounds of 256 workgroups)
        tile = 3;
        tiles_n = (N + 191) / 192;
        tl = (long)((M + H_BM - 1) / H_BM) * tiles_n;
        const int nc = K / (3 * H_BK);
        long best = -1;
        int best_ks = ks;
        for (int k2 = 1; k2 <= 8 && (size_t)k2 <= fit && k2 <= nc / 32; k2++) {
            const long cost = ((tl * k2 + 255) / 256) * ((nc + k2 - 1) / k2);
            if (best < 0 || cost < best) { best = cost; best_ks = k2; }
        }
        ks = best_ks;
        if (ks < 2) return CC_ERR_SHAPE;
    }
    const int gm = std::max(1, (int)((tl + 7) / 8) / tiles_n);
    const int rc = launch_gemm(0, 0, A, lda, B, ldb, M, N, K, ks, e, st, tile, gm);
    if (rc != CC_OK) return rc;
    const int kt = K / G_BK, per = (kt + ks - 1) / ks, ks_eff = (kt + per - 1) / per;
    const size_t n8 = (size_t)M * (N >> 3);
    if (fix)
        hipLaunchKernelGGL(k_deepk_finish_lm, dim3((int)std::min<size_t>((n8 + 255) / 256, 2048)), dim3(256), 0, st, scratch, slab, ks_eff, M, N, *fix,
                           out16, ldo);
    else
        hipLaunchKernelGGL(k_splitk_finish, dim3((int)std::min<size_t>((n8 + 255) / 256, 2048)), dim3(256), 0, st, scratch, slab, ks_eff, M, N, nullptr, 0,
                           nullptr, nullptr, out16, ldo);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

#ifdef CC_EXPERIMENTS
int skinny_image(const op16_t* W, op16_t* img, int N, int K, hipStream_t st) {
    if ((N % 64) || (K % 64) || !W || !img) return CC_ERR_SHAPE;
    hipLaunchKernelGGL(k_skinny_image, dim3(1024), dim3(256), 0, st, W, img, N, K);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}
#endif

int gemm_nt_skinny(const act_t* Aa, int lda, const op16_t* B, int ldb, int M, int N, int K, const float* bias, int act, const float* res,
                   float* out32, act_t* out16, int ldo, float* scratch, size_t scratch_bytes, hipStream_t st, const SkinnyFuse* fuse) {
    cc_shared::ProfScope _all(cc_shared::SITE_ALL_GEMMS, st, 2.0 * M * N * (double)K);
    if ((N & 7) || (ldo & 7)) return CC_ERR_SHAPE;
    const op16_t* A;
    CC_X3_NT(Aa, lda, ldb, M, K, A, 0, 0, st);
    const int tiles = ((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
    const size_t slab = (size_t)M * N;
    const bool can_slab = scratch && slab && (K % G_BK) == 0 && scratch_bytes >= slab * sizeof(float);
    const bool fused = fuse && (fuse->ln_out16 || fuse->kcache);
    const op16_t* bimg = (fuse && !kX3) ? fuse->bimg : nullptr;
    // decode-sized M: 64 x 64 tiles (gemm_nt_s64_kernel) put 2.5-5x the blocks on the chip; a block's K loop is bound by what one CU
    // can pull through its L2->LDS path (~57 GB/s measured), so the job is to have every CU pulling
    const bool s64 = g_gemm_s64 != 0 && (K % G_BK) == 0 && M <= 640 && tiles <= 256 && (lda & 7) == 0 && (ldb & 7) == 0;
    int ks;
    if (s64) {
        const int t64 = ((M + 63) / 64) * ((N + 63) / 64);
        ks = t64 >= 160 ? 1 : 256 / t64;                    // K slices only while they add blocks up to one per CU
        const int kmax = K / (4 * G_BK);
        if (ks > kmax) ks = kmax;
    } else {
        ks = 384 / (tiles > 0 ? tiles : 1);
        const int kmax = K / (2 * G_BK);                    // at least 2 K-steps per slice
        if (ks > kmax) ks = kmax;
        // wide-enough grids go single-pass with the epilogue fused into the GEMM (4-stage small-grid kernel): one launch instead of two
        if (!fused && tiles >= skinny_single_min_tiles()) ks = 1;
    }
    if (ks < 1) ks = 1;
    if (can_slab) { const size_t fit = scratch_bytes / (slab * sizeof(float)); if ((size_t)ks > fit) ks = (int)fit; }
    // K split over the waves of a block (gemm_nt_s64kw_kernel): pays from ~14 K-tiles per block, and only on grids of at most one
    // block per CU (320 x 3072 x 1024: 9.0 -> 8.3 us, 320 x 1024 x 4096 in one slice: 23.6 -> 18.4 us; c_fc's 320 blocks: no gain)
    // (round 5) a single-slice grid of 257-320 64 x 64 tiles that is <= 256 tiles of 80 x 64 (c_fc at M = 320: 5 x 64 -> 4 x 64) takes the
    // 80-row form of the same kernel: one block per CU instead of two on a quarter of them (CC_SKINNY_80=0 in the lab build: off)
    static const bool s80_on = []() { const char* e = cc_lab_env("CC_SKINNY_80"); return !e || atoi(e) != 0; }();
    auto s64_form = [&](int slices) {
        const int t64 = ((M + 63) / 64) * ((N + 63) / 64), t80 = ((M + 79) / 80) * ((N + 63) / 64);
        if (g_gemm_s64 == 1 || K / G_BK / (slices > 0 ? slices : 1) < 14) return 1;
        if ((long)t64 * slices <= 256) return 3;
        return (s80_on && slices <= 1 && t80 <= 256) ? 4 : 1;      // (no weight-image variant: both operand paths run this kernel)
    };
    if (!can_slab || (ks <= 1 && !fused)) {
        if (fused) return CC_ERR_SHAPE;                    // callers only request fusion when the slab path is available
        if (s64) {
            const int form = s64_form(1);
            if (res) { EpiResid e{out32, res, bias, ldo, M, N}; return launch_gemm_s64(A, lda, B, ldb, M, N, K, 1, form, e, nullptr, st, bimg); }
            if (out16) {
                // decode's c_attn (plain) and c_fc (gelu_new) launches: the functors without run-time switches (round 6)
                static const bool spec_on = []() { const char* v = cc_lab_env("CC_EPI_SPEC"); return !v || atoi(v) != 0; }();
                if (spec_on && act == 0) { EpiBF16Plain e{out16, bias, ldo, M, N}; return launch_gemm_s64(A, lda, B, ldb, M, N, K, 1, form, e, nullptr, st, bimg); }
                if (spec_on && act == 2) { EpiBF16T<2, 0> e{out16, nullptr, bias, ldo, M, N, 2}; return launch_gemm_s64(A, lda, B, ldb, M, N, K, 1, form, e, nullptr, st, bimg); }
                EpiBF16 e{out16, nullptr, bias, ldo, M, N, act};
                return launch_gemm_s64(A, lda, B, ldb, M, N, K, 1, form, e, nullptr, st, bimg);
            }
            EpiF32 e{out32, bias, ldo, M, N, 0, 1.0f};
            return launch_gemm_s64(A, lda, B, ldb, M, N, K, 1, form, e, nullptr, st, bimg);
        }
        if (res) { EpiResid e{out32, res, bias, ldo, M, N}; return launch_gemm(0, 0, A, lda, B, ldb, M, N, K, 1, e, st); }
        if (out16) { EpiBF16 e{out16, nullptr, bias, ldo, M, N, act}; return launch_gemm(0, 0, A, lda, B, ldb, M, N, K, 1, e, st); }
        EpiF32 e{out32, bias, ldo, M, N, 0, 1.0f};
        return launch_gemm(0, 0, A, lda, B, ldb, M, N, K, 1, e, st);
    }
    EpiF32 e{scratch, nullptr, N, M, N, 3, 1.0f};
    e.zstride = slab;
    int rc = s64 ? launch_gemm_s64(A, lda, B, ldb, M, N, K, ks, s64_form(ks), e, nullptr, st, bimg) : launch_gemm(0, 0, A, lda, B, ldb, M, N, K, ks, e, st);
    if (rc != CC_OK) return rc;
    const int kt = K / G_BK, per = (kt + ks - 1) / ks, ks_eff = (kt + per - 1) / per;
    if (N <= 3072 && (N & 3) == 0) {
        SkinnyFuse f = fuse ? *fuse : SkinnyFuse{};
        hipLaunchKernelGGL(k_splitk_finish_row, dim3(M), dim3(256), 0, st, scratch, slab, ks_eff, M, N, bias, act, res, out32, out16, ldo, f);
    } else {
        if (fused) return CC_ERR_SHAPE;
        const size_t n8 = (size_t)M * (N >> 3);
        hipLaunchKernelGGL(k_splitk_finish, dim3((int)std::min<size_t>((n8 + 255) / 256, 2048)), dim3(256), 0, st, scratch, slab, ks_eff, M, N, bias,
                           act, res, out32, out16, ldo);
    }
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}
}  // namespace CC_NS
