// Host entry points of the GEMM instantiations (one translation unit per epilogue so they build in parallel).
#pragma once
#include "common.hip.h"
#include "shared.h"

namespace CC_NS {
// NT tile choice (cc_shared::g_gemm_tile_mode): -1 chooser (default; CC_GEMM_S256 in the environment presets it), 0 = 128 x 128 only,
// 3 / 4 / 5 = force the 256 x 192 / 256 x 256 / 320 x 256 kernel wherever it is legal.  g_gemm_s64 > 0: route NT GEMMs with M <= 1024 through the 64-row
// skinny kernel (env CC_GEMM_S64); g_gemm_small_x2: small-grid NT GEMMs on the 8-wave kernel (env CC_GEMM_X2, default 1).  All three
// are test / microbenchmark hooks living in shared.cpp.
// al/bl: 0 = K-contiguous operand ([rows][K]), 1 = K-strided operand ([K][rows]).  See gemm.hip.h.
//
// Operand types.  A is an ACTIVATION (act_t), B a WEIGHT from the 16-bit operand arena, C / pre / aux activations again.  In the bf16 /
// fp16 builds act_t is the 16-bit operand type and the calls are what they say.  In the bf16x3 build (common.hip.h) activations are
// fp32: every wrapper below first splits A into its [hi | hi | lo] operand image in the call's scratch (x3_set_scratch, set by the
// C-ABI entry point from its workspace), B is the pre-split [hi | lo | hi] image the weight sync left in the operand arena
// (row stride 3 * ldb), and the unchanged NT kernels run with K' = 3 K.  Callers pass the LOGICAL lda / ldb / K in every build.
// Only al = bl = 0 is supported there.
int gemm_bf16out(int al, int bl, const act_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, act_t* C, int ldc,
                 const float* bias, int act, act_t* pre, hipStream_t st);
int gemm_resid(int al, int bl, const act_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, float* out, const float* res,
               int ld, const float* bias, hipStream_t st, Drop drop = Drop());
// mode 0 store (+bias), 1 add, 2 atomic add (required when ksplit > 1)
int gemm_f32out(int al, int bl, const act_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, float* C, int ldc,
                const float* bias, int mode, float alpha, int ksplit, hipStream_t st);
int gemm_dact(int al, int bl, const act_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, act_t* C, int ldc,
              const act_t* aux, int act, hipStream_t st);
// cref != nullptr: exponential form (C = exp(logit - cref[row]), see gemm.hip.h EpiLMHead)
int gemm_lmhead(const act_t* A, int lda, const op16_t* B, int ldb, int M, int Vp, int V, int K, act_t* C, int ldc, float* pmax,
                float* psum, int npart, const int* target, float* tgt_logit, hipStream_t st, const float* cref = nullptr);

// decode lm_head: fp32 logits [M][ldc] (Ns columns stored, Ns % 8 == 0) + per-(row, 64-column block) softmax partials over the V real
// columns (pmax / psum [M][npart], npart >= ceil(Ns / 64)) — gemm.hip.h EpiLogits
int gemm_logits_part(const act_t* A, int lda, const op16_t* B, int ldb, int M, int Ns, int V, int K, float* C, int ldc, float* pmax, float* psum,
                     int npart, hipStream_t st);

#if CC_OP == 2
// bf16x3: scratch for the operand images of the GEMM being launched (stream order makes reuse by the next GEMM safe).  Thread-local:
// the C ABI stays re-entrant across host threads / streams; every compute entry point sets it from its own workspace.
void x3_set_scratch(void* base, size_t bytes);
// split `rows` x `width` fp32 (row stride ld) into the scratch: form 0 = [hi | hi | lo] (A side), 1 = [hi | lo | hi] (B side).
// first = true restarts the scratch (one GEMM's operands live there at a time).  Returns nullptr (rc set) when it does not fit.
const op16_t* x3_operand(const float* src, size_t ld, int rows, int width, int form, bool first, hipStream_t st, int* rc);
// Producer-written operand images (round 4): x3_expect_image(p) before a GEMM wrapper call says "the A pointer p of the next call already
// IS its [hi | hi | lo] image" (written by the previous GEMM's epilogue, epi_store8) — the split pass is skipped; x3_emit_image(p, n) says
// "write the output C == p of the next gemm_bf16out / gemm_dact call as the image of an n-wide A operand".  Thread-local one-shot hints
// (consumed by the next matching call), like the scratch: the C ABI stays re-entrant.
op16_t* x3_scratch_block(size_t bytes);
void x3_expect_image(const void* a);
bool x3_take_expected(const void* a);
void x3_emit_image(const void* c, int width);
int x3_take_emit(const void* c);
#define CC_X3_NT(A, lda, ldb, M, K, A16, al, bl, st)                                      \
    {                                                                                     \
        if ((al) || (bl)) return CC_ERR_ARG;                                              \
        if (x3_take_expected(A)) {                                                        \
            A16 = reinterpret_cast<const op16_t*>(A);                                     \
        } else {                                                                          \
            int rc_ = CC_OK;                                                              \
            A16 = x3_operand(A, (size_t)(lda), M, K, 0, true, st, &rc_);                  \
            if (!A16) return rc_;                                                         \
        }                                                                                 \
        lda = 3 * (K); ldb = 3 * (ldb); K = 3 * (K);                                      \
    }
#else
#define CC_X3_NT(A, lda, ldb, M, K, A16, al, bl, st) A16 = A;
#endif
// weight gradient dW[Mw][Nw] += X^T Y with X stored [K][Mw], Y stored [K][Nw].  Split-K for occupancy: the K slices
// write fp32 slabs into `scratch` (plain stores) and a second kernel folds them into dW — fp32 atomics on the same
// tile from 7-14 concurrent blocks measured 3x slower than the whole GEMM (profiles/r01_b_gemm_microbench.md).
constexpr size_t WGRAD_SCRATCH_BYTES = size_t(96) << 20;
// Several weight gradients of one layer can share one slab-reduce launch: pass a WgradBatch, each gemm_wgrad then parks its slabs
// in its own part of `scratch` and wgrad_flush() folds all of them into their dW targets (8 launches per mapper backward instead
// of 32).  A gradient whose slabs do not fit behind the parked ones flushes the batch first.
struct WgradBatch {
    struct Item { const float* slabs; size_t slab; int ks, Nw; float* dW; int ldw; size_t n4; };
    Item it[8];
    int n = 0;
    size_t used = 0;      // bytes of scratch already holding parked slabs
    // defer = true: weight gradients that fit the 128 x 128 TT kernel are not launched by gemm_wgrad but collected (up to 4, same K)
    // and run by wgrad_flush as ONE grouped launch (gemm_tt_glds4_group_kernel) + one slab reduce.  The caller must keep every
    // operand unchanged until the flush.
    bool defer = false;
    struct Deferred { const op16_t* X; const op16_t* Y; int ldx, ldy, Mw, Nw, K; float* dW; int ldw; };      // (never used in the bf16x3 build: operands are split per call)
    Deferred d[32];
    int nd = 0;
    int cap = 4;          // deferred problems per grouped launch (<= 32)
    // direct = true (round 5; with defer): the deferred problems run as ONE launch with a single K slice per tile that adds into dW in its
    // epilogue — no slabs, no reduce launch; meant for many problems at once (all layers of a mapper backward: 576 tiles of 256 x 256),
    // where whole-K tiles still fill the CUs.  The caller keeps every operand unchanged until the flush.
    bool direct = false;
    float* scratch = nullptr;
};
int wgrad_flush(WgradBatch& b, hipStream_t st);
int gemm_wgrad(const act_t* X, int ldx, const act_t* Y, int ldy, int Mw, int Nw, int K, float* dW, int ldw, float* scratch,
               hipStream_t st, WgradBatch* batch = nullptr);
// Skinny-M NT GEMMs (KV-cached decode: M = beams x samples, a handful of 128x128 tiles): split K over blockIdx.z so every CU
// streams a distinct slice of the weights, fp32 slabs in `scratch`, then ONE finishing kernel sums the slabs and applies the
// epilogue (bias, gelu_new, fp32 residual, bf16 / fp32 stores).  Falls back to the single-pass GEMM when the grid is already wide.
// Optional fused tails of the finishing kernel (decode): LayerNorm of the finished fp32 row (N == row width) -> bf16, and the
// append of the K / V thirds of a finished qkv row to the KV cache.
struct SkinnyFuse {
    const float* ln_gamma = nullptr;   // LayerNorm over the N columns of (acc + bias + res): out -> ln_out16 [M, N]
    const float* ln_beta = nullptr;
    act_t* ln_out16 = nullptr;
    act_t* kcache = nullptr;           // qkv rows (N == 3*D): columns [D,2D) -> kcache, [2D,3D) -> vcache at (r*ctx_max + pos0 + t)*D
    act_t* vcache = nullptr;
    int Tn = 1, pos0 = 0, ctx_max = 0;
    const op16_t* bimg = nullptr;      // the weight's fragment-ordered image (k_skinny_image; cc_decode_image): B is then loaded global -> VGPR where the form allows
};
int skinny_image(const op16_t* W, op16_t* img, int N, int K, hipStream_t st);
int gemm_nt_skinny(const act_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, const float* bias, int act, const float* res,
                   float* out32, act_t* out16, int ldo, float* scratch, size_t scratch_bytes, hipStream_t st,
                   const SkinnyFuse* fuse = nullptr);
// Narrow output, very deep K (the lm_head input gradient: [B*cap, Vp] x [Vp, D] -> 10240 x 768 over K = 50304): the 256 x 256 kernel on
// 120 tiles leaves half the chip idle, so K is cut into as many slices as fill the CUs (fp32 slabs in `scratch`), and one elementwise
// pass sums the slabs into the 16-bit output.  Returns CC_ERR_SHAPE when the shape does not call for it (caller then uses gemm_bf16out).
// fix (optional): out[m][n] = fac[2m] * acc - fac[2m+1] * wte[target[m]][n] — the lm_head input gradient of the exponential form
struct LmFix { const float* fac; const int* target; const op16_t* wte; };
int gemm_nt_deepk(const act_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, act_t* out16, int ldo, float* scratch,
                  size_t scratch_bytes, hipStream_t st, const LmFix* fix = nullptr);
int skinny_single_min_tiles();   // grids of at least this many 128 x 128 tiles skip split-K (CC_SKINNY_SINGLE; tuning knob)
// whether gemm_nt_skinny will take the slab + row-finish path for this problem (the only path that supports SkinnyFuse)
bool gemm_nt_skinny_can_fuse(int M, int N, int K, size_t scratch_bytes);
}  // namespace CC_NS
