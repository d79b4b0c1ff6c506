// Host-side launchers of the non-GEMM kernels (definitions in kernels.hip / decode.hip).  Internal to the library.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include "common.hip.h"

namespace CC_NS {

int f32_to_bf16(const float* src, op16_t* dst, size_t n, hipStream_t st);      // 16-bit operand cast (weights)
int f32_to_act(const float* src, act_t* dst, size_t n, hipStream_t st);
int wire_pack(const float* src, unsigned short* dst, size_t n, hipStream_t st);        // fp32 -> bf16 gradient wire slice (any n / alignment)
int wire_unpack(const unsigned short* src, float* dst, size_t n, hipStream_t st);      // and back         // fp32 -> stored-activation type (a copy in the bf16x3 build)
int slice_f32_to_bf16(const float* src, size_t src_stride, act_t* dst, size_t dst_stride, int len, int B, hipStream_t st);
int broadcast_rows(float* dst, size_t dst_stride, const float* src, int len, int B, hipStream_t st);
int add_rows(float* dst, size_t dst_stride, const float* add, int len, int B, hipStream_t st);
int batch_sum(const float* src, size_t src_stride, float* dst, int len, int B, hipStream_t st);
int copy_rows(const float* src, size_t src_stride, float* dst, size_t dst_stride, int len, int B, hipStream_t st);

int transpose_bf16(const op16_t* src, op16_t* dst, int R, int C, hipStream_t st);
struct TransposeBatch {
    struct Item { const op16_t* src; op16_t* dst; int R, C; };
    Item it[32];
    int n = 0;
    void add(const op16_t* s, op16_t* d, int R, int C) { it[n++] = Item{s, d, R, C}; }
};
int transpose_bf16_multi(const TransposeBatch& b, hipStream_t st);   // up to 32 matrices in one launch

int ln_fwd(const float* x, int ldx, const int* row_map, const float* gamma, const float* beta, act_t* y, float* y32, float* mean,
           float* rstd, int rows, int D, hipStream_t st);
int ln_bwd(const act_t* dy, const float* x, int ldx, const int* row_map, const float* mean, const float* rstd, const float* gamma,
           const float* dres, float* dx32, act_t* dx16, float* dgamma, float* dbeta, int rows, int D, hipStream_t st,
           float* dcol = nullptr, Drop dmask = Drop());
// dcol (needs dgamma, dx16, no row_map): += column sums of the 16-bit dx16 (a bias gradient).  dmask: dropout mask applied to the
// 16-bit copy dx16 only (element index row * D + col; dx32 stays unmasked) — the residual dropout of the c_proj that consumes dx16.
int colsum_bf16(const act_t* X, int ld, int M, int N, float* out, hipStream_t st);
struct ColsumBatch { const act_t* X[32]; float* out[32]; int n = 0; void add(const act_t* x, float* o) { X[n] = x; out[n] = o; n++; } };
int colsum_bf16_multi(const ColsumBatch& b, int ld, int M, int N, hipStream_t st);      // out[i][c] += sum_r X[i][r][c] for n equally shaped matrices, one launch

int attn_probs(const act_t* qkv, int B, int S, int H, int hd, float* out, hipStream_t st);
int attn_fwd(const act_t* qkv, int B, int S, int H, int hd, bool causal, act_t* out, float* lse, hipStream_t st, Drop drop = Drop());
// o: forward output (for delta = rowsum(dO*O)); delta: fp32 scratch [B*H*S].  Both may be null -> VALU kernel.
int attn_bwd(const act_t* qkv, const act_t* dout, const act_t* o, const float* lse, float* delta, int B, int S, int H, int hd, bool causal,
             act_t* dqkv, hipStream_t st, Drop drop = Drop());
// in-place dropout of an fp32 / bf16 buffer of n elements (n % 4 == 0 / n % 8 == 0): x[i] *= keep(i) / (1 - p)
bool attn_fwd_can_image(int S, int hd);      // bf16x3: attn_fwd will honour x3_emit_image(out) (and the backward will not need the fp32 output)
bool attn_bwd_can_image(int S, int hd);      // bf16x3: attn_bwd will honour x3_emit_image(dqkv) for this shape
int dropout_f32(float* x, size_t n, Drop drop, hipStream_t st);
int dropout_bf16(act_t* x, size_t n, Drop drop, hipStream_t st);
int dropout_mask_u8(unsigned char* out, size_t n, Drop drop, hipStream_t st);   // test hook: out[i] = keep(i)

int embed_concat(const float* prefix, const long long* tokens, int cap, const float* wte, const float* wpe, float* x0, int B, int L,
                 int T, int D, int pos0, hipStream_t st);
int f32_to_op16_pad(const float* src, long long lds, int V, act_t* dst, int ldd, int M, hipStream_t st);
int embed_bwd(const float* dx0, const long long* tokens, int cap, float* dwte, float* dwpe, int B, int L, int T, int D, hipStream_t st);

int ce_rows(const float* pmax, const float* psum, int npart, const int* target, const float* tgt_logit, float* lse, float* row_loss,
            float* stats, int M, hipStream_t st);
int ce_dlogits(act_t* logits, int ld, int V, const int* target, const float* lse, const float* denom, const float* loss_scale, int M,
               hipStream_t st, op16_t* img = nullptr);   // img (bf16x3): write the gradient as the dgrad GEMM's operand image there instead of in place
// sample.hip: one sampling step per row (temperature, repetition penalty, top-k, top-p, inverse-CDF draw at u[row])
int sample_rows(const float* logits, int R, int V, int ld, float temperature, int top_k, float top_p, int mode, const long long* hist,
                int hist_len, int hist_ld, float rep_pen, const float* u, int* next_token, float* probs_out, hipStream_t st, int stop_tok = -1,
                float len_pen = 1.0f);   // stop_tok >= 0: sentence-length penalty (history tokens whose filtered value == stop id are scaled by len_pen)
int ce_targets(const long long* tokens, int* target, int* row_map, int B, int cap, int L, int T, hipStream_t st);
// Exponential form of the lm_head outputs (gemm.hip.h EpiLMHead, bf16 build):
//   lm_tgt_ref   cref[m] = hf[m] . wte[target[m]]  (16-bit operands, fp32 accumulate): the reference shift of row m
//   lm_rowfac    fac[m] = {r, w}: w = (target[m] != 0) * loss_scale / max(denom, 1), r = exp(cref[m] - lse[m]) * w
//   lm_dgrad_fix dhf[m][:] = r dhf[m][:] - w wte[target[m]][:]  (the path whose GEMM has no finishing pass of its own)
//   lm_scale_rows out[m][:] = r hf[m][:] (weight-gradient operand);  lm_wgrad_onehot dwte[target[m]][:] -= w hf[m][:]
int lm_tgt_ref(const act_t* hf, const op16_t* wte, int D, const int* target, float* cref, int M, hipStream_t st);
int lm_rowfac(const float* cref, const float* lse, const int* target, const float* denom, const float* loss_scale, float* fac, int M, hipStream_t st);
int lm_dgrad_fix(act_t* dhf, const float* fac, const int* target, const op16_t* wte, int D, int M, hipStream_t st);
int lm_scale_rows(const act_t* hf, const float* fac, act_t* out, int D, int M, hipStream_t st);
int lm_wgrad_onehot(const act_t* hf, const float* fac, const int* target, float* dwte, int D, int M, hipStream_t st);

int adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd, int step, float gscale,
          const float* loss_scale, const float* found_inf, hipStream_t st, op16_t* w16 = nullptr);   // w16: also store the 16-bit copy of the updated parameters
int grad_nonfinite(const float* g, size_t n, float* found_inf, hipStream_t st);

// bf16x3 build: GEMM operand pairs (common.hip.h).  dst[r][0:3K] = form 0 (A operand): [hi | hi | lo], form 1 (B operand): [hi | lo | hi]
// of src[r][0:K] (fp32, row stride lds); hi = bf16(x), lo = bf16(x - hi).  K % 8 == 0.
int x3_split_rows(const float* src, size_t lds, op16_t* dst, int M, int K, int form, hipStream_t st);
// several weight matrices in one launch: item = fp32 source [R][C]; tr = 0: dst[R][3C] rows of the source, tr = 1: dst[C][3R] rows of
// its transpose.  R, C % 8 == 0.
struct X3SplitBatch {
    struct Item { const float* src; op16_t* dst; int R, C, tr, form; };
    Item it[32];
    int n = 0;
    void add(const float* s, op16_t* d, int R, int C, int tr, int form) { it[n++] = Item{s, d, R, C, tr, form}; }
};
int x3_split_multi(const X3SplitBatch& b, hipStream_t st);
int loss_scale_update(float* state, float* found_inf, float growth, float backoff, int interval, hipStream_t st);

}  // namespace CC_NS
