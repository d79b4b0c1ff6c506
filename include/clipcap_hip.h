/* clipcap_hip.h — C ABI of libclipcap_hip.so: the MI355X (gfx950) ClipCap hot path.
 *
 * The reference (TheoCoombes/ClipCap) is 100 % Python and has NO plugin / FFI / operator interface for this path
 * (SURVEY.md §8b): the effective boundary is the nn.Module surface of clipcap/model/{mapper,attention,model}.py and
 * the HF GPT-2 it instantiates.  Each entry point below therefore cites the reference *Python call* it replaces;
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host".
 *   - the caller owns every buffer (parameters, gradients, optimizer state, workspace, KV cache).  The compute entry
 *     points keep no state and never allocate or synchronise; all work is enqueued on the caller's hipStream_t (passed
 *     as void*; NULL = default stream), so calls on different streams / from different host threads are independent.
 *     Per-step settings (GPT-2 dropout, loss scale) travel in the call's own arguments.  The only process-wide state is
 *     the test / measurement hooks at the end of this file (cc_gemm_tile_mode, cc_gemm_skinny_mode, cc_decode_mode,
 *     cc_prof_*), which the product path never touches.  libclipcap_hip.so does not read the environment
 *     (getenv is not among its imports) and exports exactly the functions declared here (csrc/exports.map).  The kernels and A/B
 *     code paths that were built, measured and lost — the persistent decode-layer launch, the XCD-team decode engine, decode GEMMs on
 *     fragment-ordered weight images, fp32-MFMA attention — their entry points (include/clipcap_hip_lab.h) and the CC_* environment
 *     switches that select them live in the LAB build only (`make -C clipcap_amd/csrc lab` -> libclipcap_hip_lab.so, -DCC_EXPERIMENTS,
 *     csrc/lab_env.h; CLIPCAP_HIP_LIB=lab makes the Python binding load it).
 *   - return value: 0 = ok, <0 = error (CC_ERR_*), never throws across the ABI.
 *   - OPERAND TYPE.  GEMM / attention operands and the stored 16-bit activations are bf16 (CC_OP_BF16, default) or IEEE
 *     fp16 (CC_OP_FP16 = the reference's `--fp-precision 16`, clipcap/train/args.py:30-34), selected per model by
 *     cfg->op_dtype; accumulation, master weights, residual streams, LayerNorm statistics, the loss and the optimizer are
 *     fp32 in both.  16-bit tensors cross the ABI as raw bit patterns (uint16_t), round-to-nearest-even like
 *     torch.bfloat16 / torch.float16.  fp16 training runs its backward pass under a loss scale (cc_lmhead_ce_bwd,
 *     cc_grad_nonfinite, cc_loss_scale_update, cc_adamw_step) exactly as torch.cuda.amp.GradScaler does for the
 *     reference's Lightning fp16 path.
 *     CC_OP_BF16X3 is the reference's DEFAULT precision (`--fp-precision 32`, clipcap/train/args.py:30-34, train.py:82) on a chip
 *     whose fp32 MFMA runs at 1/16 of its bf16 rate (three bf16 terms are ~5x faster than v_mfma_f32_32x32x2_f32): every GEMM operand x is split into bf16 hi = bf16(x), lo = bf16(x - hi) and each product runs as
 *     three bf16 MFMA terms hi*hi + hi*lo + lo*hi with fp32 accumulation (about 16 mantissa bits per operand, a third of the bf16
 *     rate); activations between kernels are fp32 and attention runs in fp32.  This is the mode in which logits match the fp32
 *     reference to 1e-3 at full depth.  Its buffers differ in size only: the operand arena has 6*count 16-bit elements (the
 *     [hi | lo | hi] image of every 2-D weight at 3x its offset, of its transpose at 3*(count + offset)), the KV cache holds fp32
 *     (twice the bytes per element), workspaces are whatever cc_*_ws_bytes says.  cc_*_transpose_weights, cc_adamw_step_cast,
 *     cc_cast_op16 and the bare GEMM hooks do not exist in this mode (CC_ERR_ARG): use cc_adamw_step + cc_*_sync_weights.
 *   - parameters live in flat arenas whose element offsets are defined by cc_*_param_offsets(); the fp32 arena is
 *     the master copy (nn.Parameter views alias it).  The 16-bit operand arena has 2*count elements: [0,count) is the
 *     cast of the master (same offsets, reference state-dict layouts: torch.nn.Linear weight [out,in]; HF Conv1D
 *     weight [in,out]); [count, 2*count) holds, at the same offsets, the TRANSPOSE of every 2-D GEMM weight, so that
 *     forward and dgrad GEMMs are both "NT" (both operands K-contiguous, the direct-to-LDS fast path).  It is
 *     refreshed from the master by cc_mapper_sync_weights / cc_gpt2_sync_weights after every optimizer step.
 */
#ifndef CLIPCAP_HIP_H
#define CLIPCAP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CC_OK 0
#define CC_ERR_ARG (-1)
#define CC_ERR_SHAPE (-2)
#define CC_ERR_LAUNCH (-3)
#define CC_ERR_STATE (-4)

/* bumped when an existing entry point changes.  3: cc_loss_scale_update's state is float[3] (applied-step counter in state[2]) and
 * cc_adamw_step / cc_adamw_step_cast accept step == 0 (= take the step number from loss_scale[2]) — a version-2 caller passing float[2]
 * would be written out of bounds, so clipcap_amd/_lib.py refuses a library whose version differs.  Entry points ADDED since 2:
 * cc_adamw_step_cast, cc_mapper_transpose_weights, cc_gpt2_transpose_weights, cc_comm_count, cc_decode_part_floats, cc_decode_fwd_p,
 * cc_beam_step_p; since 3: cc_decode_fwd_g, cc_decode_ws_check, cc_decode_mode, cc_grad_wire_pack, cc_grad_wire_unpack,
 * cc_sample_step_lp, cc_broadcast_bucket, cc_reduce_bucket, cc_embed_tokens_bwd; operand mode ADDED: CC_OP_BF16X3.  Round 6 (still 3): the default-off decode experiments
 * (cc_decode_image*, cc_decode_xt_image*, cc_decode_fwd_x, cc_decode_ws_check, cc_decode_last_path) moved to include/clipcap_hip_lab.h —
 * the lab library exports them, the product library does not. */
#define CC_ABI_VERSION 3
int cc_abi_version(void);

#define CC_OP_BF16 0
#define CC_OP_FP16 1
#define CC_OP_BF16X3 2   /* split operands: the reference's default --fp-precision 32 (see OPERAND TYPE above) */

/* ------------------------------------------------------------------------------------------------------------
 * Mapper: clipcap/model/mapper.py:113-130 TransformerMapper (+ :133-160 windowed), layers :91-110, MLP :70-88,
 * attention clipcap/model/attention.py:4-43.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t E;       /* encoder_embedding_size */
    int32_t D;       /* lm_embedding_size */
    int32_t P;       /* projection_length */
    int32_t L;       /* prefix_length */
    int32_t H;       /* num_heads */
    int32_t N;       /* num_layers */
    int32_t Hm;      /* MLP hidden = int(D * 2.0) (mapper.py:10,100) */
    int32_t W;       /* window count (1 = TransformerMapper; window_size+1 for TransformerMapperWindowed, model.py:28) */
    int32_t use_pos; /* windowed: learned pos_embeddings present (mapper.py:142-145) */
    int32_t op_dtype; /* CC_OP_BF16 / CC_OP_FP16: type of the w16 arena and of every 16-bit activation of this model */
} cc_mapper_cfg;

/* number of per-model tensors ahead of the layers (linear.weight, linear.bias, prefix_const, pos_embeddings) */
#define CC_MAPPER_HEAD_TENSORS 4
/* per layer, in arena order: norm1.weight norm1.bias attn.to_queries.weight attn.to_keys_values.weight
 * attn.project.weight attn.project.bias norm2.weight norm2.bias mlp.fc1.weight mlp.fc1.bias mlp.fc2.weight mlp.fc2.bias
 * (to_queries.weight [D,D] is immediately followed by to_keys_values.weight [2D,D]: one fused [3D,D] QKV operand) */
#define CC_MAPPER_LAYER_TENSORS 12

int64_t cc_mapper_param_count(const cc_mapper_cfg* cfg);
/* offsets[CC_MAPPER_HEAD_TENSORS + N*CC_MAPPER_LAYER_TENSORS] element offsets into the arenas (host array);
 * pos_embeddings offset is -1 when absent. */
int cc_mapper_param_offsets(const cc_mapper_cfg* cfg, int64_t* offsets);
/* workspace bytes for batch B; save=1 keeps every layer's activations for cc_mapper_bwd */
int64_t cc_mapper_ws_bytes(const cc_mapper_cfg* cfg, int32_t B, int32_t save);

/* w16[0:count] = bf16(w32); w16[count:2*count] = per-tensor transposes of the GEMM weights (see Conventions) */
int cc_mapper_sync_weights(const cc_mapper_cfg* cfg, const float* w32, uint16_t* w16, void* stream);
/* only the second half: w16[count:2*count] from w16[0:count] — for callers whose optimizer already wrote the cast (cc_adamw_step_cast) */
int cc_mapper_transpose_weights(const cc_mapper_cfg* cfg, uint16_t* w16, void* stream);

/* replaces model.transformer_mapper(embeddings) (mapper.py:122-130; callers model.py:46, inference/generate.py:31,
 * docs/inference.md:24).  emb fp32 [B, W, E]; out fp32 [B, L, D]. */
int cc_mapper_fwd(const cc_mapper_cfg* cfg, int32_t B, const float* w32, const uint16_t* w16, const float* emb, void* ws,
                  float* out, int32_t save, void* stream);
/* the second return value of MultiHeadAttention.forward (attention.py:32-42; Transformer.forward_with_attention, mapper.py:45-52):
 * softmax attention probabilities of one layer, fp32 [B, S, S, H] (query n, key m, head h), recomputed from the activations a
 * save=1 forward left in ws.  Inspection output; not used by training or decode. */
int cc_mapper_attention_probs(const cc_mapper_cfg* cfg, int32_t B, void* ws, int32_t layer, float* out, void* stream);
/* autograd of the above (what loss.backward() does to mapper.py in the reference).  dout fp32 [B, L, D];
 * g32 (flat, same offsets as w32) is ACCUMULATED into.  Requires the workspace of a save=1 forward. */
int cc_mapper_bwd(const cc_mapper_cfg* cfg, int32_t B, const float* w32, const uint16_t* w16, void* ws, const float* dout,
                  float* g32, void* stream);
/* the same, one slice of layers [l_lo, l_hi) at a time, called with descending ranges (first l_hi == N seeds from dout, the call
 * with l_lo == 0 also produces the linear / prefix_const / pos_embeddings gradients).  Lets the caller start the RCCL
 * all-reduce of a layer's gradient slice while the layers below it are still in backward. */
int cc_mapper_bwd_range(const cc_mapper_cfg* cfg, int32_t B, const float* w32, const uint16_t* w16, void* ws, const float* dout,
                        float* g32, int32_t l_hi, int32_t l_lo, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * GPT-2: transformers GPT2LMHeadModel as called by model.py:56 / inference/base.py:81 with inputs_embeds
 * (hf modeling_gpt2.py:514-634, blocks :262-310, attention :54-72,:144-226, MLP :229-243, lm_head :637-725).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t D;     /* n_embd */
    int32_t H;     /* n_head */
    int32_t NL;    /* n_layer */
    int32_t V;     /* vocab_size */
    int32_t Vp;    /* vocab rows of the arenas' wte: V rounded up to a multiple of 128 (zero rows) */
    int32_t NPOS;  /* n_positions */
    int32_t op_dtype; /* CC_OP_BF16 / CC_OP_FP16 (see Conventions) */
} cc_gpt2_cfg;

#define CC_GPT2_HEAD_TENSORS 2   /* wte [Vp,D], wpe [NPOS,D] */
/* per layer: ln_1.weight ln_1.bias attn.c_attn.weight[D,3D] attn.c_attn.bias attn.c_proj.weight[D,D] attn.c_proj.bias
 * ln_2.weight ln_2.bias mlp.c_fc.weight[D,4D] mlp.c_fc.bias mlp.c_proj.weight[4D,D] mlp.c_proj.bias */
#define CC_GPT2_LAYER_TENSORS 12
#define CC_GPT2_TAIL_TENSORS 2   /* ln_f.weight ln_f.bias */

/* one training / scoring pass: every GPT-2 entry point of a pass must be given the SAME shape (it fixes the
 * workspace layout).  T = L + cap_used (cap_used <= cap columns of `tokens` are consumed: T - L). */
typedef struct {
    int32_t B;     /* samples */
    int32_t L;     /* prefix rows per sample */
    int32_t T;     /* total rows per sample (prefix + caption tokens) */
    int32_t cap;   /* row stride of `tokens` (int64 [B, cap]); the loss uses T-L = cap columns */
    int32_t mode;  /* 0 inference (no activations kept); 1 training, frozen LM (dgrad only); 2 full finetune (+wgrad) */
    /* GPT-2 dropout of THIS pass (full finetune in train mode: the reference's ClipCapModel leaves the HF GPT-2 in train mode,
     * model.py:19; hf modeling_gpt2.py: embd_pdrop on inputs+positions, attn_pdrop on the attention probabilities, resid_pdrop
     * after both c_proj).  Masks are a counter-based hash of (drop_seed, site, layer, element) regenerated by the backward
     * kernels — nothing is stored; cc_gpt2_embed / cc_gpt2_fwd / cc_gpt2_bwd(_range) of one pass must be given the same values.
     * All-zero probabilities = eval behaviour (ClipCapModelPrefixOnly keeps GPT-2 in eval, model.py:120-123). */
    float p_embd, p_attn, p_resid;
    uint64_t drop_seed;
} cc_gpt2_shape;

int64_t cc_gpt2_param_count(const cc_gpt2_cfg* cfg);
int cc_gpt2_param_offsets(const cc_gpt2_cfg* cfg, int64_t* offsets);
int64_t cc_gpt2_ws_bytes(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* shp);

int cc_gpt2_sync_weights(const cc_gpt2_cfg* cfg, const float* w32, uint16_t* w16, void* stream);
int cc_gpt2_transpose_weights(const cc_gpt2_cfg* cfg, uint16_t* w16, void* stream);   /* as cc_mapper_transpose_weights */

/* x0[b,t,:] = (t<L ? prefix[b,t,:] : wte[max(tok[b,t-L],0),:]) + wpe[t,:]  — model.py:45-49 + hf :571-577.
 * prefix fp32 [B,L,D]; tokens int64 [B, cap] (may be NULL when L == T).  Writes the workspace's layer-0 input. */
int cc_gpt2_embed(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* shp, const float* w32, const float* prefix, const int64_t* tokens,
                  void* ws, void* stream);
/* same, but from a caller-provided inputs_embeds fp32 [B,T,D] (the `language_model(inputs_embeds=...)` call of
 * inference/base.py:81): x0 = inputs_embeds + wpe[arange(T)] */
int cc_gpt2_embed_from(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* shp, const float* w32, const float* inputs_embeds, void* ws,
                       void* stream);
/* transformer body (all blocks; ln_f is applied by the lm_head entry points). */
int cc_gpt2_fwd(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* shp, const float* w32, const uint16_t* w16, void* ws, void* stream);
/* ln_f + tied lm_head on ALL rows -> fp32 logits [B*T, ldl] (what `.logits` holds; hf :703); ldl >= roundup(V,8). */
int cc_gpt2_logits(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* shp, const float* w32, const uint16_t* w16, void* ws, float* logits,
                   int64_t ldl, void* stream);

/* autograd of cc_gpt2_logits for callers that differentiate `.logits` themselves (ClipCapModel.forward(...).logits.backward(), a
 * custom loss): needs a pass run with shape L == 0 (so cap == T) and mode >= 1 — cc_gpt2_embed_from, cc_gpt2_fwd, cc_gpt2_logits —
 * then this.  dlogits fp32 [B*T, ldl] (first V columns read); dx0 fp32 [B, T, D] (nullable) receives d loss / d inputs_embeds;
 * mode 2 accumulates every GPT-2 weight gradient (tied wte through lm_head, wpe, blocks, ln_f) into g32. */
int cc_gpt2_logits_bwd(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* shp, const float* w32, const uint16_t* w16, void* ws,
                       const float* dlogits, int64_t ldl, float* dx0, float* g32, void* stream);

/* training loss of model.py:94-113 on rows L-1..T-2: fused ln_f + lm_head + softmax cross-entropy (ignore_index=0,
 * pads(-1)->0).  stats (2 device floats, zeroed by this call): [0] sum of kept-row losses, [1] number of kept rows. */
int cc_lmhead_ce_fwd(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* shp, const float* w32, const uint16_t* w16, void* ws,
                     const int64_t* tokens, float* stats, void* stream);
/* backward of the above through lm_head and ln_f into the residual-stream gradient kept in the workspace.
 * denom (device float[1]) = divisor of the mean (local or all-reduced kept-row count).  loss_scale (device float[1], NULL = 1):
 * every gradient of this backward pass (dprefix and what is accumulated into the g32 arenas) is multiplied by it — fp16 operands
 * need it to keep d logits = (softmax - onehot) / denom above the fp16 underflow threshold; cc_adamw_step divides it out again.
 * g32 may be NULL unless mode 2. */
int cc_lmhead_ce_bwd(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* shp, const float* w32, const uint16_t* w16, void* ws,
                     const float* denom, const float* loss_scale, float* g32, void* stream);
/* backward through the blocks.  dprefix fp32 [B, L, D] receives d loss / d prefix (rows 0..L-1 of d x0).
 * mode 2 additionally accumulates all GPT-2 weight gradients (incl. wte/wpe) into g32. */
int cc_gpt2_bwd(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* shp, const float* w32, const uint16_t* w16, void* ws,
                const int64_t* tokens, float* dprefix, float* g32, void* stream);
/* blocks [l_lo, l_hi) only, descending ranges; the call with l_lo == 0 also writes dprefix (and wte/wpe gradients in mode 2) */
int cc_gpt2_bwd_range(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* shp, const float* w32, const uint16_t* w16, void* ws,
                      const int64_t* tokens, float* dprefix, float* g32, int32_t l_hi, int32_t l_lo, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * KV-cached decode (replaces the full re-forward of inference/base.py:81 per step).
 * kv cache: bf16 [NL][2][R][ctx_max][D] owned by the caller (R rows = beams * samples).
 * ------------------------------------------------------------------------------------------------------------ */
int64_t cc_decode_ws_bytes(const cc_gpt2_cfg* cfg, int32_t R, int32_t Tnew);
/* processes Tnew new positions per row (prefill: Tnew = prefix length; step: Tnew = 1) starting at position pos0,
 * appends K/V to the cache (row r writes cache row r) and returns fp32 logits of the LAST new position [R, ldl].
 * x fp32 [R,Tnew,D] WITHOUT wpe.  row_map (int32 [R][ctx_max], may be NULL = identity): cache row that holds position j of
 * row r — a beam reorder (base.py:93,113) then only permutes this small table instead of copying the cache. */
int cc_decode_fwd(const cc_gpt2_cfg* cfg, int32_t R, int32_t Tnew, int32_t pos0, int32_t ctx_max, const float* w32,
                  const uint16_t* w16, const float* x, uint16_t* kv, const int32_t* row_map, void* ws, float* logits, int64_t ldl,
                  void* stream);
/* cc_decode_fwd that also hands out, from the lm_head GEMM's epilogue, the softmax partials of every logits row: lpart (device, may be
 * NULL = plain cc_decode_fwd) = pmax [R][npart] followed by psum [R][npart], npart = ceil(min(Vp, round_up(V, 8)) / 64); pmax = maximum of
 * the row's logits in the 64-column block, psum = sum of exp(logit - pmax) over it (columns >= V excluded).  cc_decode_part_floats = the
 * float count of lpart (2 * R * npart).  cc_beam_step_p uses them to run the whole beam update in one launch without a pass over the
 * logits matrix. */
int64_t cc_decode_part_floats(const cc_gpt2_cfg* cfg, int32_t R);
int cc_decode_fwd_p(const cc_gpt2_cfg* cfg, int32_t R, int32_t Tnew, int32_t pos0, int32_t ctx_max, const float* w32,
                    const uint16_t* w16, const float* x, uint16_t* kv, const int32_t* row_map, void* ws, float* logits, int64_t ldl,
                    float* lpart, void* stream);
/* cc_decode_fwd_p for rows that come in groups of `group` consecutive rows sharing ancestry (beam search: group = beam width, the rows of
 * one caption; R % group == 0, 1 <= group; 1 = cc_decode_fwd_p).  Results are those of cc_decode_fwd_p for ANY row_map; the hint lets the
 * single-position attention step (Tnew == 1, group <= 8) read every distinct (cache row, position) of a group once for all its rows
 * instead of once per row (inference/base.py:80-121 re-forwards every beam separately). */
int cc_decode_fwd_g(const cc_gpt2_cfg* cfg, int32_t R, int32_t Tnew, int32_t pos0, int32_t ctx_max, const float* w32,
                    const uint16_t* w16, const float* x, uint16_t* kv, const int32_t* row_map, int32_t group, void* ws, float* logits,
                    int64_t ldl, float* lpart, void* stream);
/* reorder / expand cache rows after a beam step: kv_dst[:, :, r] = kv_src[:, :, src[r]] for positions < ctx and
 * r < R_dst (base.py:93,113: embeds.expand / embeds[next_tokens_source]); the two caches may have different row counts. */
int cc_decode_reorder(const cc_gpt2_cfg* cfg, int32_t R_src, int32_t R_dst, int32_t ctx, int32_t ctx_max, const uint16_t* kv_src,
                      uint16_t* kv_dst, const int32_t* src, void* stream);
/* one beam-search update for S independent samples of `beam` rows each (base.py:84-119): log-softmax of logits/T,
 * stopped rows -> -inf except col 0 -> 0, length-normalised top-`beam` over beam*V, outputs next token / source row /
 * updated scores, seq_lengths, has_stopped.  first!=0: step 0 (top-beam of the single row, base.py:86-94). */
int cc_beam_step(int32_t S, int32_t beam, int32_t V, const float* logits, int64_t ldl, float temperature, int32_t first,
                 int32_t stop_token, float* scores, float* seq_lengths, uint8_t* has_stopped, int32_t* next_tokens,
                 int32_t* src_rows, void* ws, void* stream);
/* the same update given the logits' partials (lpart / npart as cc_decode_fwd_p writes them for the S * beam rows; NULL = cc_beam_step):
 * row statistics from the partials, every 64-column block bounded by its maximum, only the handful of blocks that can hold a winner
 * read — one launch.  Identical results (arithmetic, tie rule); temperature != 1 or widths other than 1..5, 8 take cc_beam_step's path. */
int cc_beam_step_p(int32_t S, int32_t beam, int32_t V, const float* logits, int64_t ldl, const float* lpart, int32_t npart,
                   float temperature, int32_t first, int32_t stop_token, float* scores, float* seq_lengths, uint8_t* has_stopped,
                   int32_t* next_tokens, int32_t* src_rows, void* ws, void* stream);
int64_t cc_beam_ws_bytes(int32_t S, int32_t beam, int32_t V);
/* gathers wte rows for next tokens: out fp32 [R, D] (base.py:117) */
int cc_embed_tokens(const cc_gpt2_cfg* cfg, int32_t R, const float* w32, const int32_t* tokens, float* out, void* stream);
/* its gradient: dwte fp32 [Vp, D] += scatter of dout fp32 [R, D] by tokens (rows sharing an id accumulate; fp32 atomics) — what autograd runs
 * for `language_model.get_input_embeddings()(tokens)` in a full finetune driven through Module.forward (clipcap/model/model.py:44) */
int cc_embed_tokens_bwd(const cc_gpt2_cfg* cfg, int32_t R, const float* dout, const int32_t* tokens, float* dwte, void* stream);
/* Everything between two beam steps in one launch (base.py:104-117): row r continues row g = (r / beam) * beam + src_rows[r] of its beam
 * group (src_rows NULL: g = r).  x_out fp32 [R, D] = wte[next_tokens[r]] (w32 points at wte);  row_map_out[r][j] = row_map_in[g][j] for
 * j < pos, r for the positions still to come (tables int32 [R][ctx_max], see cc_decode_fwd; NULL = skip);  tokens_out[r][:step] =
 * tokens_in[g][:step], tokens_out[r][step] = next_tokens[r] (int32 [R][tok_ld]; NULL = skip).  In / out buffers must differ. */
int cc_beam_advance(const cc_gpt2_cfg* cfg, int32_t R, int32_t beam, const float* w32, const int32_t* next_tokens, const int32_t* src_rows,
                    int32_t pos, int32_t ctx_max, const int32_t* row_map_in, int32_t* row_map_out, int32_t step, int32_t tok_ld,
                    const int32_t* tokens_in, int32_t* tokens_out, float* x_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimizer / casts: torch.optim.AdamW as configured by model.py:73-77 (lr schedule is computed by the caller,
 * model.py:79-83), flat over one arena.  The gradient used is g32 * grad_scale / (*loss_scale) (loss_scale: device float[1],
 * NULL = 1).  found_inf (device float[1], NULL = never): a non-zero value skips the update of every element — the step a
 * GradScaler drops when the scaled fp16 backward overflowed — and leaves m, v and the parameters untouched.
 * step >= 1 is Adam's step number (bias correction).  step == 0 (needs loss_scale): the number is taken from the device,
 * 1 + loss_scale[2], the loss scaler's count of steps actually applied (cc_loss_scale_update) — a skipped step must not advance
 * the bias correction, and the host cannot know which steps were skipped without a synchronisation.
 * ------------------------------------------------------------------------------------------------------------ */
int cc_adamw_step(float* p32, const float* g32, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, float grad_scale, const float* loss_scale, const float* found_inf, void* stream);
/* cc_adamw_step that also stores w16[0:n] = cast of the updated parameters to op_dtype (untouched when the step is skipped), so that
 * the refresh of the operand arena after a step is cc_*_transpose_weights alone: one pass over the arena less per step. */
int cc_adamw_step_cast(int32_t op_dtype, float* p32, const float* g32, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int32_t step, float grad_scale, const float* loss_scale, const float* found_inf,
                       uint16_t* w16, void* stream);
/* dst = cast of src to the 16-bit operand type op_dtype */
int cc_cast_op16(int32_t op_dtype, const float* src, uint16_t* dst, int64_t n, void* stream);
/* bf16 gradient wire of the N-rank all-reduce (train/train.py:77-85 runs DDP; clipcap_amd/train/ddp.py): wire[0:n] = bf16(g32[0:n]) (round to
 * nearest even) before the collective, g32[0:n] = fp32(wire[0:n]) after it.  Any n and any element alignment (a layer's gradient slice
 * starts wherever its first parameter does); always bf16, whatever the operand mode. */
int cc_grad_wire_pack(const float* g32, uint16_t* wire, int64_t n, void* stream);
int cc_grad_wire_unpack(const uint16_t* wire, float* g32, int64_t n, void* stream);
/* Dynamic loss scaling for CC_OP_FP16 training (torch.cuda.amp.GradScaler semantics; what Lightning wraps around the reference's
 * model when --fp-precision 16, clipcap/train/train.py:77-85), entirely on the device:
 *   cc_grad_nonfinite: *found_inf = 1 if any of the n gradients is inf / nan (never clears it; call once per arena, after the
 *                      all-reduce in a multi-GPU step so that every rank takes the same decision);
 *   cc_loss_scale_update: state (device float[3]): [0] = scale, [1] = consecutive good steps, [2] = optimizer steps applied so far.
 *                      found_inf != 0: scale *= backoff, counter = 0; else counter += 1 (at `interval`: scale *= growth, counter = 0)
 *                      and state[2] += 1.  Clears *found_inf. */
int cc_grad_nonfinite(const float* g32, int64_t n, float* found_inf, void* stream);
int cc_loss_scale_update(float* state, float* found_inf, float growth, float backoff, int32_t interval, void* stream);

/* test hook for the dropout of cc_gpt2_shape: keep flags of one mask stream — site 0 embd [B*T*D], 1 attention [B*H*T*T],
 * 2 residual after attn.c_proj [B*T*D], 3 residual after mlp.c_proj [B*T*D]. */
int cc_dropout_mask(uint64_t seed, int32_t site, int32_t layer, float p, int64_t n, uint8_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Data-parallel collective: SUM all-reduce of a gradient bucket over RCCL / xGMI, one communicator per process
 * (= per GPU).  What Lightning / DeepSpeed do implicitly around the reference's training_step (clipcap/train/train.py:77-85).
 * The communicator handle is the only object the library allocates; it carries no other state.
 *   rank 0: cc_comm_unique_id(uid) -> the caller ships the 128 bytes to the other ranks (MPI, a TCP store, torch.distributed ...)
 *   every rank, with its GPU current (hipSetDevice): cc_comm_create(&comm, nranks, rank, uid)
 *   per step: cc_allreduce_bucket(comm, g32 + lo, hi - lo, CC_RED_F32, stream) for each finished slice of the gradient arena
 *             (cc_*_bwd_range), enqueued on a side stream that waits on the backward stream — in place, asynchronous
 *   cc_comm_destroy(comm)
 * RCCL is bound at run time; CC_ERR_STATE = no librccl could be loaded.
 * ------------------------------------------------------------------------------------------------------------ */
#define CC_COMM_UID_BYTES 128
#define CC_RED_F32 0
#define CC_RED_BF16 1
#define CC_RED_F16 2
int cc_comm_unique_id(uint8_t* uid_host);
int cc_comm_create(void** comm, int32_t nranks, int32_t rank, const uint8_t* uid_host);
int cc_allreduce_bucket(void* comm, void* buf, int64_t count, int32_t dtype, void* stream);
/* In-place ncclBroadcast of buf[0, count) from rank `root`: with optimizer-state sharding (--deepspeed-strategy stage 1 / 2 of
 * clipcap/train/args.py:87-92) every rank runs cc_adamw_step on its own slice of the parameter arena and the owners' slices are
 * broadcast back, one call per owner. */
int cc_broadcast_bucket(void* comm, void* buf, int64_t count, int32_t dtype, int32_t root, void* stream);
/* In-place ncclReduce (SUM) of buf[0, count) onto rank `root`; the other ranks' buffers keep their own contribution.  Gradient
 * partitioning (--deepspeed-strategy stage 2 / 3, clipcap/train/args.py:87-92): a slice of the gradient arena is summed only where it is
 * consumed — on the rank that owns the matching AdamW moments — which moves half of an all-reduce's bytes over xGMI. */
int cc_reduce_bucket(void* comm, void* buf, int64_t count, int32_t dtype, int32_t root, void* stream);
/* ncclCommCount of the communicator: the number of ranks RCCL actually connected (what bench.py prints as rccl_ranks) */
int cc_comm_count(void* comm, int32_t* nranks);
int cc_comm_destroy(void* comm);

/* ------------------------------------------------------------------------------------------------------------
 * Sampling decoders (replaces the per-step torch ops of clipcap/inference/base.py:159-184 generate_nucleus_sampling and
 * :233-262 generate_no_beam + utils.py:5-37): one step for R rows of fp32 logits [R][ld].
 *   x = logits (history tokens first scaled by the repetition penalty, utils.py:33-37) / temperature (<= 0 -> 1, base.py:163)
 *   mode 0 (nucleus): p = softmax(x); the top_k largest (<= 0 or >= V: all); minimal descending prefix with cumulative p >= top_p
 *                     (mass relative to the FULL softmax, base.py:170-176); renormalised
 *   mode 1 (filter):  keep x >= the top_k-th largest (ties kept, utils.py:14-17), then the minimal descending prefix whose softmax
 *                     mass over that set is > top_p (utils.py:19-29; top_p <= 0: off); softmax of the rest
 *   next_token[r] = inverse CDF, in token-id order, of that distribution at u[r] in [0,1).  probs_out (nullable, [R][V]) receives
 *   the distribution itself.  history: int64 [R][hist_ld], first hist_len entries per row (nullable; penalty 1.0 = off).
 * No sort, no host sync; deterministic for given (logits, u).
 * ------------------------------------------------------------------------------------------------------------ */
int cc_sample_step(const float* logits, int32_t R, int32_t V, int32_t ld, float temperature, int32_t top_k, float top_p, int32_t mode,
                   const int64_t* history, int32_t hist_len, int32_t hist_ld, float repetition_penalty, const float* u, int32_t* next_token,
                   float* probs_out, void* stream);
/* The same step with generate_no_beam's sentence-length penalty (clipcap/inference/no_beam.py:55-60 -> utils.py:40-51): AFTER the
 * filter and before the softmax, a history token whose value EQUALS float(stop_token) is multiplied by length_penalty
 * (= current_length / desired_sentence_length * sentence_length_factor; the reference compares the gathered logit VALUES with the
 * token id, and so does this).  stop_token < 0 or an empty history: exactly cc_sample_step. */
int cc_sample_step_lp(const float* logits, int32_t R, int32_t V, int32_t ld, float temperature, int32_t top_k, float top_p, int32_t mode,
                      const int64_t* history, int32_t hist_len, int32_t hist_ld, float repetition_penalty, int32_t stop_token,
                      float length_penalty, const float* u, int32_t* next_token, float* probs_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Unit-test hooks (first argument op_dtype = CC_OP_BF16 / CC_OP_FP16: the type of the 16-bit tensors).
 * cc_gemm_op16_f32: one MFMA GEMM C = A·B (+bias) with fp32 output, any of the three operand layouts
 * (al/bl: 0 = [rows][K], 1 = [K][rows]); ksplit>1 accumulates atomically into C (caller zeroes C).
 * ------------------------------------------------------------------------------------------------------------ */
int cc_gemm_op16_f32(int32_t op_dtype, int32_t al, int32_t bl, const uint16_t* A, int32_t lda, const uint16_t* B, int32_t ldb, int32_t M, int32_t N,
                     int32_t K, float* C, int32_t ldc, const float* bias, int32_t ksplit, void* stream);
/* Weight-gradient GEMM as the backward passes run it: dW[Mw][Nw] += X^T Y with X stored [K][Mw], Y stored [K][Nw] (bf16), fp32
 * accumulation into dW; K is split into slices whose partial sums go through `scratch` (cc_wgrad_scratch_bytes() bytes, device). */
int64_t cc_wgrad_scratch_bytes(void);
int cc_gemm_wgrad(int32_t op_dtype, const uint16_t* X, int32_t ldx, const uint16_t* Y, int32_t ldy, int32_t Mw, int32_t Nw, int32_t K, float* dW, int32_t ldw,
                  float* scratch, void* stream);
/* PROCESS-WIDE test knob.  NT GEMM tile choice: -1 = cost-model chooser (default), 0 = 128x128 kernels only (also for cc_gemm_wgrad), 3 / 4 / 5 / 6 =
 * force the 256x192 / 256x256 / 320x256 / 160x256 (8-wave staggered) kernel wherever it is legal; 7 = 160x256 on the persistent 4-wave kernel with the
 * instruction-level K loop (what the chooser launches for that tile when K % 128 == 0 and K >= 256), the staggered one otherwise.  Returns the previous
 * mode.  For tests and tools/gemm_bench.py only. */
int cc_gemm_tile_mode(int32_t mode);
/* Decode-sized NT GEMMs (M <= 640 rows: cc_decode_fwd's c_attn / c_proj / c_fc at rows x beams = 320) run on 64-row tiles
 * (gemm_nt_s64_kernel).  -1 = default (those call sites only), 0 = never, 1 / 2 / 3 / 4 = additionally route cc_gemm_bf16_f32's NT launches with
 * M <= 1024 through the 64 x 64 / 64 x 128 / 64 x 64 K-split-over-waves (3) / 80 x 64 K-split-over-waves (4) form.  PROCESS-WIDE test knob; returns the previous mode.  For tests and
 * tools/small_gemm_bench.py. */
int cc_gemm_skinny_mode(int32_t mode);
/* How cc_decode_fwd_g runs a single-position group step.  bit 0 (default on): beam-group attention (every distinct KV row of a group read
 * once; off = one attention launch per row set, the A/B arm of tests/test_gpu_decode_group.py).  The other bits select lab-build experiments
 * (include/clipcap_hip_lab.h) and do nothing in the product library.  mode < 0 only queries.  PROCESS-WIDE test knob; returns the previous mode. */
int cc_decode_mode(int32_t mode);
int cc_layernorm_fwd(int32_t op_dtype, const float* x, const float* gamma, const float* beta, uint16_t* y, float* mean, float* rstd, int32_t rows,
                     int32_t D, void* stream);
int cc_attention_fwd(int32_t op_dtype, const uint16_t* qkv, int32_t B, int32_t S, int32_t H, int32_t hd, int32_t causal, uint16_t* out, float* lse,
                     void* stream);
/* o = forward output and delta_ws = fp32 scratch [B*H*S] select the MFMA kernels (hd 64/96/128); NULL -> LDS/VALU kernel */
int cc_attention_bwd(int32_t op_dtype, const uint16_t* qkv, const uint16_t* dout, const uint16_t* o, const float* lse, float* delta_ws, int32_t B, int32_t S,
                     int32_t H, int32_t hd, int32_t causal, uint16_t* dqkv, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Measurement aid for bench.py's roofline line: brackets every launch of ONE GEMM call site — or, with CC_SITE_ALL_GEMMS, every
 * GEMM host launch of the library — with HIP events recorded on the launch stream.  PROCESS-WIDE, off by default, never used by
 * the product path.
 * ------------------------------------------------------------------------------------------------------------ */
#define CC_SITE_LMHEAD_FWD 1      /* [B*cap, D] x wte^T           (cc_lmhead_ce_fwd) */
#define CC_SITE_LMHEAD_DGRAD 2    /* dlogits [B*cap, Vp] x wte    (cc_lmhead_ce_bwd) */
#define CC_SITE_GPT2_FC_FWD 3     /* c_fc  [B*T, D] x [D, 4D]     (cc_gpt2_fwd, per layer) */
#define CC_SITE_GPT2_PROJ2_FWD 4  /* mlp.c_proj [B*T, 4D] x [4D, D] (cc_gpt2_fwd) */
#define CC_SITE_GPT2_FC_DGRAD 5   /* d u-> d xn2  [B*T, 4D] x [4D(k), D] (cc_gpt2_bwd) */
#define CC_SITE_MAPPER_FC1_FWD 6  /* fc1 [B*S, D] x [Hm, D]^T     (cc_mapper_fwd) */
#define CC_SITE_MAPPER_QKV_FWD 7  /* fused q/kv projection         (cc_mapper_fwd) */
#define CC_SITE_MAPPER_WGRAD_FC2 8 /* dW2 = dx^T h (split-K)       (cc_mapper_bwd) */
#define CC_SITE_ALL_GEMMS 100      /* every MFMA GEMM launch (forward, dgrad, wgrad, lm_head, decode), with its 2*M*N*K */
int cc_prof_start(int32_t site, int32_t max_samples);
/* waits for the recorded events; writes up to *n (in: capacity, out: count) durations in milliseconds and (flops_host nullable) the
 * algorithmic FLOPs 2*M*N*K of each bracketed launch (host arrays) */
int cc_prof_stop(float* ms_host, double* flops_host, int32_t* n_host);

#ifdef __cplusplus
}
#endif
#endif /* CLIPCAP_HIP_H */
