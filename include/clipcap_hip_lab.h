/* LAB BUILD ONLY — entry points of libclipcap_hip_lab.so (`make -C clipcap_amd/csrc lab`, -DCC_EXPERIMENTS) on top of include/clipcap_hip.h.
 *
 * Everything here was built, tested and measured SLOWER than the product path on MI355X (HISTORY.md has the numbers); it stays buildable as
 * the A/B arm of those measurements (tests/lab_*.py, run by tests/test_gpu_lab.py under the `lab` marker; tools/ab_decode_mode.py).  The
 * product library neither exports nor contains any of it. */
#ifndef CLIPCAP_HIP_LAB_H
#define CLIPCAP_HIP_LAB_H
#include "clipcap_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Single-position group steps (cc_decode_fwd_g, Tnew == 1, bf16 / fp16 operands, head dim 64, R <= 512) run the whole layer stack as ONE
 * persistent launch whose workgroups hand activations to each other through arrival counters in `ws`; every wait in it is bounded, and a
 * wait that gives up raises an error word in `ws`.  cc_decode_ws_check synchronises `stream` and returns CC_ERR_STATE if the last step on
 * this workspace gave up (its logits are then garbage), CC_OK otherwise.  A debugging / test aid: a correct run never trips it. */
int cc_decode_ws_check(const cc_gpt2_cfg* cfg, int32_t R, int32_t Tnew, const void* ws, void* stream);
/* Decode weight images (bf16 / fp16 operands; rebuilt whenever the weights change, after cc_gpt2_sync_weights):
 *   cc_decode_image_bytes / cc_decode_image: the four GEMM weights of every GPT-2 block in MFMA FRAGMENT ORDER (per 64-column x 64-k tile the
 *       eight 1-KiB pieces its four waves feed to v_mfma_f32_32x32x16, lane-contiguous), at the offsets the weights have in the operand
 *       arena (the image is a permutation of each matrix: cc_gpt2_param_count elements).  With it the decode-sized GEMMs whose K is split
 *       over the waves (c_attn, mlp.c_proj at rows x beams = 320) load the weight operand global -> VGPR and keep it out of LDS entirely
 *       (gemm_nt_s64kwb_kernel); 0 bytes = operand type / width not covered.
 *   cc_decode_xt_image_bytes / cc_decode_xt_image: LAB BUILD ONLY (0 / CC_ERR_SHAPE in the product library) — the image of the XCD-team
 *       engine (decode_xt.hip: every XCD runs the whole layer stack of a single-position group step for its own rows, weights streamed into
 *       registers per (workgroup, wave) in consumption order; D = 512 or 1024, head dim 64, at most 48 rows per XCD, cc_decode_mode bit 2).
 *   cc_decode_fwd_x: cc_decode_fwd_g with the images (either may be NULL: exactly cc_decode_fwd_g then).  Results equal cc_decode_fwd_g's to
 *       fp32 summation order.  The engine's waits are bounded like the persistent launch's; cc_decode_ws_check reports a step that gave up
 *       (sticky until the workspace is zeroed). */
int64_t cc_decode_image_bytes(const cc_gpt2_cfg* cfg);
int cc_decode_image(const cc_gpt2_cfg* cfg, const uint16_t* w16, uint16_t* wimg, void* stream);
int64_t cc_decode_xt_image_bytes(const cc_gpt2_cfg* cfg);
int cc_decode_xt_image(const cc_gpt2_cfg* cfg, const uint16_t* w16, uint16_t* wteam, void* stream);
int cc_decode_fwd_x(const cc_gpt2_cfg* cfg, int32_t R, int32_t Tnew, int32_t pos0, int32_t ctx_max, const float* w32,
                    const uint16_t* w16, const uint16_t* wimg, const uint16_t* wteam, const float* x, uint16_t* kv, const int32_t* row_map,
                    int32_t group, void* ws, float* logits, int64_t ldl, float* lpart, void* stream);
/* cc_decode_mode bits beyond bit 0 (lab build): bit 1 = the whole layer stack as ONE persistent launch with in-launch hand-offs (decode_pk.hip;
 * bf16 / fp16 operands); bit 2 = cc_decode_fwd_x may use the XCD-team engine when it is handed that engine's image; bit 3 = the K-split decode
 * GEMMs read the weight operand global -> VGPR from the fragment-ordered image of cc_decode_image when one is passed (bit-identical results,
 * 1-3 % slower in the real chain).  env CC_DEC_GROUP / CC_DEC_PK / CC_DEC_XT preset the mode. */
/* which path the most recent single-position group step of cc_decode_fwd_x / cc_decode_fwd_g took: 0 = launch per op, 1 = persistent launch
 * (decode_pk.hip), 2 = XCD-team engine (decode_xt.hip).  PROCESS-WIDE test / measurement hook. */
int cc_decode_last_path(void);

#ifdef __cplusplus
}
#endif
#endif
