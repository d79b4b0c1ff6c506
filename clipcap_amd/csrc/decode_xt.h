// XCD-team decode engine (decode_xt.hip): argument block shared with decode.hip.
#pragma once
#include "common.hip.h"

namespace CC_NS {
constexpr int XT_CTL_WORDS = 16 + 8 * 32;      // [0..7] team tickets, [8] arrivals, [9] error, [16 + 32 t] team t's barrier counter (128 B apart); zeroed before every launch
constexpr int XT_RING_MAX = 64;                // fragments a wave may request past the end of its stream (image padding)
constexpr int XT_PROF_WORDS = 32;              // per workgroup: [4 phase + k] s_memrealtime ticks (10 ns) of sub-interval k of phase 0..5: 0 = team-counter poll, 1 = panel fill, 2 = MFMA waves' K loop, 3 = epilogue + drain + arrive

struct XtLaunch {
    const float* w32;            // fp32 parameter arena (biases, LayerNorm parameters)
    const op16_t* wimg;          // fragment-ordered weight image (cc_decode_xt_image)
    int D, H, NL, M, group, pos0, ctx_max;
    long long layer0;            // element offset of layer 0's ln_1.weight in the arena
    float *x, *x1;
    act_t *qkv, *att, *hact, *hf;
    act_t* kv;
    size_t cache_layer;
    const int2* ent;
    const int* cnt;
    int cap;
    unsigned* ctl;               // XT_CTL_WORDS words, zeroed before the launch
    unsigned* sticky;            // one word, never cleared by the library: any launch that gave up sets it
    unsigned long long* prof;    // optional [256][XT_PROF_WORDS]
};
// frags per wave and layer / image bytes; 0 when the width is not covered
int64_t xt_image_bytes(int D, int NL);
int xt_build_image(int D, int NL, long long layer0, long long total, const op16_t* w16, op16_t* img, hipStream_t st);
// true when decode_layers_xt would launch for this geometry on this device (nothing is launched)
bool xt_covers(const XtLaunch& L);
// CC_OK, or CC_ERR_SHAPE when the geometry / build / device is not covered (the caller then takes the launch-per-op path)
int decode_layers_xt(const XtLaunch& L, hipStream_t st);
}  // namespace CC_NS
