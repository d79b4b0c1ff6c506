// Persistent decode-layer launch (decode_pk.hip): argument blocks shared with decode.hip.
#pragma once
#include "common.hip.h"

namespace CC_NS {
constexpr int PK_MAX_RT = 8;             // 64-row tiles: M <= 512 rows
constexpr int PK_CTR_WORDS = 7 * PK_MAX_RT + 8;   // arrival counters [phase][row tile] + the error word (zeroed before every launch)

struct PkArgs {
    const float* w32;
    const op16_t* w16t;
    int D, H, NL;
    long long layer0, layer_stride;      // element offset of layer 0's ln_1.weight in the parameter arena; elements per layer
    int M, NG, nrt, pos0, ctx_max;
    float scale;
    float *x, *x1;
    act_t *xn, *qkv, *att, *hact, *hf;
    float* slab;
    act_t* kv;
    size_t cache_layer;                  // elements of one layer's K + V cache
    const int2* ent;
    const int* cnt;
    int cap;
    unsigned* ctr;
    unsigned long long* prof;            // optional [workgroups][7 phases][poll time, total time, tasks] in 10 ns units (s_memrealtime); NULL = off
    int ks3, ks5, ks3_eff, ks5_eff;
    int n2[PK_MAX_RT], nfin_rt[PK_MAX_RT];
    int attn_floats;
};

struct PkLaunch {
    const float* w32;
    const op16_t* w16t;
    int D, H, NL, M, group, pos0, ctx_max;
    long long layer0;
    float *x, *x1;
    act_t *xn, *qkv, *att, *hact, *hf;
    float* slab;
    size_t slab_bytes;
    act_t* kv;
    size_t cache_layer;
    const int2* ent;
    const int* cnt;
    int cap;
    unsigned* ctr;
    unsigned long long* prof;
};
// CC_OK, or CC_ERR_SHAPE when the geometry / build is not covered (the caller then takes the launch-per-op path)
int decode_layers_persistent(const PkLaunch& L, hipStream_t st);
}  // namespace CC_NS
