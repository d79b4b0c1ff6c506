// Shared device helpers for the ClipCap gfx950 kernels (wave64, CDNA4 only — no portability shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lab_env.h"

// ---- operand type of this build of the kernels ------------------------------------------------------------------------------
// Every translation unit is compiled twice (Makefile): CC_OP = 0 with bf16 GEMM / attention operands (namespace cc_bf16, C symbols
// <name>_bf16) and CC_OP = 1 with IEEE fp16 operands (namespace cc_f16, <name>_f16) — the reference's `--fp-precision 16`
// (clipcap/train/args.py:30-34).  Same MFMA rate, same fp32 accumulation and fp32 master weights / residual streams; fp16 has 3 more
// mantissa bits (the route to the 1e-3 logits bar at GPT-2 width) and 3 fewer exponent bits (backward runs under a loss scale).
// abi_dispatch.cpp picks the variant per call from cfg->op_dtype.
#ifndef CC_OP
#define CC_OP 0
#endif
#if CC_OP == 1
#define CC_NS cc_f16
#define CC_API(name) name##_f16
#elif CC_OP == 2
// CC_OP = 2, "bf16x3" (namespace cc_x3, <name>_x3): the reference's DEFAULT precision (`--fp-precision 32`, clipcap/train/args.py:30-34,
// train.py:82).  gfx950's fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate and there is no xf32, so every GEMM operand x is split into bf16 hi = bf16(x) and lo = bf16(x - hi) and a
// product runs as three bf16 MFMA terms hi*hi + hi*lo + lo*hi with fp32 accumulation (~16 mantissa bits per operand, 1/3 of the bf16
// MFMA rate).  Realised without touching the GEMM main loops: the A operand is laid out [hi | hi | lo] and the B operand [hi | lo | hi]
// along K, i.e. the same NT kernels run with K' = 3K (gemm_x3.hip.h).  Activations between kernels are stored as fp32 (act_t) and split
// at the GEMM boundary; attention runs on the fp32 VALU kernels.  This is the parity mode (logits within 1e-3 of the fp32 reference at
// full depth), not the throughput mode.
#define CC_NS cc_x3
#define CC_API(name) name##_x3
#else
#define CC_NS cc_bf16
#define CC_API(name) name##_bf16
#endif

// process-wide knobs shared by both variants (shared.cpp): test / measurement hooks only, never touched by the product path
namespace cc_shared {
extern int g_gemm_tile_mode;   // cc_gemm_tile_mode
extern int g_gemm_s64;         // cc_gemm_skinny_mode
extern int g_gemm_small_x2;    // env CC_GEMM_X2
extern int g_decode_last_path; // cc_decode_last_path
}  // namespace cc_shared

namespace CC_NS {
using cc_shared::g_gemm_s64;
using cc_shared::g_gemm_small_x2;
using cc_shared::g_gemm_tile_mode;

typedef unsigned short op16_t;  // raw 16-bit operand storage (bf16 or fp16 bit pattern); conversions are round-to-nearest-even like torch
// act_t: element type of the activations the kernels hand each other through HBM (normalised rows, qkv, attention output, MLP hidden,
// their gradients, the logits kept for the backward, the KV cache).  16-bit operand storage in the bf16 / fp16 builds; fp32 in the
// bf16x3 build, where the 16-bit hi / lo operand pair is made at the GEMM boundary.
#if CC_OP == 2
constexpr bool kX3 = true;
typedef float act_t;
#else
constexpr bool kX3 = false;
typedef op16_t act_t;
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2_hw;
#if CC_OP == 1
typedef __attribute__((ext_vector_type(8))) _Float16 op16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 op16x2_hw;
#define CC_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define CC_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
__device__ __forceinline__ float op2f(op16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// one 32-bit word holding two operands -> two floats (v_cvt_f32_f16 on each half)
__device__ __forceinline__ void unpack2(unsigned w, float& lo, float& hi) {
    const f32x2_hw f = __builtin_convertvector(__builtin_bit_cast(op16x2_hw, w), f32x2_hw);
    lo = f.x; hi = f.y;
}
#else
typedef __attribute__((ext_vector_type(8))) __bf16 op16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 op16x2_hw;
#define CC_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define CC_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
__device__ __forceinline__ float op2f(op16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ void unpack2(unsigned w, float& lo, float& hi) {
    lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
}
#endif
// fp32 -> operand type on the gfx950 converters (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32: round-to-nearest-even, NaN stays NaN) — one
// instruction per PAIR; the integer-arithmetic bf16 rounding it replaces cost ~7 VALU instructions and a divergent NaN branch per
// element and made every 16-bit-storing epilogue instruction-bound.
__device__ __forceinline__ unsigned pack2op(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_hw{lo, hi}, op16x2_hw));
}
__device__ __forceinline__ op16_t f2op(float f) { return (op16_t)(pack2op(f, 0.f) & 0xffffu); }
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    unpack2(v.x, f[0], f[1]); unpack2(v.y, f[2], f[3]); unpack2(v.z, f[4], f[5]); unpack2(v.w, f[6], f[7]);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack2op(f[0], f[1]), pack2op(f[2], f[3]), pack2op(f[4], f[5]), pack2op(f[6], f[7]));
}

// ---- stored activations (act_t): 8 / 4 consecutive elements <-> floats.  One 16-B (8-B) vector in the 16-bit builds, two (one)
// 16-B vectors in the bf16x3 build.  p must be aligned to the vector size.
#if CC_OP == 2
struct act_raw8 { float4 a, b; };          // raw register image of 8 stored elements (loaded early, unpacked late)
struct act_raw4 { float4 a; };
__device__ __forceinline__ float act2f(act_t v) { return v; }
__device__ __forceinline__ act_t f2act(float f) { return f; }
__device__ __forceinline__ float act_round(float f) { return f; }      // value a stored activation takes: no rounding here
__device__ __forceinline__ act_raw8 act_ldraw8(const act_t* p) {
    return act_raw8{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)};
}
__device__ __forceinline__ act_raw8 act_zero8() { return act_raw8{make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)}; }
__device__ __forceinline__ void act_unpack8(const act_raw8& r, float (&f)[8]) {
    f[0] = r.a.x; f[1] = r.a.y; f[2] = r.a.z; f[3] = r.a.w; f[4] = r.b.x; f[5] = r.b.y; f[6] = r.b.z; f[7] = r.b.w;
}
__device__ __forceinline__ void act_straw8(act_t* p, const act_raw8& r) {
    *reinterpret_cast<float4*>(p) = r.a; *reinterpret_cast<float4*>(p + 4) = r.b;
}
__device__ __forceinline__ act_raw8 act_pack8(const float (&f)[8]) {
    return act_raw8{make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7])};
}
__device__ __forceinline__ act_raw4 act_ldraw4(const act_t* p) { return act_raw4{*reinterpret_cast<const float4*>(p)}; }
__device__ __forceinline__ void act_unpack4(const act_raw4& r, float& a, float& b, float& c, float& d) { a = r.a.x; b = r.a.y; c = r.a.z; d = r.a.w; }
__device__ __forceinline__ act_raw4 act_pack4(float a, float b, float c, float d) { return act_raw4{make_float4(a, b, c, d)}; }
__device__ __forceinline__ void act_straw4(act_t* p, const act_raw4& r) { *reinterpret_cast<float4*>(p) = r.a; }
#else
struct act_raw8 { uint4 a; };
struct act_raw4 { uint2 a; };
__device__ __forceinline__ float act2f(act_t v) { return op2f(v); }
__device__ __forceinline__ act_t f2act(float f) { return f2op(f); }
__device__ __forceinline__ float act_round(float f) { return op2f(f2op(f)); }
__device__ __forceinline__ act_raw8 act_ldraw8(const act_t* p) { return act_raw8{*reinterpret_cast<const uint4*>(p)}; }
__device__ __forceinline__ act_raw8 act_zero8() { return act_raw8{make_uint4(0, 0, 0, 0)}; }
__device__ __forceinline__ void act_unpack8(const act_raw8& r, float (&f)[8]) { unpack8(r.a, f); }
__device__ __forceinline__ void act_straw8(act_t* p, const act_raw8& r) { *reinterpret_cast<uint4*>(p) = r.a; }
__device__ __forceinline__ act_raw8 act_pack8(const float (&f)[8]) { return act_raw8{pack8(f)}; }
__device__ __forceinline__ act_raw4 act_ldraw4(const act_t* p) { return act_raw4{*reinterpret_cast<const uint2*>(p)}; }
__device__ __forceinline__ void act_unpack4(const act_raw4& r, float& a, float& b, float& c, float& d) { unpack2(r.a.x, a, b); unpack2(r.a.y, c, d); }
__device__ __forceinline__ act_raw4 act_pack4(float a, float b, float c, float d) { return act_raw4{make_uint2(pack2op(a, b), pack2op(c, d))}; }
__device__ __forceinline__ void act_straw4(act_t* p, const act_raw4& r) { *reinterpret_cast<uint2*>(p) = r.a; }
#endif
// non-temporal form for tensors that are written now and read much later (the forward's saved-for-backward copies): they should not
// displace what the next kernels re-read from the 256 MB Infinity Cache
__device__ __forceinline__ void act_st8_nt(act_t* p, const float (&f)[8]) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_hw;
    const act_raw8 r = act_pack8(f);
#if CC_OP == 2
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_hw, r.a), reinterpret_cast<u32x4_hw*>(p));
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_hw, r.b), reinterpret_cast<u32x4_hw*>(p + 4));
#else
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_hw, r.a), reinterpret_cast<u32x4_hw*>(p));
#endif
}
__device__ __forceinline__ void act_ld8(const act_t* p, float (&f)[8]) { act_unpack8(act_ldraw8(p), f); }
__device__ __forceinline__ void act_st8(act_t* p, const float (&f)[8]) { act_straw8(p, act_pack8(f)); }
__device__ __forceinline__ void act_st4(act_t* p, float a, float b, float c, float d) { act_straw4(p, act_pack4(a, b, c, d)); }

// gelu_new (tanh approximation) and its derivative — transformers.activations.NewGELUActivation, in the sigmoid form
//   0.5 (1 + tanh u) = 1 / (1 + exp(-2u)) = s,   u = k0 (x + k1 x^3)   =>   gelu = x s,   gelu' = s + x s (1 - s) 2 k0 (1 + 3 k1 x^2)
// evaluated on the hardware exp2 / rcp units (v_exp_f32 / v_rcp_f32, ~1e-7 relative): 5 (8) VALU + 2 transcendental operations per
// element instead of ~12 (~20) for the literal tanh formulation.  These epilogues are VALU-bound: the c_fc forward / activation-
// gradient GEMMs (K = 768) spend as long in them as in their MFMA loop.  Saturates correctly: exp2 -> inf gives s = 0, exp2 -> 0 s = 1.
__device__ __forceinline__ float gelu_sigmoid(float x, float x2) {
    constexpr float LOG2E = 1.4426950408889634f, K0 = 0.7978845608028654f, K1 = 0.044715f;
    const float p = __builtin_fmaf(x2, -2.0f * K0 * K1 * LOG2E, -2.0f * K0 * LOG2E);      // -2u log2(e) = x p
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
}
__device__ __forceinline__ float gelu_new_f(float x) { return x * gelu_sigmoid(x, x * x); }
// gelu_new and its derivative from ONE sigmoid evaluation (the forward epilogue that also stores the derivative for the backward pass)
__device__ __forceinline__ void gelu_new_both(float x, float& g, float& dg) {
    constexpr float K0 = 0.7978845608028654f, K1 = 0.044715f;
    const float x2 = x * x;
    const float s = gelu_sigmoid(x, x2);
    const float q = __builtin_fmaf(x2, 6.0f * K0 * K1, 2.0f * K0);
    g = x * s;
    dg = __builtin_fmaf(x * q, __builtin_fmaf(-s, s, s), s);
}
__device__ __forceinline__ float gelu_new_grad(float x) {
    constexpr float K0 = 0.7978845608028654f, K1 = 0.044715f;
    const float x2 = x * x;
    const float s = gelu_sigmoid(x, x2);
    const float q = __builtin_fmaf(x2, 6.0f * K0 * K1, 2.0f * K0);                        // d(2u)/dx
    return __builtin_fmaf(x * q, __builtin_fmaf(-s, s, s), s);                            // s + x q s (1 - s)
}

// ---- dropout (GPT-2 full finetune in train mode: embd / attention-probability / residual dropout, hf modeling_gpt2.py) --------
// Counter-based: keep(element) is a hash of (seed, stream = site * 256 + layer, element index), so the backward pass regenerates the
// forward's mask instead of storing it.  One 32-bit hash serves TWO neighbouring elements (idx >> 1; low / high 16 bits), so the vector
// call sites (drop_mul_pair) pay half a hash per element — the hash's three 32-bit multiplies are quarter-rate VALU operations and
// made the residual-dropout epilogues as expensive as a gelu.  thresh = p * 2^16 (0 = dropout off, p is honoured to 1.5e-5); kept
// values are scaled by 1 / (1 - p).
struct Drop {
    unsigned thresh = 0, seed_lo = 0, seed_hi = 0, stream = 0;
    float scale = 1.0f;
};
__host__ __device__ __forceinline__ unsigned drop_hash(unsigned seed_lo, unsigned seed_hi, unsigned stream, unsigned idx) {
    unsigned x = idx ^ seed_lo;
    x *= 0x9E3779B1u; x ^= x >> 15;
    x += (stream * 0x85EBCA6Bu) ^ seed_hi;
    x *= 0xC2B2AE35u; x ^= x >> 13;
    x *= 0x27D4EB2Fu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool drop_keep(const Drop& d, unsigned idx) {
    const unsigned h = drop_hash(d.seed_lo, d.seed_hi, d.stream, idx >> 1);
    return ((idx & 1u) ? (h >> 16) : (h & 0xffffu)) >= d.thresh;
}
// multiplier of element idx: 0 or 1 / (1 - p)
__device__ __forceinline__ float drop_mul(const Drop& d, unsigned idx) { return drop_keep(d, idx) ? d.scale : 0.f; }
// multipliers of elements idx_even, idx_even + 1 (idx_even must be even) from ONE hash
__device__ __forceinline__ void drop_mul_pair(const Drop& d, unsigned idx_even, float& m0, float& m1) {
    const unsigned h = drop_hash(d.seed_lo, d.seed_hi, d.stream, idx_even >> 1);
    m0 = (h & 0xffffu) >= d.thresh ? d.scale : 0.f;
    m1 = (h >> 16) >= d.thresh ? d.scale : 0.f;
}
enum { DROP_EMBD = 0, DROP_ATTN = 1, DROP_RESID_ATTN = 2, DROP_RESID_MLP = 3 };
inline Drop make_drop(float p, unsigned long long seed, unsigned site, unsigned layer) {
    Drop d;
    if (p > 0.f) {
        const double t = (double)p * 65536.0 + 0.5;
        d.thresh = t >= 65535.0 ? 65535u : (unsigned)t;
        d.scale = 1.0f / (1.0f - p);
        d.seed_lo = (unsigned)seed; d.seed_hi = (unsigned)(seed >> 32); d.stream = site * 256u + layer;
    }
    return d;
}

// Wave reductions on DPP (data-parallel primitives: quad_perm / row_half_mirror / row_mirror inside each row of 16 lanes, then the four
// row results through v_readlane) instead of the xor-butterfly of __shfl_xor, which hipcc lowers to ds_bpermute_b32: six DEPENDENT LDS
// round trips per reduction.  With many waves per SIMD those hide; where a wave runs (almost) alone — the decode finish rows, the decode
// attention, the XCD-team engine's I/O waves — they are the kernel's latency (round 5: a LayerNorm row is two reductions = 12 trips).
// The result is wave-uniform.  Summation order differs from the butterfly's (fp32 rounding only).  CC_WAVE_SHFL restores the old form (A/B).
// PRECONDITION: the FULL wave calls these (EXEC = all 64 lanes) — update_dpp with old = 0 / bound_ctrl off substitutes 0 for an
// inactive source lane (wrong for wave_max of negative values) and v_readlane of an inactive lane returns stale register contents.  Every call
// site here is wave-uniform (block sizes are multiples of 64, no early exit before a reduction); a kernel that cannot guarantee that must build
// with CC_WAVE_SHFL, whose shuffles degrade per lane instead.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// sum / max over each aligned group of 8 lanes, in every lane of the group (the three xor steps 1, 2, 4 of a butterfly)
__device__ __forceinline__ float sum8(float v) {
#ifdef CC_WAVE_SHFL
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
#else
    v += dpp_mov<0xB1>(v);                                  // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);                                  // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);                                 // row_half_mirror
#endif
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#ifdef CC_WAVE_SHFL
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#else
    v = sum8(v);
    v += dpp_mov<0x140>(v);                                 // row_mirror: every lane holds its row's sum
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef CC_WAVE_SHFL
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
#else
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
#endif
}

// XCD-aware block order: the dispatcher places block b on XCD b%8; give each XCD a contiguous run of logical blocks (bijective for any
// grid size), so that neighbours in the logical order — GEMM tiles sharing an A row panel, the beams of one caption in the decode
// attention — meet in the same L2.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

}  // namespace CC_NS

// status codes returned across the C ABI (same values as include/clipcap_hip.h)
#ifndef CC_OK
#define CC_OK 0
#define CC_ERR_ARG (-1)
#define CC_ERR_SHAPE (-2)
#define CC_ERR_LAUNCH (-3)
#define CC_ERR_STATE (-4)
#endif
