// C ABI (include/clipcap_hip.h) — orchestration of the mapper and GPT-2 training/inference passes out of the HIP
// kernels in gemm.hip.h / kernels.hip.  No state, no allocation, no synchronisation: everything is enqueued on the
// caller's stream and lives in caller-owned arenas / workspaces.
#include "../../include/clipcap_hip.h"
#include "gemm_api.h"
#include "kernels.h"
#include "shared.h"

using namespace CC_NS;

#define CC_TRY(expr)                 \
    do {                             \
        int _e = (expr);             \
        if (_e != CC_OK) return _e;  \
    } while (0)

#include <vector>

namespace {

constexpr int MAX_LAYERS = 96;

#define CC_TIMED(site, st, expr)                 \
    do {                                         \
        cc_shared::ProfScope _ps(site, st);      \
        int _e = (expr);                         \
        if (_e != CC_OK) return _e;              \
    } while (0)

// tuning knob (environment, read once): force a GEMM tile kernel at ONE call-site class for A/B measurements —
// CC_TILE_FC = c_fc forward (gelu epilogue, two outputs), CC_TILE_DACT = its activation-gradient dgrad, CC_TILE_PROJ / CC_TILE_PROJ2 = the
// two residual-epilogue c_proj forwards.  Unset = the chooser.
struct TileScope {
    int old;
    bool on;
    explicit TileScope(int mode) : old(cc_shared::g_gemm_tile_mode), on(mode != -2) { if (on) cc_shared::g_gemm_tile_mode = mode; }
    ~TileScope() { if (on) cc_shared::g_gemm_tile_mode = old; }
};
inline int env_tile(const char* name) {
    const char* e = cc_lab_env(name);
    return e ? atoi(e) : -2;
}

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <class T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return r;
    }
};

inline hipStream_t S_(void* s) { return static_cast<hipStream_t>(s); }
inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// Operand arena addressing.  16-bit builds: w16[off] is the cast of w32[off], w16[total + off] the transposed copy.  bf16x3 build: every
// 2-D GEMM weight [R][C] at element offset off owns 3*R*C operand elements at 3*off — rows [hi | lo | hi] of 3C (the B-operand image,
// common.hip.h) — and its transposed image [C][3R] at 3*(total + off); the arena has 6*total elements (1-D tensors leave holes).
constexpr int PL = kX3 ? 3 : 1;
inline const uint16_t* W16(const uint16_t* w16, int64_t off) { return w16 + (size_t)PL * off; }
inline uint16_t* W16(uint16_t* w16, int64_t off) { return w16 + (size_t)PL * off; }
// bf16x3: bytes of operand-image scratch for GEMMs whose largest A image is rows x depth (x3_operand rounds each image up to 256 B)
inline size_t x3_img(size_t rows, size_t depth) { return ((rows * 3 * depth * sizeof(op16_t)) + 255) & ~size_t(255); }
#if CC_OP == 2
#define X3_SCRATCH(w) x3_set_scratch((w).x3, (w).x3_bytes)
#else
#define X3_SCRATCH(w) (void)0
#endif

// ------------------------------------------------------------------------------------------------------------
// mapper
// ------------------------------------------------------------------------------------------------------------
struct MapperOff {
    int64_t lin_w, lin_b, prefix, pos;
    struct Layer {
        int64_t n1w, n1b, wq, wkv, wp, bp, n2w, n2b, w1, b1, w2, b2;
    } layer[MAX_LAYERS];
    int64_t total;
};

bool mapper_cfg_ok(const cc_mapper_cfg* c) {
    return c && (c->op_dtype == CC_OP) && c->E > 0 && c->D > 0 && c->P > 0 && c->L > 0 && c->H > 0 && c->N >= 0 && c->N <= MAX_LAYERS && c->Hm > 0 && c->W >= 1 &&
           (c->E % 8) == 0 && (c->D % 8) == 0 && (c->Hm % 8) == 0 && (c->D % c->H) == 0 && ((c->D / c->H) % 8) == 0;
}

void mapper_offsets(const cc_mapper_cfg* c, MapperOff& o) {
    int64_t p = 0;
    const int64_t D = c->D, E = c->E, PD = (int64_t)c->P * D, Hm = c->Hm;
    o.lin_w = p; p += PD * E;
    o.lin_b = p; p += PD;
    o.prefix = p; p += (int64_t)c->L * D;
    if (c->W > 1 && c->use_pos) { o.pos = p; p += (int64_t)c->W * PD; } else o.pos = -1;
    for (int l = 0; l < c->N; l++) {
        auto& y = o.layer[l];
        y.n1w = p; p += D;
        y.n1b = p; p += D;
        y.wq = p; p += D * D;
        y.wkv = p; p += 2 * D * D;
        y.wp = p; p += D * D;
        y.bp = p; p += D;
        y.n2w = p; p += D;
        y.n2b = p; p += D;
        y.w1 = p; p += Hm * D;
        y.b1 = p; p += Hm;
        y.w2 = p; p += D * Hm;
        y.b2 = p; p += D;
    }
    o.total = p;
}

constexpr int kDeferRows = 40960;      // cc_mapper_bwd_range defers the weight gradients of a call up to this many rows (and 8 layers)
struct MapperWS {
    act_t* emb16;
    float* lin_tmp;
    float* x[MAX_LAYERS + 1];
    float* x1[MAX_LAYERS];
    act_t *xn1[MAX_LAYERS], *xn2[MAX_LAYERS], *qkv[MAX_LAYERS], *att[MAX_LAYERS], *h[MAX_LAYERS];
    float *lse[MAX_LAYERS], *mean1[MAX_LAYERS], *rstd1[MAX_LAYERS], *mean2[MAX_LAYERS], *rstd2[MAX_LAYERS];
    // backward scratch
    float* dx32;
    act_t *dx16, *dx16b, *dh16, *dxn16, *datt16, *dqkv16, *dlin16;
    // per-layer copies of the four output gradients that are weight-gradient operands (round 5): with them every layer's weight gradients can
    // wait for ONE grouped launch at the end of the backward call (gdx[l] = d x[l+1], the gradient entering layer l from above).  The bf16x3
    // build splits operands per call and keeps the single buffers: there every entry aliases them.
    act_t *gdx[MAX_LAYERS], *gdxb[MAX_LAYERS], *gdh[MAX_LAYERS], *gdqkv[MAX_LAYERS];
    bool own_grads;     // every layer has its own copies (else they alias the single dx16 / dx16b / dh16 / dqkv16 buffers and weight gradients flush per layer)
    float* wg_scratch;
    float* adelta;
    uint16_t* gimg;    // bf16x3 build: the [hi | hi | lo] image of the output gradient both GEMMs of a layer step read
    char* x3;          // bf16x3 build: operand-image scratch of the GEMM in flight (gemm_api.h)
    size_t x3_bytes;
    size_t bytes;
};

void mapper_carve(const cc_mapper_cfg* c, int B, int save, void* ws, MapperWS& w) {
    Carver cv(ws);
    const int S = c->W * c->P + c->L;
    const size_t M = (size_t)B * S, D = c->D;
    w.emb16 = cv.take<act_t>((size_t)B * c->W * c->E);
    w.lin_tmp = c->W > 1 ? cv.take<float>((size_t)B * c->W * c->P * D) : nullptr;
    const int nx = save ? c->N + 1 : 2;
    float* xb[MAX_LAYERS + 1];
    for (int i = 0; i < nx; i++) xb[i] = cv.take<float>(M * D);
    for (int l = 0; l <= c->N; l++) w.x[l] = save ? xb[l] : xb[l & 1];
    const int nl = save ? c->N : 1;
    for (int l = 0; l < c->N; l++) {
        const bool fresh = l < nl;
        const int s = fresh ? l : 0;
        if (fresh) {
            w.x1[l] = cv.take<float>(M * D);
            w.xn1[l] = cv.take<act_t>(M * D);
            w.xn2[l] = cv.take<act_t>(M * D);
            w.qkv[l] = cv.take<act_t>(M * 3 * D);
            w.att[l] = cv.take<act_t>(M * D);
            w.h[l] = cv.take<act_t>(M * c->Hm);
            w.lse[l] = cv.take<float>((size_t)B * c->H * S);
            w.mean1[l] = cv.take<float>(M);
            w.rstd1[l] = cv.take<float>(M);
            w.mean2[l] = cv.take<float>(M);
            w.rstd2[l] = cv.take<float>(M);
        } else {
            w.x1[l] = w.x1[s]; w.xn1[l] = w.xn1[s]; w.xn2[l] = w.xn2[s]; w.qkv[l] = w.qkv[s]; w.att[l] = w.att[s]; w.h[l] = w.h[s];
            w.lse[l] = w.lse[s]; w.mean1[l] = w.mean1[s]; w.rstd1[l] = w.rstd1[s]; w.mean2[l] = w.mean2[s]; w.rstd2[l] = w.rstd2[s];
        }
    }
    if (save) {
        w.dx32 = cv.take<float>(M * D);
        w.dx16 = cv.take<act_t>(M * D);
        w.dx16b = cv.take<act_t>(M * D);        // gradient after the LN2 backward: a second buffer so that the layer's four weight
                                                // gradients can run as one grouped launch once all their operands exist
        w.dh16 = cv.take<act_t>(M * c->Hm);
        w.dxn16 = cv.take<act_t>(M * D);
        w.datt16 = cv.take<act_t>(M * D);
        w.dqkv16 = cv.take<act_t>(M * 3 * D);
        w.dlin16 = cv.take<act_t>((size_t)B * c->W * c->P * D);
        // per-layer copies only where cc_mapper_bwd_range can defer (<= kDeferRows rows and <= 8 layers): a B = 4096 call would otherwise
        // carry (N - 1) x M x (5 D + Hm) unused 16-bit elements (~6 GB at D = 768, N = 8) — ADVICE r5
        w.own_grads = !kX3 && M <= (size_t)kDeferRows && c->N <= 8;
        for (int l = 0; l < c->N; l++) {
            const bool own = w.own_grads;                // every layer its own (a deferred weight gradient reads them at the END of the call)
            w.gdx[l] = own ? cv.take<act_t>(M * D) : w.dx16;
            w.gdxb[l] = own ? cv.take<act_t>(M * D) : w.dx16b;
            w.gdh[l] = own ? cv.take<act_t>(M * c->Hm) : w.dh16;
            w.gdqkv[l] = own ? cv.take<act_t>(M * 3 * D) : w.dqkv16;
        }
        w.wg_scratch = cv.take<float>(WGRAD_SCRATCH_BYTES / sizeof(float));
        w.adelta = cv.take<float>((size_t)B * c->H * S);
        // bf16x3: an output gradient is the operand of TWO GEMMs in the same image form (its layer's weight gradient and input gradient):
        // split once into this buffer, read twice (the widest is d qkv)
        w.gimg = kX3 ? reinterpret_cast<uint16_t*>(cv.take<char>(x3_img(M, std::max<size_t>(3 * (size_t)D, c->Hm)))) : nullptr;
    } else {
        w.dx32 = nullptr; w.dx16 = w.dx16b = w.dh16 = w.dxn16 = w.datt16 = w.dqkv16 = w.dlin16 = nullptr; w.wg_scratch = nullptr; w.adelta = nullptr;
        w.gimg = nullptr;
        for (int l = 0; l < c->N; l++) w.gdx[l] = w.gdxb[l] = w.gdh[l] = w.gdqkv[l] = nullptr;
        w.own_grads = false;
    }
    w.x3 = nullptr; w.x3_bytes = 0;
    if (kX3) {
        const size_t Hm = c->Hm, Mb = (size_t)B * c->W, PD = (size_t)c->P * D;
        size_t need = std::max(x3_img(Mb, c->E), x3_img(M, std::max(D, Hm)));                          // forward A images
        if (save) {
            need = std::max(need, x3_img(M, 3 * D));                                                     // dgrad through the fused QKV weight
            need = std::max(need, std::max(x3_img(M, D) + x3_img(M, Hm), x3_img(M, 3 * D) + x3_img(M, D)));   // weight-gradient pairs
            need = std::max(need, x3_img(Mb, PD) + x3_img(Mb, c->E));
        }
        w.x3_bytes = need;
        w.x3 = cv.take<char>(need);
    }
    w.bytes = (cv.off + 255) & ~size_t(255);
}

// ------------------------------------------------------------------------------------------------------------
// GPT-2
// ------------------------------------------------------------------------------------------------------------
struct Gpt2Off {
    int64_t wte, wpe, lnf_w, lnf_b;
    struct Layer {
        int64_t l1w, l1b, aw, ab, pw, pb, l2w, l2b, fw, fb, p2w, p2b;
    } layer[MAX_LAYERS];
    int64_t total;
};

bool gpt2_cfg_ok(const cc_gpt2_cfg* c) {
    return c && (c->op_dtype == CC_OP) && c->D > 0 && c->H > 0 && c->NL > 0 && c->NL <= MAX_LAYERS && c->V > 0 && c->Vp >= c->V && (c->Vp % 128) == 0 &&
           c->NPOS > 0 && (c->D % 8) == 0 && (c->D % c->H) == 0 && ((c->D / c->H) % 8) == 0;
}

void gpt2_offsets(const cc_gpt2_cfg* c, Gpt2Off& o) {
    int64_t p = 0;
    const int64_t D = c->D;
    o.wte = p; p += (int64_t)c->Vp * D;
    o.wpe = p; p += (int64_t)c->NPOS * D;
    for (int l = 0; l < c->NL; l++) {
        auto& y = o.layer[l];
        y.l1w = p; p += D;
        y.l1b = p; p += D;
        y.aw = p; p += D * 3 * D;
        y.ab = p; p += 3 * D;
        y.pw = p; p += D * D;
        y.pb = p; p += D;
        y.l2w = p; p += D;
        y.l2b = p; p += D;
        y.fw = p; p += D * 4 * D;
        y.fb = p; p += 4 * D;
        y.p2w = p; p += 4 * D * D;
        y.p2b = p; p += D;
    }
    o.lnf_w = p; p += D;
    o.lnf_b = p; p += D;
    o.total = p;
}

struct Gpt2WS {
    float* x[MAX_LAYERS + 1];
    float* x1[MAX_LAYERS];
    act_t *xn1[MAX_LAYERS], *xn2[MAX_LAYERS], *qkv[MAX_LAYERS], *att[MAX_LAYERS], *u[MAX_LAYERS], *hact[MAX_LAYERS];
    float *lse[MAX_LAYERS], *mean1[MAX_LAYERS], *rstd1[MAX_LAYERS], *mean2[MAX_LAYERS], *rstd2[MAX_LAYERS];
    // lm head / loss
    act_t* hf16;       // [Mh, D]   ln_f output rows (Mh = max(B*cap, B*T) so the parity API can use it too)
    float *meanf, *rstdf;
    int *target, *row_map;
    act_t* logits16;   // [B*cap, Vp]
    float *pmax, *psum, *tgt_logit, *lse_row, *row_loss;
    float *cref, *lmfac;   // exponential form of the lm_head outputs (bf16 build): reference shift [Mc], row factors {r, w} [Mc][2]
    act_t* hfs16;          // full finetune: r * hf rows, the weight-gradient operand of the exponential form
    // backward
    float* dx32;
    act_t *dx16, *dx16b, *dhf16, *du16, *dxn16, *datt16, *dqkv16;
    float* wg_scratch;
    float* adelta;
    char* x3;          // bf16x3 build: operand-image scratch
    size_t x3_bytes;
    size_t bytes;
};

void gpt2_carve(const cc_gpt2_cfg* c, int B, int T, int cap, int mode, void* ws, Gpt2WS& w) {
    Carver cv(ws);
    const size_t M = (size_t)B * T, D = c->D, Mc = (size_t)B * cap;
    const bool keep = mode >= 1, full = mode >= 2;
    const int nx = keep ? c->NL + 1 : 2;
    float* xb[MAX_LAYERS + 1];
    for (int i = 0; i < nx; i++) xb[i] = cv.take<float>(M * D);
    for (int l = 0; l <= c->NL; l++) w.x[l] = keep ? xb[l] : xb[l & 1];
    for (int l = 0; l < c->NL; l++) {
        const bool k0 = keep || l == 0, f0 = full || l == 0;
        w.x1[l] = k0 ? cv.take<float>(M * D) : w.x1[0];
        w.qkv[l] = k0 ? cv.take<act_t>(M * 3 * D) : w.qkv[0];
        w.u[l] = k0 ? cv.take<act_t>(M * 4 * D) : w.u[0];
        w.lse[l] = k0 ? cv.take<float>((size_t)B * c->H * T) : w.lse[0];
        w.mean1[l] = k0 ? cv.take<float>(M) : w.mean1[0];
        w.rstd1[l] = k0 ? cv.take<float>(M) : w.rstd1[0];
        w.mean2[l] = k0 ? cv.take<float>(M) : w.mean2[0];
        w.rstd2[l] = k0 ? cv.take<float>(M) : w.rstd2[0];
        w.xn1[l] = f0 ? cv.take<act_t>(M * D * ((kX3 && !full) ? 3 : 2) / 2) : w.xn1[0];       // bf16x3, frozen LM: LayerNorm writes c_attn's / c_fc's operand image
        w.xn2[l] = f0 ? cv.take<act_t>(M * D * ((kX3 && !full) ? 3 : 2) / 2) : w.xn2[0];
        w.att[l] = k0 ? cv.take<act_t>(M * D * ((kX3 && !full) ? 3 : 2) / 2) : w.att[0];    // attention output: needed by the backward's delta = rowsum(dO*O) (bf16x3, frozen LM: attn.c_proj's operand image)
        w.hact[l] = f0 ? cv.take<act_t>(M * 4 * D * ((kX3 && !full) ? 3 : 2) / 2) : w.hact[0];    // bf16x3, frozen LM: holds mlp.c_proj's operand IMAGE (6 B / element), written by c_fc's epilogue
    }
    const size_t Mh = std::max(M, Mc);
    w.hf16 = cv.take<act_t>(Mh * D);
    w.meanf = cv.take<float>(Mh);
    w.rstdf = cv.take<float>(Mh);
    w.target = cv.take<int>(Mh);
    w.row_map = cv.take<int>(Mh);
    if (keep) {
        const int npart = c->Vp / 64;
        w.logits16 = cv.take<act_t>(Mc * c->Vp * ((kX3 && !full) ? 3 : 2) / 2);               // bf16x3, frozen LM: E as an operand image (lm_exp_form)
        w.pmax = cv.take<float>(Mc * npart);
        w.psum = cv.take<float>(Mc * npart);
        w.tgt_logit = cv.take<float>(Mc);
        w.lse_row = cv.take<float>(Mc);
        w.row_loss = cv.take<float>(Mc);
        w.cref = cv.take<float>(Mc);
        w.lmfac = cv.take<float>(2 * Mc);
        w.hfs16 = full ? cv.take<act_t>(Mc * D) : nullptr;
        w.dx32 = cv.take<float>(M * D);
        w.dx16 = cv.take<act_t>(M * D * ((kX3 && !full) ? 3 : 2) / 2);                            // bf16x3, frozen LM: operand images written by their producers (LayerNorm backward, attention backward)
        w.dx16b = full ? cv.take<act_t>(M * D) : w.dx16;    // full finetune: second copy so a layer's weight gradients can run grouped
        w.dhf16 = cv.take<act_t>(Mc * D);
        w.du16 = cv.take<act_t>(M * 4 * D * ((kX3 && !full) ? 3 : 2) / 2);                       // likewise: c_fc's input-gradient operand image, written by the gelu' epilogue
        w.dxn16 = cv.take<act_t>(M * D);
        w.datt16 = cv.take<act_t>(M * D);
        w.dqkv16 = cv.take<act_t>(M * 3 * D * ((kX3 && !full) ? 3 : 2) / 2);
        w.wg_scratch = full ? cv.take<float>(WGRAD_SCRATCH_BYTES / sizeof(float)) : nullptr;
        w.adelta = cv.take<float>((size_t)B * c->H * T);
    } else {
        w.wg_scratch = nullptr; w.adelta = nullptr;
        w.logits16 = nullptr; w.pmax = w.psum = w.tgt_logit = w.lse_row = w.row_loss = nullptr;
        w.cref = w.lmfac = nullptr; w.hfs16 = nullptr;
        w.dx32 = nullptr; w.dx16 = w.dx16b = w.dhf16 = w.du16 = w.dxn16 = w.datt16 = w.dqkv16 = nullptr;
    }
    w.x3 = nullptr; w.x3_bytes = 0;
    if (kX3) {
        size_t need = x3_img(Mh, 4 * D);                                                       // forward A images (mlp.c_proj is the deepest)
        if (keep) need = std::max(need, x3_img(Mc, c->Vp));                                    // lm_head input gradient: A = dlogits
        if (full) need = std::max(need, std::max(x3_img(M, D) + x3_img(M, 4 * D), x3_img(Mc, c->Vp) + x3_img(Mc, D)));   // weight-gradient pairs
        w.x3_bytes = need;
        w.x3 = cv.take<char>(need);
    }
    w.bytes = (cv.off + 255) & ~size_t(255);
}

}  // namespace

extern "C" {

int CC_API(cc_mapper_bwd_range)(const cc_mapper_cfg* c, int32_t B, const float* w32, const uint16_t* w16, void* ws, const float* dout, float* g32,
                        int32_t l_hi, int32_t l_lo, void* stream);
int CC_API(cc_gpt2_bwd_range)(const cc_gpt2_cfg* c, const cc_gpt2_shape* s, const float* w32, const uint16_t* w16, void* ws, const int64_t* tokens,
                      float* dprefix, float* g32, int32_t l_hi, int32_t l_lo, void* stream);

// ---------------------------------------------------------------- mapper ----------------------------------------
int64_t CC_API(cc_mapper_param_count)(const cc_mapper_cfg* cfg) {
    if (!mapper_cfg_ok(cfg)) return CC_ERR_SHAPE;
    MapperOff o;
    mapper_offsets(cfg, o);
    return o.total;
}

int CC_API(cc_mapper_param_offsets)(const cc_mapper_cfg* cfg, int64_t* offs) {
    if (!mapper_cfg_ok(cfg) || !offs) return CC_ERR_SHAPE;
    MapperOff o;
    mapper_offsets(cfg, o);
    int k = 0;
    offs[k++] = o.lin_w; offs[k++] = o.lin_b; offs[k++] = o.prefix; offs[k++] = o.pos;
    for (int l = 0; l < cfg->N; l++) {
        const auto& y = o.layer[l];
        const int64_t v[12] = {y.n1w, y.n1b, y.wq, y.wkv, y.wp, y.bp, y.n2w, y.n2b, y.w1, y.b1, y.w2, y.b2};
        for (int i = 0; i < 12; i++) offs[k++] = v[i];
    }
    return CC_OK;
}

int64_t CC_API(cc_mapper_ws_bytes)(const cc_mapper_cfg* cfg, int32_t B, int32_t save) {
    if (!mapper_cfg_ok(cfg) || B <= 0) return CC_ERR_SHAPE;
    MapperWS w;
    mapper_carve(cfg, B, save, nullptr, w);
    return (int64_t)w.bytes;
}

#if CC_OP == 2
// bf16x3: [hi | lo | hi] images of every GEMM weight and of its transpose, straight from the fp32 master (W16 layout above)
static int mapper_sync_x3(const cc_mapper_cfg* c, const float* w32, uint16_t* w16, hipStream_t st) {
    MapperOff o;
    mapper_offsets(c, o);
    const int D = c->D, Hm = c->Hm;
    X3SplitBatch sb;
    sb.add(w32 + o.lin_w, W16(w16, o.lin_w), c->P * D, c->E, 0, 1);
    for (int l = 0; l < c->N; l++) {
        const auto& y = o.layer[l];
        const int64_t off[4] = {y.wq, y.wp, y.w1, y.w2};
        const int R[4] = {3 * D, D, Hm, D}, C[4] = {D, D, D, Hm};
        for (int i = 0; i < 4; i++) {
            if (sb.n + 2 > 32) { CC_TRY(x3_split_multi(sb, st)); sb.n = 0; }
            sb.add(w32 + off[i], W16(w16, off[i]), R[i], C[i], 0, 1);
            sb.add(w32 + off[i], W16(w16, o.total + off[i]), R[i], C[i], 1, 1);
        }
    }
    return x3_split_multi(sb, st);
}
#endif

static int mapper_transposes(const cc_mapper_cfg* c, uint16_t* w16, hipStream_t st) {
    if (kX3) return CC_ERR_ARG;      // the operand images are made from the fp32 master (cc_mapper_sync_weights)
    MapperOff o;
    mapper_offsets(c, o);
    op16_t* t = w16 + o.total;
    const int D = c->D, Hm = c->Hm;
    TransposeBatch tb;                      // 8 layers x 4 matrices: one launch per 32 matrices instead of one each
    for (int l = 0; l < c->N; l++) {
        const auto& y = o.layer[l];
        tb.add(w16 + y.wq, t + y.wq, 3 * D, D);   // fused [3D, D] -> [D, 3D]
        tb.add(w16 + y.wp, t + y.wp, D, D);
        tb.add(w16 + y.w1, t + y.w1, Hm, D);       // [Hm, D] -> [D, Hm]
        tb.add(w16 + y.w2, t + y.w2, D, Hm);       // [D, Hm] -> [Hm, D]
        if (tb.n == 32) { CC_TRY(transpose_bf16_multi(tb, st)); tb.n = 0; }
    }
    CC_TRY(transpose_bf16_multi(tb, st));
    return CC_OK;
}

int CC_API(cc_mapper_sync_weights)(const cc_mapper_cfg* c, const float* w32, uint16_t* w16, void* stream) {
    if (!mapper_cfg_ok(c) || !w32 || !w16) return CC_ERR_ARG;
    hipStream_t st = S_(stream);
#if CC_OP == 2
    return mapper_sync_x3(c, w32, w16, st);
#else
    MapperOff o;
    mapper_offsets(c, o);
    CC_TRY(f32_to_bf16(w32, w16, (size_t)o.total, st));
    return mapper_transposes(c, w16, st);
#endif
}

int CC_API(cc_mapper_transpose_weights)(const cc_mapper_cfg* c, uint16_t* w16, void* stream) {
    if (!mapper_cfg_ok(c) || !w16) return CC_ERR_ARG;
    return mapper_transposes(c, w16, S_(stream));
}

int CC_API(cc_mapper_fwd)(const cc_mapper_cfg* c, int32_t B, const float* w32, const uint16_t* w16, const float* emb, void* ws, float* out,
                  int32_t save, void* stream) {
    if (!mapper_cfg_ok(c) || B <= 0 || !w32 || !w16 || !emb || !ws || !out) return CC_ERR_ARG;
    hipStream_t st = S_(stream);
    MapperOff o;
    mapper_offsets(c, o);
    MapperWS w;
    mapper_carve(c, B, save, ws, w);
    X3_SCRATCH(w);
    const int D = c->D, PP = c->W * c->P, S = PP + c->L, M = B * S, H = c->H, hd = D / H, Hm = c->Hm;
    const int PD = c->P * D;
    // linear (mapper.py:123): [B*W, E] x [P*D, E]^T + b -> rows 0..PP-1 of every sample of x[0]
    CC_TRY(f32_to_act(emb, w.emb16, (size_t)B * c->W * c->E, st));
    if (c->W == 1) {
        CC_TRY(gemm_f32out(0, 0, w.emb16, c->E, W16(w16, o.lin_w), c->E, B, PD, c->E, w.x[0], S * D, w32 + o.lin_b, 0, 1.0f, 1, st));
    } else {
        CC_TRY(gemm_f32out(0, 0, w.emb16, c->E, W16(w16, o.lin_w), c->E, B * c->W, PD, c->E, w.lin_tmp, PD, w32 + o.lin_b, 0, 1.0f, 1, st));
        CC_TRY(copy_rows(w.lin_tmp, (size_t)PP * D, w.x[0], (size_t)S * D, PP * D, B, st));
        if (o.pos >= 0) CC_TRY(add_rows(w.x[0], (size_t)S * D, w32 + o.pos, PP * D, B, st));
    }
    // cat learned prefix_const (mapper.py:125-126)
    CC_TRY(broadcast_rows(w.x[0] + (size_t)PP * D, (size_t)S * D, w32 + o.prefix, c->L * D, B, st));
    for (int l = 0; l < c->N; l++) {
        const auto& y = o.layer[l];
        // x1 = x + project(attn(LN1 x))  (mapper.py:108, attention.py:17-43)
        CC_TRY(ln_fwd(w.x[l], D, nullptr, w32 + y.n1w, w32 + y.n1b, w.xn1[l], nullptr, w.mean1[l], w.rstd1[l], M, D, st));
        CC_TIMED(CC_SITE_MAPPER_QKV_FWD, st, gemm_bf16out(0, 0, w.xn1[l], D, W16(w16, y.wq), D, M, 3 * D, D, w.qkv[l], 3 * D, nullptr, 0, nullptr, st));
        CC_TRY(attn_fwd(w.qkv[l], B, S, H, hd, false, w.att[l], w.lse[l], st));
        CC_TRY(gemm_resid(0, 0, w.att[l], D, W16(w16, y.wp), D, M, D, D, w.x1[l], w.x[l], D, w32 + y.bp, st));
        // x = x1 + fc2(relu(fc1(LN2 x1)))  (mapper.py:109, :82-88)
        CC_TRY(ln_fwd(w.x1[l], D, nullptr, w32 + y.n2w, w32 + y.n2b, w.xn2[l], nullptr, w.mean2[l], w.rstd2[l], M, D, st));
        CC_TIMED(CC_SITE_MAPPER_FC1_FWD, st, gemm_bf16out(0, 0, w.xn2[l], D, W16(w16, y.w1), D, M, Hm, D, w.h[l], Hm, w32 + y.b1, 1, nullptr, st));
        CC_TRY(gemm_resid(0, 0, w.h[l], Hm, W16(w16, y.w2), Hm, M, D, Hm, w.x[l + 1], w.x1[l], D, w32 + y.b2, st));
    }
    // out = rows [PP:] (mapper.py:128)
    CC_TRY(copy_rows(w.x[c->N] + (size_t)PP * D, (size_t)S * D, out, (size_t)c->L * D, c->L * D, B, st));
    return CC_OK;
}

int CC_API(cc_mapper_attention_probs)(const cc_mapper_cfg* c, int32_t B, void* ws, int32_t layer, float* out, void* stream) {
    if (!mapper_cfg_ok(c) || B <= 0 || !ws || !out || layer < 0 || layer >= c->N) return CC_ERR_ARG;
    MapperWS w;
    mapper_carve(c, B, 1, ws, w);
    const int S = c->W * c->P + c->L;
    return attn_probs(w.qkv[layer], B, S, c->H, c->D / c->H, out, S_(stream));
}

int CC_API(cc_mapper_bwd)(const cc_mapper_cfg* c, int32_t B, const float* w32, const uint16_t* w16, void* ws, const float* dout, float* g32,
                  void* stream) {
    if (!mapper_cfg_ok(c)) return CC_ERR_ARG;
    return CC_API(cc_mapper_bwd_range)(c, B, w32, w16, ws, dout, g32, c->N, 0, stream);
}

int CC_API(cc_mapper_bwd_range)(const cc_mapper_cfg* c, int32_t B, const float* w32, const uint16_t* w16, void* ws, const float* dout, float* g32,
                        int32_t l_hi, int32_t l_lo, void* stream) {
    if (!mapper_cfg_ok(c) || B <= 0 || !w32 || !w16 || !ws || !dout || !g32 || l_lo < 0 || l_hi > c->N || l_lo > l_hi) return CC_ERR_ARG;
    hipStream_t st = S_(stream);
    MapperOff o;
    mapper_offsets(c, o);
    MapperWS w;
    mapper_carve(c, B, 1, ws, w);
    X3_SCRATCH(w);
    const int D = c->D, PP = c->W * c->P, S = PP + c->L, M = B * S, H = c->H, hd = D / H, Hm = c->Hm;
    const int PD = c->P * D;
    const uint16_t* w16t = W16(w16, o.total);   // transposed weight copies: dgrad GEMMs are NT
    if (l_hi == c->N) {   // seed: d x[N][:, PP:, :] = dout, rows [0:PP] = 0
        if (hipMemsetAsync(w.dx32, 0, (size_t)M * D * sizeof(float), st) != hipSuccess) return CC_ERR_LAUNCH;
        CC_TRY(copy_rows(dout, (size_t)c->L * D, w.dx32 + (size_t)PP * D, (size_t)S * D, c->L * D, B, st));
        if (c->N > 0) CC_TRY(f32_to_act(w.dx32, w.gdx[c->N - 1], (size_t)M * D, st));      // (a mapper without layers has no 16-bit consumer)
    }
    // The weight gradients are off the dependency chain.  bf16 / fp16 builds (round 5): every layer of this call parks its four problems and
    // ONE grouped launch at the end runs them all — whole-K 256 x 256 tiles that add into dW in their epilogue (8 layers: 576 tiles), no K
    // slices, no slabs, no per-layer reduce launch; their operands live in per-layer buffers (MapperWS::gd*).  A caller that wants the
    // gradients of a layer range early (the DDP reducer's 2-layer slices) gets them at the end of ITS call: the flush is per call.
    // bf16x3 build: per layer, as before (operand images are split per GEMM call).
    WgradBatch wb;
    wb.defer = true;
    constexpr bool kDeferAll = !kX3;
    static const bool defer_all_on = []() { const char* e = cc_lab_env("CC_MAPPER_WGRAD_DEFER"); return !e || atoi(e) != 0; }();     // lab build: A/B switch
    // Whole-K tiles fill the CUs worse than the per-layer launches' K slices (8 layers: 576 tiles = 2.25 rounds); the launch therefore cuts the
    // tiles beyond its last full round into K slices (TTGroup::whole / split: 2 rounds + 64 tiles x 4 slices, slabs + a small reduce).  What the
    // deferred launch saves does not grow with the row count, what is left of the fill loss does: ahead at B = 256 (M = 5120: 2.63 -> 2.50 ms)
    // and B = 1024 (7.64 -> 7.43), behind at B = 4096 (28.04 -> 28.35) — calls above 40 960 rows keep the per-layer form.
    static const int defer_rows = []() { const char* e = cc_lab_env("CC_MAPPER_DEFER_ROWS"); return e ? std::min(atoi(e), kDeferRows) : kDeferRows; }();      // lab build: a lower row limit
    const bool defer_all = kDeferAll && defer_all_on && w.own_grads && 4 * (l_hi - l_lo) <= 32 && M <= defer_rows;
    if (defer_all) { wb.direct = true; wb.cap = 4 * (l_hi - l_lo); }
    ColsumBatch cs;
    // bf16x3: `G2(t, width)` = the tensor as both of its GEMMs take it — split ONCE into w.gimg ([hi | hi | lo], the form of the weight
    // gradient's first operand and of the input gradient's A operand alike); every use re-arms the one-shot image hint
#if CC_OP == 2
    static const bool share = []() { const char* e = cc_lab_env("CC_X3_SHARE"); return !e || atoi(e) != 0; }();
    int g2rc = CC_OK;
    auto G2 = [&](const act_t* t, int width) -> const act_t* {
        if (!share) return t;
        g2rc = x3_split_rows(t, (size_t)width, w.gimg, M, width, 0, st);
        return reinterpret_cast<const act_t*>(w.gimg);
    };
    auto USE = [&](const act_t* img) -> const act_t* { if (share) x3_expect_image(img); return img; };
#else
    auto G2 = [&](const act_t* t, int) -> const act_t* { return t; };
    auto USE = [&](const act_t* img) -> const act_t* { return img; };
    const int g2rc = CC_OK;
#endif
    for (int l = l_hi - 1; l >= l_lo; l--) {
        const auto& y = o.layer[l];
        act_t* dx16 = w.gdx[l];                                   // d x[l+1]: written by the seed / by the LN1 backward of layer l + 1
        act_t* dx16b = w.gdxb[l];
        act_t* dh16 = w.gdh[l];
        act_t* dqkv16 = w.gdqkv[l];
        // d x[l] in 16 bits: the gradient entering layer l - 1.  Below layer 0 nobody reads it (the linear's gradient comes from dx32), and it
        // must NOT land in a buffer a deferred weight gradient still has to read: it goes to the attention-gradient scratch, dead by then
        act_t* dx16_below = l > 0 ? w.gdx[l - 1] : w.datt16;
        // fc2: y = h W2^T + b2
        const act_t* gx = G2(dx16, D);
        CC_TRY(g2rc);
        CC_TIMED(CC_SITE_MAPPER_WGRAD_FC2, st, gemm_wgrad(USE(gx), D, w.h[l], Hm, D, Hm, M, g32 + y.w2, Hm, w.wg_scratch, st, &wb));
        // fc2.bias gradient = column sums of dx16: for every layer but the top one the LN1 backward of the layer above produced
        // them together with dx16 (ln_bwd dcol); the top layer's dx16 comes from the seed
        if (l == c->N - 1) CC_TRY(colsum_bf16(dx16, D, M, D, g32 + y.b2, st));
        CC_TRY(gemm_dact(0, 0, USE(gx), D, W16(w16t, y.w2), D, M, Hm, D, dh16, Hm, w.h[l], 1, st));          // W2^T [Hm, D]
        // fc1
        const act_t* gh = G2(dh16, Hm);
        CC_TRY(g2rc);
        CC_TRY(gemm_wgrad(USE(gh), Hm, w.xn2[l], D, Hm, D, M, g32 + y.w1, D, w.wg_scratch, st, &wb));
        if (defer_all) cs.add(dh16, g32 + y.b1);                  // fc1.bias gradient: with the deferred weight gradients, one launch for all layers
        else CC_TRY(colsum_bf16(dh16, Hm, M, Hm, g32 + y.b1, st));
        CC_TRY(gemm_bf16out(0, 0, USE(gh), Hm, W16(w16t, y.w1), Hm, M, D, Hm, w.dxn16, D, nullptr, 0, nullptr, st));   // W1^T [D, Hm]
        CC_TRY(ln_bwd(w.dxn16, w.x1[l], D, nullptr, w.mean2[l], w.rstd2[l], w32 + y.n2w, w.dx32, w.dx32, dx16b, g32 + y.n2w,
                      g32 + y.n2b, M, D, st, g32 + y.bp));         // + project.bias gradient (column sums of dx16b)
        // project
        const act_t* gb = G2(dx16b, D);
        CC_TRY(g2rc);
        CC_TRY(gemm_wgrad(USE(gb), D, w.att[l], D, D, D, M, g32 + y.wp, D, w.wg_scratch, st, &wb));
        CC_TRY(gemm_bf16out(0, 0, USE(gb), D, W16(w16t, y.wp), D, M, D, D, w.datt16, D, nullptr, 0, nullptr, st));     // Wp^T
        CC_TRY(attn_bwd(w.qkv[l], w.datt16, w.att[l], w.lse[l], w.adelta, B, S, H, hd, false, dqkv16, st));
        // fused q/kv projection (to_queries.weight ++ to_keys_values.weight = [3D, D])
        const act_t* gq = G2(dqkv16, 3 * D);
        CC_TRY(g2rc);
        CC_TRY(gemm_wgrad(USE(gq), 3 * D, w.xn1[l], D, 3 * D, D, M, g32 + y.wq, D, w.wg_scratch, st, &wb));
        CC_TRY(gemm_bf16out(0, 0, USE(gq), 3 * D, W16(w16t, y.wq), 3 * D, M, D, 3 * D, w.dxn16, D, nullptr, 0, nullptr, st));  // Wqkv^T [D, 3D]
        // per-layer form: the deferred weight gradients read dx16, dh16, dx16b, dqkv16 — run them before the LN1 backward overwrites dx16
        // with the next layer's input gradient (with per-layer buffers nothing is overwritten and the flush waits for the end of the call)
        if (!defer_all) CC_TRY(wgrad_flush(wb, st));
        CC_TRY(ln_bwd(w.dxn16, w.x[l], D, nullptr, w.mean1[l], w.rstd1[l], w32 + y.n1w, w.dx32, w.dx32, dx16_below, g32 + y.n1w,
                      g32 + y.n1b, M, D, st, l > 0 ? g32 + o.layer[l - 1].b2 : nullptr));   // + fc2.bias gradient of the layer below
    }
    CC_TRY(wgrad_flush(wb, st));
    CC_TRY(colsum_bf16_multi(cs, Hm, M, Hm, st));
    if (l_lo > 0) return CC_OK;
    // prefix_const, pos_embeddings, linear
    CC_TRY(batch_sum(w.dx32 + (size_t)PP * D, (size_t)S * D, g32 + o.prefix, c->L * D, B, st));
    if (o.pos >= 0) CC_TRY(batch_sum(w.dx32, (size_t)S * D, g32 + o.pos, PP * D, B, st));
    CC_TRY(slice_f32_to_bf16(w.dx32, (size_t)S * D, w.dlin16, (size_t)PP * D, PP * D, B, st));
    CC_TRY(gemm_wgrad(w.dlin16, PD, w.emb16, c->E, PD, c->E, B * c->W, g32 + o.lin_w, c->E, w.wg_scratch, st));
    CC_TRY(colsum_bf16(w.dlin16, PD, B * c->W, PD, g32 + o.lin_b, st));
    return CC_OK;
}

// ---------------------------------------------------------------- GPT-2 -----------------------------------------
namespace {
// Exponential form of the lm_head outputs (gemm.hip.h EpiLMHead): the bf16 build's training path stores exp(logit - target logit) and
// never materialises the softmax gradient.  fp16 lacks the exponent range, the bf16x3 build keeps fp32 logits.  CC_LM_EXPFORM=0: A/B switch.
// c_fc forward stores gelu_new'(u) where it used to store u (CC_GELU_GRAD_FWD=0: A/B switch; forward and backward read the same setting)
static bool gelu_grad_fwd() {
    static const bool on = []() { const char* e = cc_lab_env("CC_GELU_GRAD_FWD"); return !e || atoi(e) != 0; }();
    return on;
}
// bf16x3, frozen LM: c_fc's forward epilogue and the gelu' input-gradient epilogue write the [hi | hi | lo] operand image of their consumer
// GEMM directly instead of an fp32 activation that a split pass re-reads (CC_X3_IMG=0: A/B switch)
static bool x3_img_on() {
    static const bool on = []() { const char* e = cc_lab_env("CC_X3_IMG"); return !e || atoi(e) != 0; }();
    return on;
}
#if CC_OP == 2
static bool gpt2_bwd_images(const cc_gpt2_shape* s) { return s->mode == 1 && x3_img_on() && s->p_resid == 0.f && s->p_attn == 0.f && s->p_embd == 0.f; }
#endif
// bf16x3: the exponential form for frozen-LM runs (round 4) — E leaves the GEMM as the [hi | hi | lo] operand image of the input-gradient
// GEMM (in the logits buffer, 1.5x), so neither fp32 logits nor the softmax-gradient pass over them exist; the full finetune keeps the
// logit form (its tied weight gradient reads the fp32 gradient).
static bool lm_exp_form(const cc_gpt2_shape* s) {
    static const bool env = []() { const char* e = cc_lab_env("CC_LM_EXPFORM"); return !e || atoi(e) != 0; }();
    if (CC_OP == 0) return env;
    if (CC_OP == 2) return env && x3_img_on() && s->mode == 1;
    return false;
}
static bool shape_ok(const cc_gpt2_cfg* c, const cc_gpt2_shape* s) {
    return s && s->B > 0 && s->T > 0 && s->L >= 0 && s->L <= s->T && s->cap >= s->T - s->L && s->mode >= 0 && s->mode <= 2 && s->T <= c->NPOS &&
           s->p_embd >= 0.f && s->p_embd < 1.f && s->p_attn >= 0.f && s->p_attn < 1.f && s->p_resid >= 0.f && s->p_resid < 1.f;
}
}  // namespace

int64_t CC_API(cc_gpt2_param_count)(const cc_gpt2_cfg* cfg) {
    if (!gpt2_cfg_ok(cfg)) return CC_ERR_SHAPE;
    Gpt2Off o;
    gpt2_offsets(cfg, o);
    return o.total;
}

int CC_API(cc_gpt2_param_offsets)(const cc_gpt2_cfg* cfg, int64_t* offs) {
    if (!gpt2_cfg_ok(cfg) || !offs) return CC_ERR_SHAPE;
    Gpt2Off o;
    gpt2_offsets(cfg, o);
    int k = 0;
    offs[k++] = o.wte; offs[k++] = o.wpe;
    for (int l = 0; l < cfg->NL; l++) {
        const auto& y = o.layer[l];
        const int64_t v[12] = {y.l1w, y.l1b, y.aw, y.ab, y.pw, y.pb, y.l2w, y.l2b, y.fw, y.fb, y.p2w, y.p2b};
        for (int i = 0; i < 12; i++) offs[k++] = v[i];
    }
    offs[k++] = o.lnf_w; offs[k++] = o.lnf_b;
    return CC_OK;
}

int64_t CC_API(cc_gpt2_ws_bytes)(const cc_gpt2_cfg* cfg, const cc_gpt2_shape* s) {
    if (!gpt2_cfg_ok(cfg) || !shape_ok(cfg, s)) return CC_ERR_SHAPE;
    Gpt2WS w;
    gpt2_carve(cfg, s->B, s->T, s->T - s->L, s->mode, nullptr, w);
    return (int64_t)w.bytes;
}

#if CC_OP == 2
static int gpt2_sync_x3(const cc_gpt2_cfg* c, const float* w32, uint16_t* w16, hipStream_t st) {
    Gpt2Off o;
    gpt2_offsets(c, o);
    const int D = c->D;
    X3SplitBatch sb;
    sb.add(w32 + o.wte, W16(w16, o.wte), c->Vp, D, 0, 1);                       // lm_head forward / logits / decode
    sb.add(w32 + o.wte, W16(w16, o.total + o.wte), c->Vp, D, 1, 1);             // [D][3 Vp]: lm_head input gradient
    CC_TRY(x3_split_multi(sb, st));
    sb.n = 0;
    for (int l = 0; l < c->NL; l++) {
        const auto& y = o.layer[l];
        const int64_t off[4] = {y.aw, y.pw, y.fw, y.p2w};
        const int R[4] = {D, D, D, 4 * D}, C[4] = {3 * D, D, 4 * D, D};          // Conv1D [in][out]
        for (int i = 0; i < 4; i++) {
            if (sb.n + 2 > 32) { CC_TRY(x3_split_multi(sb, st)); sb.n = 0; }
            sb.add(w32 + off[i], W16(w16, off[i]), R[i], C[i], 0, 1);
            sb.add(w32 + off[i], W16(w16, o.total + off[i]), R[i], C[i], 1, 1);
        }
    }
    return x3_split_multi(sb, st);
}
#endif

static int gpt2_transposes(const cc_gpt2_cfg* c, uint16_t* w16, hipStream_t st) {
    if (kX3) return CC_ERR_ARG;      // the operand images are made from the fp32 master (cc_gpt2_sync_weights)
    Gpt2Off o;
    gpt2_offsets(c, o);
    op16_t* t = w16 + o.total;
    const int D = c->D;
    CC_TRY(transpose_bf16(w16 + o.wte, t + o.wte, c->Vp, D, st));     // [Vp, D] -> [D, Vp]  (lm_head dgrad)
    TransposeBatch tb;
    for (int l = 0; l < c->NL; l++) {
        const auto& y = o.layer[l];
        tb.add(w16 + y.aw, t + y.aw, D, 3 * D);    // Conv1D [in,out] -> [out,in]: forward is NT on these
        tb.add(w16 + y.pw, t + y.pw, D, D);
        tb.add(w16 + y.fw, t + y.fw, D, 4 * D);
        tb.add(w16 + y.p2w, t + y.p2w, 4 * D, D);
        if (tb.n == 32) { CC_TRY(transpose_bf16_multi(tb, st)); tb.n = 0; }
    }
    CC_TRY(transpose_bf16_multi(tb, st));
    return CC_OK;
}

int CC_API(cc_gpt2_sync_weights)(const cc_gpt2_cfg* c, const float* w32, uint16_t* w16, void* stream) {
    if (!gpt2_cfg_ok(c) || !w32 || !w16) return CC_ERR_ARG;
    hipStream_t st = S_(stream);
#if CC_OP == 2
    return gpt2_sync_x3(c, w32, w16, st);
#else
    Gpt2Off o;
    gpt2_offsets(c, o);
    CC_TRY(f32_to_bf16(w32, w16, (size_t)o.total, st));
    return gpt2_transposes(c, w16, st);
#endif
}

int CC_API(cc_gpt2_transpose_weights)(const cc_gpt2_cfg* c, uint16_t* w16, void* stream) {
    if (!gpt2_cfg_ok(c) || !w16) return CC_ERR_ARG;
    return gpt2_transposes(c, w16, S_(stream));
}

// GPT-2 dropout (full finetune in train mode) is part of the pass's cc_gpt2_shape: embed / fwd / bwd(_range) of one pass read the
// same (p_embd, p_attn, p_resid, drop_seed); all zero = eval behaviour.
int CC_API(cc_dropout_mask)(uint64_t seed, int32_t site, int32_t layer, float p, int64_t n, uint8_t* out, void* stream) {
    if (!out || n < 0 || site < 0 || site > 3 || layer < 0 || layer > 255 || p < 0.f || p >= 1.f) return CC_ERR_ARG;
    return dropout_mask_u8(out, (size_t)n, make_drop(p, seed, (unsigned)site, (unsigned)layer), S_(stream));
}

int CC_API(cc_gpt2_embed)(const cc_gpt2_cfg* c, const cc_gpt2_shape* s, const float* w32, const float* prefix, const int64_t* tokens, void* ws,
                  void* stream) {
    if (!gpt2_cfg_ok(c) || !shape_ok(c, s) || !w32 || !ws || (s->L > 0 && !prefix) || (s->T > s->L && !tokens)) return CC_ERR_ARG;
    Gpt2Off o;
    gpt2_offsets(c, o);
    Gpt2WS w;
    gpt2_carve(c, s->B, s->T, s->T - s->L, s->mode, ws, w);
    CC_TRY(embed_concat(prefix, reinterpret_cast<const long long*>(tokens), s->cap, w32 + o.wte, w32 + o.wpe, w.x[0], s->B, s->L, s->T,
                        c->D, 0, S_(stream)));
    // embd dropout on inputs_embeds + position_embeds (hf GPT2Model.forward: self.drop)
    return dropout_f32(w.x[0], (size_t)s->B * s->T * c->D, make_drop(s->p_embd, s->drop_seed, DROP_EMBD, 0), S_(stream));
}

int CC_API(cc_gpt2_embed_from)(const cc_gpt2_cfg* c, const cc_gpt2_shape* s, const float* w32, const float* inputs_embeds, void* ws,
                       void* stream) {
    if (!gpt2_cfg_ok(c) || !shape_ok(c, s) || !w32 || !ws || !inputs_embeds) return CC_ERR_ARG;
    Gpt2Off o;
    gpt2_offsets(c, o);
    Gpt2WS w;
    gpt2_carve(c, s->B, s->T, s->T - s->L, s->mode, ws, w);
    return embed_concat(inputs_embeds, nullptr, 0, w32 + o.wte, w32 + o.wpe, w.x[0], s->B, s->T, s->T, c->D, 0, S_(stream));
}

int CC_API(cc_gpt2_fwd)(const cc_gpt2_cfg* c, const cc_gpt2_shape* s, const float* w32, const uint16_t* w16, void* ws, void* stream) {
    if (!gpt2_cfg_ok(c) || !shape_ok(c, s) || !w32 || !w16 || !ws) return CC_ERR_ARG;
    hipStream_t st = S_(stream);
    Gpt2Off o;
    gpt2_offsets(c, o);
    Gpt2WS w;
    gpt2_carve(c, s->B, s->T, s->T - s->L, s->mode, ws, w);
    X3_SCRATCH(w);
    const int D = c->D, M = s->B * s->T, H = c->H, hd = D / H;
    const uint16_t* w16t = W16(w16, o.total);   // Conv1D weights transposed to [out,in]: every forward GEMM is NT
    for (int l = 0; l < c->NL; l++) {
        const auto& y = o.layer[l];
        // hf :262-310: x1 = x + c_proj(attn(c_attn(ln_1 x)))
#if CC_OP == 2
        const bool ximg = s->mode <= 1 && x3_img_on();          // no weight gradient reads the normalised rows: LayerNorm writes the GEMM operand image
        if (ximg) x3_emit_image(w.xn1[l], D);
#endif
        CC_TRY(ln_fwd(w.x[l], D, nullptr, w32 + y.l1w, w32 + y.l1b, w.xn1[l], nullptr, w.mean1[l], w.rstd1[l], M, D, st));
#if CC_OP == 2
        if (ximg) x3_expect_image(w.xn1[l]);
#endif
        CC_TRY(gemm_bf16out(0, 0, w.xn1[l], D, W16(w16t, y.aw), D, M, 3 * D, D, w.qkv[l], 3 * D, w32 + y.ab, 0, nullptr, st));
#if CC_OP == 2
        const bool aimg = ximg && attn_fwd_can_image(s->T, hd);      // the fp32-VALU attention pair: the backward recomputes what it needs from qkv
        if (aimg) x3_emit_image(w.att[l], D);
#endif
        CC_TRY(attn_fwd(w.qkv[l], s->B, s->T, H, hd, true, w.att[l], w.lse[l], st, make_drop(s->p_attn, s->drop_seed, DROP_ATTN, l)));
        {
            static const int tile_proj = env_tile("CC_TILE_PROJ");
            TileScope ts(tile_proj);
#if CC_OP == 2
            if (aimg) x3_expect_image(w.att[l]);
#endif
            CC_TRY(gemm_resid(0, 0, w.att[l], D, W16(w16t, y.pw), D, M, D, D, w.x1[l], w.x[l], D, w32 + y.pb, st,
                              make_drop(s->p_resid, s->drop_seed, DROP_RESID_ATTN, l)));
        }
        // x = x1 + c_proj(gelu_new(c_fc(ln_2 x1)))   (hf :229-243)
#if CC_OP == 2
        if (ximg) x3_emit_image(w.xn2[l], D);
#endif
        CC_TRY(ln_fwd(w.x1[l], D, nullptr, w32 + y.l2w, w32 + y.l2b, w.xn2[l], nullptr, w.mean2[l], w.rstd2[l], M, D, st));
        {
            static const int tile_fc = env_tile("CC_TILE_FC");
            TileScope ts(tile_fc);
            // act 3: the pre-activation slot receives gelu_new'(u) — one sigmoid serves both, and the backward's epilogue is a multiply
#if CC_OP == 2
            if (ximg) { x3_emit_image(w.hact[l], 4 * D); x3_expect_image(w.xn2[l]); }   // nobody but mlp.c_proj reads hact without a weight gradient: write its operand image directly
#endif
            CC_TIMED(CC_SITE_GPT2_FC_FWD, st, gemm_bf16out(0, 0, w.xn2[l], D, W16(w16t, y.fw), D, M, 4 * D, D, w.hact[l], 4 * D, w32 + y.fb, (s->mode >= 1 && gelu_grad_fwd()) ? 3 : 2,
                                                            s->mode >= 1 ? w.u[l] : nullptr, st));
        }
        {
            static const int tile_proj2 = env_tile("CC_TILE_PROJ2");
            TileScope ts(tile_proj2);
#if CC_OP == 2
            if (s->mode <= 1 && x3_img_on()) x3_expect_image(w.hact[l]);
#endif
            CC_TIMED(CC_SITE_GPT2_PROJ2_FWD, st, gemm_resid(0, 0, w.hact[l], 4 * D, W16(w16t, y.p2w), 4 * D, M, D, 4 * D, w.x[l + 1], w.x1[l], D, w32 + y.p2b, st,
                                                           make_drop(s->p_resid, s->drop_seed, DROP_RESID_MLP, l)));
        }
    }
    return CC_OK;
}

int CC_API(cc_gpt2_logits)(const cc_gpt2_cfg* c, const cc_gpt2_shape* s, const float* w32, const uint16_t* w16, void* ws, float* logits,
                   int64_t ldl, void* stream) {
    if (!gpt2_cfg_ok(c) || !shape_ok(c, s) || !w32 || !w16 || !ws || !logits) return CC_ERR_ARG;
    const int Ns = std::min(c->Vp, rup(c->V, 8));
    if (ldl < Ns || (ldl & 3) || ldl > 0x7fffffff) return CC_ERR_SHAPE;
    hipStream_t st = S_(stream);
    Gpt2Off o;
    gpt2_offsets(c, o);
    Gpt2WS w;
    gpt2_carve(c, s->B, s->T, s->T - s->L, s->mode, ws, w);
    X3_SCRATCH(w);
    const int D = c->D, M = s->B * s->T;
    CC_TRY(ln_fwd(w.x[c->NL], D, nullptr, w32 + o.lnf_w, w32 + o.lnf_b, w.hf16, nullptr, w.meanf, w.rstdf, M, D, st));
    return gemm_f32out(0, 0, w.hf16, D, W16(w16, o.wte), D, M, Ns, D, logits, (int)ldl, nullptr, 0, 1.0f, 1, st);
}

int CC_API(cc_lmhead_ce_fwd)(const cc_gpt2_cfg* c, const cc_gpt2_shape* s, const float* w32, const uint16_t* w16, void* ws, const int64_t* tokens,
                     float* stats, void* stream) {
    if (!gpt2_cfg_ok(c) || !shape_ok(c, s) || s->mode < 1 || s->L < 1 || !w32 || !w16 || !ws || !tokens || !stats) return CC_ERR_ARG;
    hipStream_t st = S_(stream);
    Gpt2Off o;
    gpt2_offsets(c, o);
    Gpt2WS w;
    const int cap = s->T - s->L;
    gpt2_carve(c, s->B, s->T, cap, s->mode, ws, w);
    X3_SCRATCH(w);
    const int D = c->D, Mc = s->B * cap, npart = c->Vp / 64;
    if (cap != s->cap) return CC_ERR_SHAPE;  // the loss consumes every token column (model.py:108-109)
    if (hipMemsetAsync(stats, 0, 2 * sizeof(float), st) != hipSuccess) return CC_ERR_LAUNCH;
    CC_TRY(ce_targets(reinterpret_cast<const long long*>(tokens), w.target, w.row_map, s->B, cap, s->L, s->T, st));
    // ln_f only on the rows the loss reads: L-1 .. T-2 of every sample (model.py:108)
    CC_TRY(ln_fwd(w.x[c->NL], D, w.row_map, w32 + o.lnf_w, w32 + o.lnf_b, w.hf16, nullptr, w.meanf, w.rstdf, Mc, D, st));
    const bool ef = lm_exp_form(s);
    const op16_t* wte_rows = kX3 ? reinterpret_cast<const op16_t*>(w32 + o.wte) : W16(w16, o.wte);      // rows read elementwise: bf16x3 takes the fp32 master
    if (ef) CC_TRY(lm_tgt_ref(w.hf16, wte_rows, D, w.target, w.cref, Mc, st));
#if CC_OP == 2
    if (ef) x3_emit_image(w.logits16, c->Vp);
#endif
    CC_TIMED(CC_SITE_LMHEAD_FWD, st, gemm_lmhead(w.hf16, D, W16(w16, o.wte), D, Mc, c->Vp, c->V, D, w.logits16, c->Vp, w.pmax, w.psum, npart, w.target, w.tgt_logit, st,
                                                 ef ? w.cref : nullptr));
    CC_TRY(ce_rows(w.pmax, w.psum, npart, w.target, ef ? w.cref : w.tgt_logit, w.lse_row, w.row_loss, stats, Mc, st));   // cref IS the target logit
    return CC_OK;
}

int CC_API(cc_lmhead_ce_bwd)(const cc_gpt2_cfg* c, const cc_gpt2_shape* s, const float* w32, const uint16_t* w16, void* ws, const float* denom,
                     const float* loss_scale, float* g32, void* stream) {
    if (!gpt2_cfg_ok(c) || !shape_ok(c, s) || s->mode < 1 || !w32 || !w16 || !ws || !denom || (s->mode == 2 && !g32)) return CC_ERR_ARG;
    hipStream_t st = S_(stream);
    Gpt2Off o;
    gpt2_offsets(c, o);
    Gpt2WS w;
    const int cap = s->T - s->L;
    gpt2_carve(c, s->B, s->T, cap, s->mode, ws, w);
    X3_SCRATCH(w);
    const int D = c->D, Mc = s->B * cap, M = s->B * s->T;
    const bool full = s->mode == 2;
    // exponential form: logits16 holds E = exp(logit - cref); d logits = r E - w onehot is never written — the row factors go into the
    // GEMMs' finishing passes (EpiLMHead comment).  Otherwise: the in-place softmax-gradient pass over the stored logits.
    const bool ef = lm_exp_form(s);
    const LmFix fix{w.lmfac, w.target, kX3 ? reinterpret_cast<const op16_t*>(w32 + o.wte) : W16(w16, o.wte)};
    if (ef) CC_TRY(lm_rowfac(w.cref, w.lse_row, w.target, denom, loss_scale, w.lmfac, Mc, st));
    const act_t* dlog = w.logits16;            // the A operand of the input-gradient GEMM
#if CC_OP == 2
    // frozen LM: nothing but that GEMM reads d logits -> the softmax-gradient pass writes its [hi | hi | lo] operand image straight into the
    // call's operand scratch (sized for exactly this image, gpt2_carve) instead of fp32 values a split pass would re-read
    op16_t* dimg = (!ef && !full && x3_img_on()) ? x3_scratch_block(x3_img(Mc, c->Vp)) : nullptr;
    if (ef) dimg = reinterpret_cast<op16_t*>(w.logits16);        // exponential form: the forward GEMM wrote E there as the operand image
    if (dimg) dlog = reinterpret_cast<const act_t*>(dimg);
    if (!ef) CC_TRY(ce_dlogits(w.logits16, c->Vp, c->V, w.target, w.lse_row, denom, loss_scale, Mc, st, dimg));
#else
    if (!ef) CC_TRY(ce_dlogits(w.logits16, c->Vp, c->V, w.target, w.lse_row, denom, loss_scale, Mc, st));
#endif
    // d hf = dlogits · wte   ([Mc,Vp] x [Vp(k), D(n)])
    // K = Vp is deep and the output narrow: K slices over the idle CUs, slabs parked in du16 (free until the first layer's backward)
    const auto lm_dgrad = [&]() {
#if CC_OP == 2
        if (dimg) x3_expect_image(dlog);
#endif
        const int rc = gemm_nt_deepk(dlog, c->Vp, W16(w16, o.total + o.wte), c->Vp, Mc, D, c->Vp, w.dhf16, D, reinterpret_cast<float*>(w.du16),
                                     (size_t)M * 4 * D * sizeof(act_t), st, ef ? &fix : nullptr);
        if (rc != CC_ERR_SHAPE) return rc;
#if CC_OP == 2
        if (dimg) x3_expect_image(dlog);
#endif
        const int rc2 = gemm_bf16out(0, 0, dlog, c->Vp, W16(w16, o.total + o.wte), c->Vp, Mc, D, c->Vp, w.dhf16, D, nullptr, 0, nullptr, st);
        return (rc2 != CC_OK || !ef) ? rc2 : lm_dgrad_fix(w.dhf16, w.lmfac, w.target, fix.wte, D, Mc, st);
    };
    CC_TIMED(CC_SITE_LMHEAD_DGRAD, st, lm_dgrad());
    if (full) {      // tied lm_head: d wte += dlogits^T hf  (= E^T (r hf) - onehot^T (w hf) in the exponential form)
        if (ef) {
            CC_TRY(lm_scale_rows(w.hf16, w.lmfac, w.hfs16, D, Mc, st));
            CC_TRY(gemm_wgrad(w.logits16, c->Vp, w.hfs16, D, c->Vp, D, Mc, g32 + o.wte, D, w.wg_scratch, st));
            CC_TRY(lm_wgrad_onehot(w.hf16, w.lmfac, w.target, g32 + o.wte, D, Mc, st));
        } else {
            CC_TRY(gemm_wgrad(w.logits16, c->Vp, w.hf16, D, c->Vp, D, Mc, g32 + o.wte, D, w.wg_scratch, st));
        }
    }
    if (hipMemsetAsync(w.dx32, 0, (size_t)M * D * sizeof(float), st) != hipSuccess) return CC_ERR_LAUNCH;
    size_t dx16_bytes = (size_t)M * D * sizeof(act_t);
#if CC_OP == 2
    if (gpt2_bwd_images(s)) { dx16_bytes = (size_t)M * D * 3 * sizeof(op16_t); x3_emit_image(w.dx16, D); }     // the top layer's backward reads it as an operand image
#endif
    if (hipMemsetAsync(w.dx16, 0, dx16_bytes, st) != hipSuccess) return CC_ERR_LAUNCH;
    CC_TRY(ln_bwd(w.dhf16, w.x[c->NL], D, w.row_map, w.meanf, w.rstdf, w32 + o.lnf_w, nullptr, w.dx32, w.dx16, full ? g32 + o.lnf_w : nullptr,
                  full ? g32 + o.lnf_b : nullptr, Mc, D, st));
    return CC_OK;
}

int CC_API(cc_gpt2_logits_bwd)(const cc_gpt2_cfg* c, const cc_gpt2_shape* s, const float* w32, const uint16_t* w16, void* ws, const float* dlogits,
                       int64_t ldl, float* dx0, float* g32, void* stream) {
    if (!gpt2_cfg_ok(c) || !shape_ok(c, s) || s->mode < 1 || s->L != 0 || !w32 || !w16 || !ws || !dlogits || ldl < c->V || (s->mode == 2 && !g32))
        return CC_ERR_ARG;
    hipStream_t st = S_(stream);
    Gpt2Off o;
    gpt2_offsets(c, o);
    Gpt2WS w;
    gpt2_carve(c, s->B, s->T, s->T, s->mode, ws, w);      // L == 0: the loss-side buffers are sized for all B*T rows
    X3_SCRATCH(w);
    const int D = c->D, M = s->B * s->T;
    const bool full = s->mode == 2;
    CC_TRY(f32_to_op16_pad(dlogits, ldl, c->V, w.logits16, c->Vp, M, st));
    // d hf = dlogits · wte ([M,Vp] x [Vp(k), D(n)]); tied lm_head: d wte += dlogits^T hf (hf16 = the rows cc_gpt2_logits normalised)
    CC_TRY(gemm_bf16out(0, 0, w.logits16, c->Vp, W16(w16, o.total + o.wte), c->Vp, M, D, c->Vp, w.dhf16, D, nullptr, 0, nullptr, st));
    if (full) CC_TRY(gemm_wgrad(w.logits16, c->Vp, w.hf16, D, c->Vp, D, M, g32 + o.wte, D, w.wg_scratch, st));
    if (hipMemsetAsync(w.dx32, 0, (size_t)M * D * sizeof(float), st) != hipSuccess) return CC_ERR_LAUNCH;
#if CC_OP == 2
    if (gpt2_bwd_images(s)) x3_emit_image(w.dx16, D);
#endif
    CC_TRY(ln_bwd(w.dhf16, w.x[c->NL], D, nullptr, w.meanf, w.rstdf, w32 + o.lnf_w, nullptr, w.dx32, w.dx16, full ? g32 + o.lnf_w : nullptr,
                  full ? g32 + o.lnf_b : nullptr, M, D, st));
    CC_TRY(CC_API(cc_gpt2_bwd_range)(c, s, w32, w16, ws, nullptr, nullptr, g32, c->NL, 0, stream));
    if (dx0 && hipMemcpyAsync(dx0, w.dx32, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return CC_ERR_LAUNCH;
    return CC_OK;
}

int CC_API(cc_gpt2_bwd)(const cc_gpt2_cfg* c, const cc_gpt2_shape* s, const float* w32, const uint16_t* w16, void* ws, const int64_t* tokens,
                float* dprefix, float* g32, void* stream) {
    if (!gpt2_cfg_ok(c)) return CC_ERR_ARG;
    return CC_API(cc_gpt2_bwd_range)(c, s, w32, w16, ws, tokens, dprefix, g32, c->NL, 0, stream);
}

int CC_API(cc_gpt2_bwd_range)(const cc_gpt2_cfg* c, const cc_gpt2_shape* s, const float* w32, const uint16_t* w16, void* ws, const int64_t* tokens,
                      float* dprefix, float* g32, int32_t l_hi, int32_t l_lo, void* stream) {
    if (!gpt2_cfg_ok(c) || !shape_ok(c, s) || s->mode < 1 || !w32 || !w16 || !ws || (s->L > 0 && !dprefix) || (s->mode == 2 && !g32) ||
        l_lo < 0 || l_hi > c->NL || l_lo > l_hi)
        return CC_ERR_ARG;
    hipStream_t st = S_(stream);
    Gpt2Off o;
    gpt2_offsets(c, o);
    Gpt2WS w;
    gpt2_carve(c, s->B, s->T, s->T - s->L, s->mode, ws, w);
    X3_SCRATCH(w);
    const int D = c->D, M = s->B * s->T, H = c->H, hd = D / H, D3 = 3 * D, D4 = 4 * D;
    const bool full = s->mode == 2;
    WgradBatch wb;          // full finetune: a layer's four weight gradients as one grouped launch + one slab reduce
    wb.defer = full;
    WgradBatch* wbp = full ? &wb : nullptr;
    // The 16-bit copy of the residual gradient that feeds a c_proj's backward GEMMs is the dropout-masked one, and the c_proj's bias
    // gradient is its column sum.  Both are produced by the LayerNorm backward that writes the copy (ln_bwd dmask / dcol) — except for
    // the top layer, whose copy comes from ln_f's row-mapped backward and is masked / summed by separate launches.
#if CC_OP == 2
    // frozen LM, no dropout: the 16-bit gradient copies are read by input-gradient GEMMs only -> their producers write operand images
    const bool bimg = gpt2_bwd_images(s);
#endif
    for (int l = l_hi - 1; l >= l_lo; l--) {
        const auto& y = o.layer[l];
        const bool top = l == c->NL - 1;
        // mlp.c_proj (Conv1D [4D, D]): y = hact W + b
        if (top) CC_TRY(dropout_bf16(w.dx16, (size_t)M * D, make_drop(s->p_resid, s->drop_seed, DROP_RESID_MLP, l), st));
        if (full) {
            CC_TRY(gemm_wgrad(w.hact[l], D4, w.dx16, D, D4, D, M, g32 + y.p2w, D, w.wg_scratch, st, wbp));
            if (top) CC_TRY(colsum_bf16(w.dx16, D, M, D, g32 + y.p2b, st));
        }
        {
            static const int tile_dact = env_tile("CC_TILE_DACT");
            TileScope ts(tile_dact);
#if CC_OP == 2
            if (!full && x3_img_on()) x3_emit_image(w.du16, D4);                     // frozen LM: du is read by c_fc's input-gradient GEMM only
#endif
#if CC_OP == 2
            if (bimg) x3_expect_image(w.dx16);                        // written as an image by the LayerNorm backward above it (or the lm_head's)
#endif
            CC_TRY(gemm_dact(0, 0, w.dx16, D, W16(w16, y.p2w), D, M, D4, D, w.du16, D4, w.u[l], gelu_grad_fwd() ? 3 : 2, st));
        }
        // mlp.c_fc (Conv1D [D, 4D])
        if (full) {
            CC_TRY(gemm_wgrad(w.xn2[l], D, w.du16, D4, D, D4, M, g32 + y.fw, D4, w.wg_scratch, st, wbp));
            CC_TRY(colsum_bf16(w.du16, D4, M, D4, g32 + y.fb, st));
        }
#if CC_OP == 2
        if (!full && x3_img_on()) x3_expect_image(w.du16);
#endif
        CC_TIMED(CC_SITE_GPT2_FC_DGRAD, st, gemm_bf16out(0, 0, w.du16, D4, W16(w16, y.fw), D4, M, D, D4, w.dxn16, D, nullptr, 0, nullptr, st));
#if CC_OP == 2
        if (bimg) x3_emit_image(w.dx16b, D);
#endif
        CC_TRY(ln_bwd(w.dxn16, w.x1[l], D, nullptr, w.mean2[l], w.rstd2[l], w32 + y.l2w, w.dx32, w.dx32, w.dx16b, full ? g32 + y.l2w : nullptr,
                      full ? g32 + y.l2b : nullptr, M, D, st, full ? g32 + y.pb : nullptr,
                      make_drop(s->p_resid, s->drop_seed, DROP_RESID_ATTN, l)));
        // attn.c_proj (Conv1D [D, D])
        if (full) CC_TRY(gemm_wgrad(w.att[l], D, w.dx16b, D, D, D, M, g32 + y.pw, D, w.wg_scratch, st, wbp));
#if CC_OP == 2
        if (bimg) x3_expect_image(w.dx16b);
#endif
        CC_TRY(gemm_bf16out(0, 0, w.dx16b, D, W16(w16, y.pw), D, M, D, D, w.datt16, D, nullptr, 0, nullptr, st));
#if CC_OP == 2
        const bool qimg = bimg && attn_bwd_can_image(s->T, hd);
        if (qimg) x3_emit_image(w.dqkv16, D3);
#endif
        CC_TRY(attn_bwd(w.qkv[l], w.datt16, w.att[l], w.lse[l], w.adelta, s->B, s->T, H, hd, true, w.dqkv16, st,
                        make_drop(s->p_attn, s->drop_seed, DROP_ATTN, l)));
        // attn.c_attn (Conv1D [D, 3D])
        if (full) {
            CC_TRY(gemm_wgrad(w.xn1[l], D, w.dqkv16, D3, D, D3, M, g32 + y.aw, D3, w.wg_scratch, st, wbp));
            CC_TRY(colsum_bf16(w.dqkv16, D3, M, D3, g32 + y.ab, st));
        }
#if CC_OP == 2
        if (qimg) x3_expect_image(w.dqkv16);
#endif
        CC_TRY(gemm_bf16out(0, 0, w.dqkv16, D3, W16(w16, y.aw), D3, M, D, D3, w.dxn16, D, nullptr, 0, nullptr, st));
        // deferred weight gradients: dx16 (masked layer-input gradient), du16, dx16b, dqkv16 are all still intact here
        if (full) CC_TRY(wgrad_flush(wb, st));
#if CC_OP == 2
        if (bimg) x3_emit_image(w.dx16, D);
#endif
        CC_TRY(ln_bwd(w.dxn16, w.x[l], D, nullptr, w.mean1[l], w.rstd1[l], w32 + y.l1w, w.dx32, w.dx32, w.dx16, full ? g32 + y.l1w : nullptr,
                      full ? g32 + y.l1b : nullptr, M, D, st, (full && l > 0) ? g32 + o.layer[l - 1].p2b : nullptr,
                      l > 0 ? make_drop(s->p_resid, s->drop_seed, DROP_RESID_MLP, l - 1) : Drop()));
    }
    if (l_lo > 0) return CC_OK;
    CC_TRY(dropout_f32(w.dx32, (size_t)M * D, make_drop(s->p_embd, s->drop_seed, DROP_EMBD, 0), st));   // d(inputs + wpe)
    if (s->L > 0) CC_TRY(copy_rows(w.dx32, (size_t)s->T * D, dprefix, (size_t)s->L * D, s->L * D, s->B, st));
    if (full)
        CC_TRY(embed_bwd(w.dx32, reinterpret_cast<const long long*>(tokens), s->cap, g32 + o.wte, g32 + o.wpe, s->B, s->L, s->T, D, st));
    return CC_OK;
}

// ---------------------------------------------------------------- optimizer / casts / test hooks -----------------
int CC_API(cc_adamw_step)(float* p32, const float* g32, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, float grad_scale, const float* loss_scale, const float* found_inf, void* stream) {
    if (!p32 || !g32 || !m || !v || n < 0 || step < 0 || (step == 0 && !loss_scale)) return CC_ERR_ARG;
    return adamw(p32, g32, m, v, (size_t)n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, loss_scale, found_inf, S_(stream));
}

int CC_API(cc_adamw_step_cast)(float* p32, const float* g32, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, int32_t step, float grad_scale, const float* loss_scale, const float* found_inf, uint16_t* w16, void* stream) {
    if (!p32 || !g32 || !m || !v || !w16 || n < 0 || step < 0 || (step == 0 && !loss_scale)) return CC_ERR_ARG;
    if (kX3) return CC_ERR_ARG;      // bf16x3: the operand images are per-matrix row images, not a flat cast (use cc_adamw_step + cc_*_sync_weights)
    return adamw(p32, g32, m, v, (size_t)n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, loss_scale, found_inf, S_(stream), w16);
}

int CC_API(cc_cast_op16)(const float* src, uint16_t* dst, int64_t n, void* stream) {
    if (!src || !dst || n < 0 || kX3) return CC_ERR_ARG;
    return f32_to_bf16(src, dst, (size_t)n, S_(stream));
}

int CC_API(cc_grad_wire_pack)(const float* g32, uint16_t* wire, int64_t n, void* stream) {
    if (!g32 || !wire || n < 0) return CC_ERR_ARG;
    return wire_pack(g32, wire, (size_t)n, S_(stream));
}

int CC_API(cc_grad_wire_unpack)(const uint16_t* wire, float* g32, int64_t n, void* stream) {
    if (!g32 || !wire || n < 0) return CC_ERR_ARG;
    return wire_unpack(wire, g32, (size_t)n, S_(stream));
}

int CC_API(cc_grad_nonfinite)(const float* g32, int64_t n, float* found_inf, void* stream) {
    if (!g32 || !found_inf || n < 0) return CC_ERR_ARG;
    return grad_nonfinite(g32, (size_t)n, found_inf, S_(stream));
}

int CC_API(cc_loss_scale_update)(float* state, float* found_inf, float growth, float backoff, int32_t interval, void* stream) {
    if (!state || !found_inf || growth < 1.f || backoff <= 0.f || backoff > 1.f || interval < 1) return CC_ERR_ARG;
    return loss_scale_update(state, found_inf, growth, backoff, interval, S_(stream));
}

int CC_API(cc_gemm_op16_f32)(int32_t al, int32_t bl, const uint16_t* A, int32_t lda, const uint16_t* B, int32_t ldb, int32_t M, int32_t N, int32_t K,
                     float* C, int32_t ldc, const float* bias, int32_t ksplit, void* stream) {
    if (!A || !B || !C || kX3) return CC_ERR_ARG;      // (bf16x3 GEMMs need a workspace for their operand images: no bare hook)
    return gemm_f32out(al, bl, reinterpret_cast<const act_t*>(A), lda, B, ldb, M, N, K, C, ldc, ksplit > 1 ? nullptr : bias, ksplit > 1 ? 2 : 0, 1.0f, ksplit, S_(stream));
}

int CC_API(cc_sample_step)(const float* logits, int32_t R, int32_t V, int32_t ld, float temperature, int32_t top_k, float top_p, int32_t mode,
                   const int64_t* history, int32_t hist_len, int32_t hist_ld, float repetition_penalty, const float* u, int32_t* next_token,
                   float* probs_out, void* stream) {
    if (!logits || !u || !next_token || R < 0) return CC_ERR_ARG;
    return sample_rows(logits, R, V, ld, temperature, top_k, top_p, mode, reinterpret_cast<const long long*>(history), hist_len, hist_ld,
                       repetition_penalty, u, next_token, probs_out, S_(stream));
}

int CC_API(cc_sample_step_lp)(const float* logits, int32_t R, int32_t V, int32_t ld, float temperature, int32_t top_k, float top_p, int32_t mode,
                      const int64_t* history, int32_t hist_len, int32_t hist_ld, float repetition_penalty, int32_t stop_token,
                      float length_penalty, const float* u, int32_t* next_token, float* probs_out, void* stream) {
    if (!logits || !u || !next_token || R < 0) return CC_ERR_ARG;
    return sample_rows(logits, R, V, ld, temperature, top_k, top_p, mode, reinterpret_cast<const long long*>(history), hist_len, hist_ld,
                       repetition_penalty, u, next_token, probs_out, S_(stream), stop_token, length_penalty);
}

int64_t CC_API(cc_wgrad_scratch_bytes)(void) { return (int64_t)WGRAD_SCRATCH_BYTES; }

int CC_API(cc_gemm_wgrad)(const uint16_t* X, int32_t ldx, const uint16_t* Y, int32_t ldy, int32_t Mw, int32_t Nw, int32_t K, float* dW, int32_t ldw,
                  float* scratch, void* stream) {
    if (!X || !Y || !dW || kX3) return CC_ERR_ARG;
    return gemm_wgrad(reinterpret_cast<const act_t*>(X), ldx, reinterpret_cast<const act_t*>(Y), ldy, Mw, Nw, K, dW, ldw, scratch, S_(stream));
}

int CC_API(cc_layernorm_fwd)(const float* x, const float* gamma, const float* beta, uint16_t* y, float* mean, float* rstd, int32_t rows, int32_t D,
                     void* stream) {
    if (!x || !gamma || !beta || !y) return CC_ERR_ARG;
    return ln_fwd(x, D, nullptr, gamma, beta, reinterpret_cast<act_t*>(y), nullptr, mean, rstd, rows, D, S_(stream));
}

int CC_API(cc_attention_fwd)(const uint16_t* qkv, int32_t B, int32_t S, int32_t H, int32_t hd, int32_t causal, uint16_t* out, float* lse, void* stream) {
    if (!qkv || !out) return CC_ERR_ARG;
    return attn_fwd(reinterpret_cast<const act_t*>(qkv), B, S, H, hd, causal != 0, reinterpret_cast<act_t*>(out), lse, S_(stream));
}

int CC_API(cc_attention_bwd)(const uint16_t* qkv, const uint16_t* dout, const uint16_t* o, const float* lse, float* delta_ws, int32_t B, int32_t S,
                     int32_t H, int32_t hd, int32_t causal, uint16_t* dqkv, void* stream) {
    if (!qkv || !dout || !lse || !dqkv) return CC_ERR_ARG;
    return attn_bwd(reinterpret_cast<const act_t*>(qkv), reinterpret_cast<const act_t*>(dout), reinterpret_cast<const act_t*>(o), lse, delta_ws, B, S, H, hd,
                    causal != 0, reinterpret_cast<act_t*>(dqkv), S_(stream));
}

}  // extern "C"
