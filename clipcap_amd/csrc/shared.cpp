// Variant-independent part of the C ABI: version, the process-wide test knobs and the measurement hooks (include/clipcap_hip.h).
#include "../../include/clipcap_hip.h"
#ifdef CC_EXPERIMENTS
#include "../../include/clipcap_hip_lab.h"
#endif
#include "shared.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace cc_shared {
int g_gemm_tile_mode = []() { const char* e = cc_lab_env("CC_GEMM_S256"); return e ? atoi(e) : -1; }();
int g_gemm_s64 = []() { const char* e = cc_lab_env("CC_GEMM_S64"); return e ? atoi(e) : -1; }();
int g_gemm_small_x2 = []() { const char* e = cc_lab_env("CC_GEMM_X2"); return e ? atoi(e) : 1; }();
// bit 0: beam-group attention step (k_decode_attn_group), bit 1: the layer stack of a group step as one persistent launch (decode_pk.hip;
// measured slower than the per-op launches on MI355X, HISTORY.md 4.5 — kept for A/B runs), bit 2: the XCD-team engine (decode_xt.hip) when cc_decode_fwd_x
// is given a weight image, bit 3: the K-split decode GEMMs take their weight operand global -> VGPR from the fragment-ordered image of cc_decode_image
// (gemm_nt_s64kwb_kernel; bit-identical results; with the weights cold from HBM, as in the chain, 1-3 % slower than both operands through LDS —
// HISTORY.md 4.5, tools/ab_decode_mode.py; default off).  env CC_DEC_GROUP / CC_DEC_PK / CC_DEC_XT preset bits 0-2.
int g_decode_mode = []() {
    const char* g = cc_lab_env("CC_DEC_GROUP");
    const char* p = cc_lab_env("CC_DEC_PK");
    const char* x = cc_lab_env("CC_DEC_XT");
    return ((g ? atoi(g) : 1) ? 1 : 0) | ((p ? atoi(p) : 0) ? 2 : 0) | ((x ? atoi(x) : 0) ? 4 : 0);
}();
Prof g_prof;
int g_decode_last_path = 0;
}  // namespace cc_shared

using namespace cc_shared;

extern "C" {

int cc_abi_version(void) { return CC_ABI_VERSION; }

int cc_gemm_tile_mode(int32_t mode) {
    const int old = g_gemm_tile_mode;
    g_gemm_tile_mode = mode;
    return old;
}

int cc_decode_mode(int32_t mode) {
    const int old = g_decode_mode;
    if (mode >= 0) g_decode_mode = mode & 15;
    return old;
}

#ifdef CC_EXPERIMENTS
int cc_decode_last_path(void) { return g_decode_last_path; }
#endif

int cc_gemm_skinny_mode(int32_t mode) {
    const int old = g_gemm_s64;
    g_gemm_s64 = mode;
    return old;
}

int cc_prof_start(int32_t site, int32_t max_samples) {
    if (site < 0 || max_samples < 0 || max_samples > (1 << 16)) return CC_ERR_ARG;
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.flops.assign((size_t)max_samples, 0.0);
    g_prof.n = 0;
    g_prof.cap = max_samples;
    g_prof.site = site;
    for (int i = 0; i < 2 * max_samples; i++) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return CC_ERR_LAUNCH;
        g_prof.ev.push_back(e);
    }
    return CC_OK;
}

int cc_prof_stop(float* ms_host, double* flops_host, int32_t* n_host) {
    if (!ms_host || !n_host) return CC_ERR_ARG;
    const int n = std::min(g_prof.n, (int)*n_host);
    for (int i = 0; i < n; i++) {
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return CC_ERR_LAUNCH;
        if (hipEventElapsedTime(&ms_host[i], g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return CC_ERR_LAUNCH;
        if (flops_host) flops_host[i] = g_prof.flops[i];
    }
    *n_host = n;
    g_prof.site = 0;
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.flops.clear();
    g_prof.n = g_prof.cap = 0;
    return CC_OK;
}

// ---------------------------------------------------------------- RCCL collective (cc_allreduce_bucket) ----------
// RCCL is bound at run time (dlopen of the librccl the process already carries — PyTorch-ROCm ships one — or the ROCm one), so the
// library has no link-time dependency on a particular copy and single-GPU users never load it.
namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*get_uid)(ncclUniqueId*) = nullptr;
    ncclResult_t (*init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*count)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*bcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
    Rccl() {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return;
        get_uid = reinterpret_cast<decltype(get_uid)>(dlsym(h, "ncclGetUniqueId"));
        init_rank = reinterpret_cast<decltype(init_rank)>(dlsym(h, "ncclCommInitRank"));
        all_reduce = reinterpret_cast<decltype(all_reduce)>(dlsym(h, "ncclAllReduce"));
        destroy = reinterpret_cast<decltype(destroy)>(dlsym(h, "ncclCommDestroy"));
        count = reinterpret_cast<decltype(count)>(dlsym(h, "ncclCommCount"));
        bcast = reinterpret_cast<decltype(bcast)>(dlsym(h, "ncclBroadcast"));
        reduce = reinterpret_cast<decltype(reduce)>(dlsym(h, "ncclReduce"));
        ok = get_uid && init_rank && all_reduce && destroy;
    }
};
extern "C++" Rccl& rccl() {
    static Rccl r;
    return r;
}
static_assert(sizeof(ncclUniqueId) == CC_COMM_UID_BYTES, "cc_comm unique id size");
}  // namespace

int cc_comm_unique_id(uint8_t* uid_host) {
    if (!uid_host) return CC_ERR_ARG;
    if (!rccl().ok) return CC_ERR_STATE;
    ncclUniqueId id;
    if (rccl().get_uid(&id) != ncclSuccess) return CC_ERR_LAUNCH;
    memcpy(uid_host, &id, sizeof(id));
    return CC_OK;
}

int cc_comm_create(void** comm, int32_t nranks, int32_t rank, const uint8_t* uid_host) {
    if (!comm || !uid_host || nranks < 1 || rank < 0 || rank >= nranks) return CC_ERR_ARG;
    if (!rccl().ok) return CC_ERR_STATE;
    ncclUniqueId id;
    memcpy(&id, uid_host, sizeof(id));
    ncclComm_t c = nullptr;
    if (rccl().init_rank(&c, nranks, id, rank) != ncclSuccess) return CC_ERR_LAUNCH;
    *comm = c;
    return CC_OK;
}

int cc_allreduce_bucket(void* comm, void* buf, int64_t count, int32_t dtype, void* stream) {
    if (!comm || !buf || count < 0 || dtype < CC_RED_F32 || dtype > CC_RED_F16) return CC_ERR_ARG;
    if (!rccl().ok) return CC_ERR_STATE;
    if (count == 0) return CC_OK;
    const ncclDataType_t t = dtype == CC_RED_F32 ? ncclFloat32 : dtype == CC_RED_BF16 ? ncclBfloat16 : ncclFloat16;
    return rccl().all_reduce(buf, buf, (size_t)count, t, ncclSum, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)) == ncclSuccess
               ? CC_OK : CC_ERR_LAUNCH;
}

int cc_broadcast_bucket(void* comm, void* buf, int64_t count, int32_t dtype, int32_t root, void* stream) {
    if (!comm || !buf || count < 0 || root < 0 || dtype < CC_RED_F32 || dtype > CC_RED_F16) return CC_ERR_ARG;
    if (!rccl().ok || !rccl().bcast) return CC_ERR_STATE;
    if (count == 0) return CC_OK;
    const ncclDataType_t t = dtype == CC_RED_F32 ? ncclFloat32 : dtype == CC_RED_BF16 ? ncclBfloat16 : ncclFloat16;
    return rccl().bcast(buf, buf, (size_t)count, t, root, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)) == ncclSuccess
               ? CC_OK : CC_ERR_LAUNCH;
}

int cc_reduce_bucket(void* comm, void* buf, int64_t count, int32_t dtype, int32_t root, void* stream) {
    if (!comm || !buf || count < 0 || root < 0 || dtype < CC_RED_F32 || dtype > CC_RED_F16) return CC_ERR_ARG;
    if (!rccl().ok || !rccl().reduce) return CC_ERR_STATE;
    if (count == 0) return CC_OK;
    const ncclDataType_t t = dtype == CC_RED_F32 ? ncclFloat32 : dtype == CC_RED_BF16 ? ncclBfloat16 : ncclFloat16;
    return rccl().reduce(buf, buf, (size_t)count, t, ncclSum, root, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)) == ncclSuccess
               ? CC_OK : CC_ERR_LAUNCH;
}

int cc_comm_count(void* comm, int32_t* nranks) {
    if (!comm || !nranks) return CC_ERR_ARG;
    if (!rccl().ok || !rccl().count) return CC_ERR_STATE;
    int n = 0;
    if (rccl().count(static_cast<ncclComm_t>(comm), &n) != ncclSuccess) return CC_ERR_LAUNCH;
    *nranks = n;
    return CC_OK;
}

int cc_comm_destroy(void* comm) {
    if (!comm) return CC_ERR_ARG;
    if (!rccl().ok) return CC_ERR_STATE;
    return rccl().destroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? CC_OK : CC_ERR_LAUNCH;
}

}  // extern "C"
