// Variant-independent part of the C ABI: version, the process-wide test knobs and the measurement hooks (include/clipcap_hip.h).
#include "../../include/clipcap_hip.h"
#include "shared.h"
#include <algorithm>
#include <cstdlib>

namespace cc_shared {
int g_gemm_tile_mode = []() { const char* e = getenv("CC_GEMM_S256"); return e ? atoi(e) : -1; }();
int g_gemm_s64 = []() { const char* e = getenv("CC_GEMM_S64"); return e ? atoi(e) : -1; }();
int g_gemm_small_x2 = []() { const char* e = getenv("CC_GEMM_X2"); return e ? atoi(e) : 1; }();
Prof g_prof;
}  // namespace cc_shared

using namespace cc_shared;

extern "C" {

int cc_abi_version(void) { return CC_ABI_VERSION; }

int cc_gemm_tile_mode(int32_t mode) {
    const int old = g_gemm_tile_mode;
    g_gemm_tile_mode = mode;
    return old;
}

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

}  // extern "C"
