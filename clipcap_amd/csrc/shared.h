// State shared by the two operand-type builds of the kernels (bf16 / fp16): process-wide test and measurement hooks only.
// The compute entry points of the C ABI keep no state of their own.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

#include "lab_env.h"

namespace cc_shared {

extern int g_gemm_tile_mode;   // cc_gemm_tile_mode: -1 chooser, 0 = 128 x 128 only, 3 / 4 = force 256 x 192 / 256 x 256
extern int g_gemm_s64;         // cc_gemm_skinny_mode
extern int g_gemm_small_x2;    // env CC_GEMM_X2
extern int g_decode_mode;      // cc_decode_mode
extern int g_decode_last_path; // cc_decode_last_path

// cc_prof_start / cc_prof_stop: HIP events around the launches of one call site (or of every GEMM, CC_SITE_ALL_GEMMS)
struct Prof {
    int site = 0, cap = 0, n = 0;
    bool busy = false;            // a bracket is open: nested host wrappers (split-K fallbacks) must not open a second one
    std::vector<hipEvent_t> ev;   // 2 per sample
    std::vector<double> flops;    // 2*M*N*K of the bracketed launch (CC_SITE_ALL_GEMMS), else 0
};
extern Prof g_prof;

struct ProfScope {
    hipStream_t st;
    bool on;
    ProfScope(int site, hipStream_t s, double flops = 0.0) : st(s), on(g_prof.site == site && g_prof.n < g_prof.cap && !g_prof.busy) {
        if (on) {
            g_prof.busy = true;
            g_prof.flops[g_prof.n] = flops;
            (void)hipEventRecord(g_prof.ev[2 * g_prof.n], st);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st);
            g_prof.n++;
            g_prof.busy = false;
        }
    }
};

constexpr int SITE_ALL_GEMMS = 100;   // == CC_SITE_ALL_GEMMS in include/clipcap_hip.h

}  // namespace cc_shared
