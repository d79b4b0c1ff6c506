// cc_lab_env: the one place the library may look at the environment (lab build only).
#pragma once
#include <cstdlib>

// Lab switches.  Environment variables select A/B code paths ONLY in the lab build (`make lab` -> libclipcap_hip_lab.so, -DCC_EXPERIMENTS):
// the product library (libclipcap_hip.so) never reads the environment — every such switch is frozen at its default, so the only
// process-wide state it has is the cc_*_mode / cc_prof_* hooks declared at the end of include/clipcap_hip.h.
#ifdef CC_EXPERIMENTS
inline const char* cc_lab_env(const char* name) { return getenv(name); }
#else
inline const char* cc_lab_env(const char*) { return nullptr; }
#endif

