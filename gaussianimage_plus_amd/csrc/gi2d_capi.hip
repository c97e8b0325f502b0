// Error plumbing and version string of the C ABI (include/gi2d.h).
#include <string>

#include "gi2d_common.h"

namespace gi2d {

static thread_local std::string g_last_error;

void set_error(const char *msg) { g_last_error = msg ? msg : ""; }

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return GI2D_OK;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return (int)e;
}

#define GI2D_STR2(x) #x
#define GI2D_STR(x) GI2D_STR2(x)
// Every development switch of the kernels (cut-off, knock-out, trace and tuning macros: csrc/Makefile passes them to all
// translation units alike).  A library built with any of them says so in its version string -- " dev[...]" -- and
// gaussianimage_plus_amd/_lib.py refuses to load it unless GI2D_ALLOW_DEV_BUILD=1: several of them give wrong results
// on purpose (timing aids).
static std::string dev_switches() {
    std::string s;
    const auto add = [&](const char *name, const char *value) {
        if (!s.empty()) s += ' ';
        s += name;
        if (value && *value) s += std::string("=") + value;
    };
    (void)add;
#ifdef GI2D_STOP_AFTER
    add("GI2D_STOP_AFTER", GI2D_STR(GI2D_STOP_AFTER));
#endif
#ifdef GI2D_FUSED_TRACE
    add("GI2D_FUSED_TRACE", "");
#endif
#ifdef GI2D_UPDATE_STOP
    add("GI2D_UPDATE_STOP", GI2D_STR(GI2D_UPDATE_STOP));
#endif
#ifdef GI2D_INBOX_STATS
    add("GI2D_INBOX_STATS", "");
#endif
#ifdef GI2D_DEV_VARIANT /* any other experiment: -DGI2D_DEV_VARIANT=name next to its own macros */
    add("GI2D_DEV_VARIANT", GI2D_STR(GI2D_DEV_VARIANT));
#endif
    return s;
}

}  // namespace gi2d

extern "C" {
const char *gi2d_version(void) {
    static const std::string v = [] {
        std::string s = "gi2d 0.2.0 (gfx950)";
        const std::string d = gi2d::dev_switches();
        if (!d.empty()) s += " dev[" + d + "]";
        return s;
    }();
    return v.c_str();
}
const char *gi2d_last_error_string(void) { return gi2d::g_last_error.c_str(); }
}
