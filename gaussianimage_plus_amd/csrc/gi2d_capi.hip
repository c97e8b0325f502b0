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

}  // namespace gi2d

extern "C" {
const char *gi2d_version(void) { return "gi2d 0.1.0 (gfx950)"; }
const char *gi2d_last_error_string(void) { return gi2d::g_last_error.c_str(); }
}
