// Error reporting shared by all entry points.
#include "bcbf_common.h"
#include <stdio.h>
#include <string.h>

namespace bcbf {
static thread_local char g_err[256] = "";

void set_error(const char* what, hipError_t err) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
}
int check_launch(const char* what) {
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) { set_error(what, err); return BCBF_ELAUNCH; }
    return BCBF_OK;
}
}  // namespace bcbf

extern "C" int bcbf_version(void) { return BCBF_VERSION_MAJOR * 100 + BCBF_VERSION_MINOR; }
extern "C" const char* bcbf_last_error(void) { return bcbf::g_err; }
