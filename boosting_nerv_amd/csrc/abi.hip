// abi.hip -- error reporting and version entry points of the C-ABI.
#include "common.h"
#include <string.h>

namespace { thread_local char g_err[512] = ""; }

int bnerv_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int bnerv_abi_version(void) { return BNERV_ABI_VERSION; }
extern "C" const char* bnerv_last_error(void) { return g_err; }
extern "C" const char* bnerv_build_arch(void) { return "gfx950"; }
