// Error plumbing and version of the amtx C ABI (include/amtx.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/amtx.h"

static thread_local char g_err[1024] = "";

void amtx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* amtx_last_error(void) { return g_err; }
extern "C" int amtx_version(void) { return 100; }
