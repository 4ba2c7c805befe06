// C-ABI housekeeping: version, last-error string.
#include "gsplat_common.h"
#include "../../include/gsplat.h"

#include <stdarg.h>
#include <stdio.h>

namespace {
thread_local char g_err[512] = "";
}

void gs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gs_last_error(void) { return g_err; }

extern "C" int gs_version(void) { return GS_ABI_VERSION; }

#ifndef GS_SOURCE_SHA
#define GS_SOURCE_SHA "unknown"
#endif
extern "C" const char* gs_source_sha(void) { return GS_SOURCE_SHA; }

extern "C" int gs_tile_size(void) { return GS_TILE; }

extern "C" int gs_grad_row_floats(void) { return GS_GRAD_ROW; }
extern "C" int gs_sh_jac_row_floats(void) { return GS_SH_JAC_ROW; }
