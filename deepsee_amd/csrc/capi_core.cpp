// Error state + version of libdeepsee_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/deepsee_hip.h"

static thread_local char g_err[512] = "";

void dsee_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
int dsee_version(void) { return 100; }
const char* dsee_last_error(void) { return g_err; }
}
