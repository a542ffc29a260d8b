// Error state + version of libdeepsee_hip.so.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/deepsee_hip.h"

static thread_local char g_err[512] = "";

void dsee_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Device-side epoch of every Philox stream (dsee_rng.h): the kernels that draw noise add *epoch to the (seed, offset) they
// are given by value, so that a captured hipGraph -- whose kernel arguments are frozen -- draws fresh noise on every replay
// once the host (or a captured device-side add) has advanced the epoch.
static const uint64_t* g_rng_epoch = nullptr;
const uint64_t* dsee_rng_epoch() { return g_rng_epoch; }

extern "C" {
int dsee_rng_set_epoch(const uint64_t* epoch_dev) {
  g_rng_epoch = epoch_dev;
  return DSEE_OK;
}
int dsee_version(void) { return 100; }
const char* dsee_last_error(void) { return g_err; }
}
