// Error string + version for libmggan_hip.so.
#include "common.h"
#include "../../include/mggan_hip.h"
#include <stdarg.h>
#include <mutex>

static std::mutex g_err_mu;
static char g_err[512] = "";

void mggan_set_error(const char* fmt, ...) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// One lane writes the constant-rate (100 MHz) device clock: a time mark inside a stream or a captured graph,
// read back after the fact (profilers perturb a latency-bound multi-stream graph; marks do not).
__global__ void timestamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

extern "C" {
const char* mggan_last_error(void) { return g_err; }
int mggan_version(void) { return 100; }

int mggan_timestamp(unsigned long long* slot, hipStream_t stream) {
  MG_CHECK_ARG(slot, "timestamp: null pointer");
  hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, stream, slot);
  MG_LAUNCH_CHECK("timestamp");
  return MGGAN_OK;
}
}
