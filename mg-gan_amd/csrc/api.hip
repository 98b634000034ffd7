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

// ---- launch log (measurement aid) ------------------------------------------------------------------------------
// MG_LAUNCH notes (host function pointer, threads) of every launch while the log is on; mggan_launch_log_read hands back
// the symbols (hipKernelNameRefByPtr: the mangled device-function names, i.e. what rocprofv3 prints without ".kd").
int g_mggan_launch_log = 0;
struct LaunchNote {
  const void* fn;
  long threads;
};
static std::mutex g_log_mu;
static LaunchNote g_log[64];
static int g_log_n = 0;

void mggan_note_launch(const void* host_fn, dim3 grid, dim3 block) {
  std::lock_guard<std::mutex> lk(g_log_mu);
  if (g_log_n < 64) {
    g_log[g_log_n].fn = host_fn;
    g_log[g_log_n].threads = (long)grid.x * grid.y * grid.z * block.x * block.y * block.z;
    ++g_log_n;
  }
}

// One lane writes the constant-rate (100 MHz) device clock: a time mark inside a stream or a captured graph,
// read back after the fact (profilers perturb a latency-bound multi-stream graph; marks do not).
__global__ void timestamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

extern "C" {
const char* mggan_last_error(void) { return g_err; }
int mggan_version(void) { return 100; }

int mggan_launch_log(int on) {
  std::lock_guard<std::mutex> lk(g_log_mu);
  g_log_n = 0;
  g_mggan_launch_log = on ? 1 : 0;
  return MGGAN_OK;
}

int mggan_launch_log_read(char* out, int cap) {
  MG_CHECK_ARG(out && cap > 0, "launch_log_read: no buffer");
  std::lock_guard<std::mutex> lk(g_log_mu);
  int pos = 0;
  out[0] = 0;
  for (int i = 0; i < g_log_n; ++i) {
    const char* nm = hipKernelNameRefByPtr(g_log[i].fn, nullptr);
    (void)hipGetLastError();
    const int w = snprintf(out + pos, (size_t)(cap - pos), "%s%s:%ld", i ? ";" : "", nm ? nm : "?", g_log[i].threads);
    if (w < 0 || w >= cap - pos) {
      out[pos] = 0;
      break;
    }
    pos += w;
  }
  g_log_n = 0;
  return MGGAN_OK;
}

int mggan_timestamp(unsigned long long* slot, hipStream_t stream) {
  MG_CHECK_ARG(slot, "timestamp: null pointer");
  MG_LAUNCH(timestamp_kernel, dim3(1), dim3(1), 0, stream, slot);
  MG_LAUNCH_CHECK("timestamp");
  return MGGAN_OK;
}
}
