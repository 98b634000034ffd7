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

extern "C" {
const char* mggan_last_error(void) { return g_err; }
int mggan_version(void) { return 100; }
}
