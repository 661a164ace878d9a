// host_util.cpp -- thread-local error text behind ddrl_last_error().
#include "host_util.h"

static thread_local char g_err[1024] = "";

int ddrl_fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char *ddrl_last_error(void) { return g_err; }
