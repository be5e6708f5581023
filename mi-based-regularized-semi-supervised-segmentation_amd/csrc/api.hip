// Library-level entry points: version, thread-local error message.
#include <stdarg.h>

#include "common.h"

namespace miseg_core {
static thread_local char g_err[512] = "";
char* last_error_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace miseg_core

extern "C" int miseg_version(void) { return 301; }   // round 3 (profiles/r03_*.json are keyed by this)
extern "C" const char* miseg_last_error(void) { return miseg::last_error_buf(); }
