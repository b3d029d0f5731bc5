#include "pd_common.h"
#include <cstring>

namespace pd {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace pd

extern "C" const char* pd_last_error(void) { return pd::err_buf(); }
extern "C" int pd_version(void) { return 1; }
