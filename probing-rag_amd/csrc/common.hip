// Version + thread-local error string of libprag.so.
#include "prag_common.h"

namespace prag {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace prag

extern "C" int prag_version(void) { return PRAG_VERSION; }
extern "C" const char* prag_last_error(void) { return prag::g_err; }
