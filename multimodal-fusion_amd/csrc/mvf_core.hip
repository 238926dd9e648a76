// Library-wide state of the mvfusion C ABI: version and the thread-local error
// string.  There is no other global state: every buffer belongs to the caller.
#include "mvf_common.h"

namespace mvf {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

}  // namespace mvf

extern "C" int mvf_version(void) { return MVF_VERSION; }
extern "C" const char *mvf_last_error(void) { return mvf::g_error; }
