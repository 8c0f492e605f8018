// Library-wide pieces of the C ABI: version and error reporting.
#include "common.h"

namespace ppo {

char *error_buffer()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ppo

extern "C" int ppo_version(void) { return 1; }
extern "C" const char *ppo_last_error(void) { return ppo::error_buffer(); }
