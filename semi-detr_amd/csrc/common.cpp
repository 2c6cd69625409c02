#include "common.h"

namespace semidetr {

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

}  // namespace semidetr

extern "C" int semidetr_abi_version(void) { return SEMIDETR_ABI_VERSION; }
extern "C" const char *semidetr_last_error(void) { return semidetr::error_buffer(); }
