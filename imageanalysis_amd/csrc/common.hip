// libiamx: error plumbing + version.
#include <stdarg.h>

#include "iamx_common.h"

namespace iamx {

char *err_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(IAMX_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return IAMX_OK;
}

}  // namespace iamx

extern "C" int iamx_version(void) { return 100; }
extern "C" const char *iamx_last_error(void) { return iamx::err_buf(); }
extern "C" const char *iamx_arch(void) { return "gfx950"; }
