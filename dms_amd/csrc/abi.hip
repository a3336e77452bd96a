// dms_amd/csrc/abi.hip -- version + thread-local error string of libdmn_hip.so.
#include "common.h"

namespace dmn {

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

}  // namespace dmn

extern "C" int dmn_abi_version(void) { return DMN_ABI_VERSION; }
extern "C" const char *dmn_last_error(void) { return dmn::err_buf(); }
