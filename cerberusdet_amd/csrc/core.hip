// Library plumbing: version, thread-local error string, device info.
#include <stdarg.h>

#include "common.h"

namespace cdet {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace cdet

extern "C" int cdet_version(void) { return CDET_ABI_VERSION; }
extern "C" const char* cdet_last_error(void) { return cdet::g_err; }
extern "C" int cdet_device_info(int32_t* out3) {
    if (!out3) return -1001;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        cdet::set_error("hipGetDevice: %s", hipGetErrorString(e));
        return -(int)e;
    }
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        cdet::set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return -(int)e;
    }
    out3[0] = p.multiProcessorCount;
    out3[1] = (int32_t)p.maxSharedMemoryPerMultiProcessor;
    int arch = 0;
    sscanf(p.gcnArchName, "gfx%d", &arch);
    out3[2] = arch;
    return 0;
}
