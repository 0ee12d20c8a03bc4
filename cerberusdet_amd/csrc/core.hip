// Library plumbing: version, thread-local error string, device info.
#include <stdarg.h>

#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "common.h"
#include "switches.h"

namespace cdet {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- switches (switches.h): one table, filled from the environment when the library is loaded ------------------------------------
struct SwitchDef {
    const char* name;
    const char* env;
    int dflt;
};
static const SwitchDef kSwitches[SW_COUNT] = {
    {"conv_pp", "CDET_CONV_PP", 1},          {"conv_pair", "CDET_CONV_PAIR", 1},       {"halo_ng", "CDET_HALO_NG", SW_AUTO},
    {"halo_wg3", "CDET_HALO_WG3", SW_AUTO},  {"halo_ks", "CDET_HALO_KS", SW_AUTO},     {"wgrad_halo", "CDET_WGRAD_HALO", SW_AUTO},
    {"wgrad_patch", "CDET_WGRAD_PATCH", SW_AUTO}, {"wgrad_s2", "CDET_WGRAD_S2", 1},    {"peer_spin_ms", "CDET_PEER_SPIN_MS", 600000},
};
static std::atomic<int> g_sw[SW_COUNT];
namespace {
struct SwitchInit {
    SwitchInit() {
        for (int i = 0; i < SW_COUNT; ++i) {
            const char* e = getenv(kSwitches[i].env);
            g_sw[i].store((e && *e) ? atoi(e) : kSwitches[i].dflt, std::memory_order_relaxed);
        }
    }
} g_switch_init;
}  // namespace
int sw(int id) { return g_sw[id].load(std::memory_order_relaxed); }
static int switch_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < SW_COUNT; ++i)
        if (!strcmp(name, kSwitches[i].name) || !strcmp(name, kSwitches[i].env)) return i;
    return -1;
}
}  // namespace cdet

extern "C" int cdet_set_switch(const char* name, int32_t value) {
    const int i = cdet::switch_index(name);
    CDET_CHECK_ARG(i >= 0, "cdet_set_switch: unknown switch '%s'", name ? name : "(null)");
    cdet::g_sw[i].store(value == CDET_SWITCH_DEFAULT ? cdet::kSwitches[i].dflt : (int)value, std::memory_order_relaxed);
    return 0;
}

extern "C" int cdet_get_switch(const char* name, int32_t* value) {
    const int i = cdet::switch_index(name);
    CDET_CHECK_ARG(i >= 0 && value, "cdet_get_switch: unknown switch '%s'", name ? name : "(null)");
    *value = cdet::sw(i);
    return 0;
}

// "name=value ..." for every switch that is not at its default, then the build flavour. Returns the length needed (excluding the terminator).
extern "C" int cdet_active_switches(char* buf, int32_t n) {
    char tmp[1024];
    size_t at = 0;
    for (int i = 0; i < cdet::SW_COUNT; ++i) {
        const int v = cdet::sw(i);
        if (v == cdet::kSwitches[i].dflt) continue;
        at += (size_t)snprintf(tmp + at, sizeof(tmp) - at, "%s=%d ", cdet::kSwitches[i].name, v);
        if (at >= sizeof(tmp) - 64) break;
    }
#ifdef CDET_EXPERIMENTS
    at += (size_t)snprintf(tmp + at, sizeof(tmp) - at, "build=experiments ");
#endif
#ifdef CDET_PROFILING
    at += (size_t)snprintf(tmp + at, sizeof(tmp) - at, "build=profiling ");
#endif
    if (at > 0 && tmp[at - 1] == ' ') --at;
    tmp[at] = 0;
    if (buf && n > 0) {
        strncpy(buf, tmp, (size_t)n - 1);
        buf[n - 1] = 0;
    }
    return (int)at;
}

// 1 when the library carries the opt-in forms kept for the record (make EXTRA=-DCDET_EXPERIMENTS: 384- / 512-pixel tiles, in-launch BatchNorm fold)
extern "C" int cdet_has_experiments(void) {
#ifdef CDET_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

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
