// Error reporting + version for libmi355ppo (thread-local last-error string; no global mutable state).
#include "common.h"
#include <string.h>

namespace mi355ppo {
static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}
}  // namespace mi355ppo

extern "C" MI355PPO_API int mi355ppo_version(void) { return MI355PPO_VERSION; }

// Capability check (SURVEY section 8(b) names an init entry point; the library keeps no per-device state, so this only answers
// "can the kernels of this library run on that device?"): the code objects are gfx950 only.
extern "C" MI355PPO_API int mi355ppo_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        mi355ppo::set_error("mi355ppo_init: no HIP device (%s)", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return MI355PPO_EHIP;
    }
    MI355_REQUIRE(device >= 0 && device < n, MI355PPO_EINVAL, "mi355ppo_init: device %d out of range (%d visible)", device, n);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        mi355ppo::set_error("mi355ppo_init: hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
        return MI355PPO_EHIP;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        mi355ppo::set_error("mi355ppo_init: device %d is %s; libmi355ppo contains gfx950 (MI355X) code only", device, prop.gcnArchName);
        return MI355PPO_EHIP;
    }
    return MI355PPO_OK;
}
extern "C" MI355PPO_API const char* mi355ppo_last_error(void) { return mi355ppo::g_last_error; }
