// Error reporting + version for libmi355ppo (thread-local last-error string; no global mutable state).
#include "common.h"

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
extern "C" MI355PPO_API const char* mi355ppo_last_error(void) { return mi355ppo::g_last_error; }
