// Library bookkeeping: ABI version + last-error text.
#include <string>

#include "common.h"

namespace dfine {
static thread_local std::string g_last_error;
void set_last_error(hipError_t e) { g_last_error = hipGetErrorString(e); }
}  // namespace dfine

extern "C" {
int dfine_abi_version(void) { return 1; }
const char *dfine_last_error(void) { return dfine::g_last_error.c_str(); }
}
