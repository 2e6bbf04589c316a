// capi.hip -- library-level C-ABI entry points (error string, version).
#include "../../include/rllab_amd.h"
#include "capi_util.h"

namespace rl {
char* error_buffer() {
    static thread_local char buf[ERROR_BUFFER_LEN] = {0};
    return buf;
}
}  // namespace rl

extern "C" const char* rl_last_error(void) { return rl::error_buffer(); }
extern "C" int rl_abi_version(void) { return 14; }
