// Error channel + misc entry points of libsummerset_hip.so.
#include "smr_common.h"

namespace smr {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
}  // namespace smr

extern "C" {

const char *smr_last_error(void) { return smr::g_last_error.c_str(); }

int smr_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        smr::set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return SMR_ERR_DEVICE;
    }
    return n;
}

uint32_t smr_abi_version(void) { return 1; }

}  // extern "C"
