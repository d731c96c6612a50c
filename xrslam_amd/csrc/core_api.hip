// core_api.hip -- device discovery / error reporting entry points of the C ABI.
#include "common.hip.h"

using namespace xrhip;

extern "C" {

const char *xrhip_last_error(void) { return xr_err_buf(); }

int xrhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int xrhip_set_device(int device) {
    XR_HIP(hipSetDevice(device));
    return xr_require_device();
}

}   // extern "C"
