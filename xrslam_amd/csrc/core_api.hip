// core_api.hip -- device discovery / error reporting entry points of the C ABI.
#include "common.hip.h"
#include "kernel_rev.gen.h"   // written by build.sh: hash of the device sources

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

const char *xrhip_kernel_revision(void) { return XRHIP_KERNEL_REV; }

int xrhip_get_device(int *device) {
    if (!device) return xr_fail(XRHIP_EINVAL, "xrhip_get_device: null argument");
    XR_HIP(hipGetDevice(device));
    return XRHIP_OK;
}

int xrhip_bind_device(int device) {
    XR_HIP(hipSetDevice(device));
    return XRHIP_OK;
}

}   // extern "C"
