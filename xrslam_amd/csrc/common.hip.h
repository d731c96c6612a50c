// common.hip.h -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "hostprof.hpp"

#include <cstdio>
#include <cstring>

#include "../../include/xrslam_hip.h"

namespace xrhip {

inline char *xr_err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int xr_fail(int code, const char *msg) {
    std::snprintf(xr_err_buf(), 512, "%s", msg);
    return code;
}

inline int xr_fail_hip(hipError_t e, const char *expr, const char *file, int line) {
    std::snprintf(xr_err_buf(), 512, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, expr);
    return XRHIP_EHIP;
}

// Fails loudly when there is no gfx950 device: the product has no CPU path.
inline int xr_require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return xr_fail(XRHIP_ENODEVICE, "no HIP device available: the XRSLAM hot path requires an MI355X (gfx950); there is no CPU fallback");
    }
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return xr_fail_hip(e, "hipGetDeviceProperties", __FILE__, __LINE__);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::snprintf(xr_err_buf(), 512, "device %d is %s; this library carries gfx950 code objects only", dev, prop.gcnArchName);
        return XRHIP_ENODEVICE;
    }
    return XRHIP_OK;
}


}   // namespace xrhip

#define XR_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t xr_e_ = (expr);                                                \
        if (xr_e_ != hipSuccess) return ::xrhip::xr_fail_hip(xr_e_, #expr, __FILE__, __LINE__); \
    } while (0)
