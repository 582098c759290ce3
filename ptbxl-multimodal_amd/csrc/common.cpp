// common.cpp — error plumbing, version and device check of libecg_hip.so.
#include "common.h"
#include <cstring>

namespace ecg {

char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ecg

ECG_API int ecg_version(void) { return 100; }

ECG_API const char *ecg_last_error(void) { return ecg::err_buf(); }

ECG_API int ecg_check_device(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return ecg::fail(ECG_ENODEV, "no HIP device visible");
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return ecg::fail(ECG_ENODEV, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return ecg::fail(ECG_ENODEV, "device %d is %s; libecg_hip.so carries gfx950 code only", dev,
                         prop.gcnArchName);
    return ECG_OK;
}
