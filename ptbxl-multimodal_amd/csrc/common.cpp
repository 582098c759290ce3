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

ECG_API int ecg_host_gather_rows(const void *src, size_t row_bytes, const long long *rows, int n, void *dst) {
    ECG_REQUIRE(n >= 0 && (n == 0 || (src && rows && dst && row_bytes > 0)), "host_gather_rows: n=%d row_bytes=%zu", n, row_bytes);
    const char *s = static_cast<const char *>(src);
    char *d = static_cast<char *>(dst);
    int i = 0;
    while (i < n) {                                    // consecutive records (an unshuffled epoch): one copy per run
        ECG_REQUIRE(rows[i] >= 0, "host_gather_rows: negative row index %lld", rows[i]);
        int j = i + 1;
        while (j < n && rows[j] == rows[j - 1] + 1) ++j;
        memcpy(d + (size_t)i * row_bytes, s + (size_t)rows[i] * row_bytes, (size_t)(j - i) * row_bytes);
        i = j;
    }
    return ECG_OK;
}
