// common.h — shared host/device helpers for libecg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "ecg_hip.h"

#define ECG_API extern "C" __attribute__((visibility("default")))

namespace ecg {

constexpr int kWave = 64;  // CDNA4 wavefront

// thread-local error text: entry points are re-entrant and keep no shared mutable state
char *err_buf();
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(ecg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ECG_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return ECG_OK;
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

#define ECG_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::ecg::fail(ECG_EINVAL, __VA_ARGS__); } while (0)

// ---- device helpers ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// BatchNorm affine exactly as every kernel (and the oracle) evaluates it.
__device__ __forceinline__ float bn_apply1(float y, float mean, float scale, float beta) {
    return __fmaf_rn(y - mean, scale, beta);
}

}  // namespace ecg
