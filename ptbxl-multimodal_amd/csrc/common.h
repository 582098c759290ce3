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
// Lane reductions: the first four butterfly levels are DPP row operations (fused into
// v_add_f32_dpp, ~1 issue slot each); only the 16- and 32-lane levels need the LDS crossbar
// (ds_bpermute, ~100 cycles of latency each).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// every lane ends with the sum over its 16-lane row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);    // row_half_mirror
    v += dpp_move<0x140>(v);    // row_mirror
    return v;
}
// sum over each 32-lane half of the wave (the column axis of a 32x32 MFMA accumulator)
__device__ __forceinline__ float half32_sum(float v) {
    v = row16_sum(v);
    return v + __shfl_xor(v, 16, 64);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = half32_sum(v);
    return v + __shfl_xor(v, 32, 64);
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
