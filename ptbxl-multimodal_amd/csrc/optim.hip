// optim.hip — flat-buffer AdamW step and the per-lead z-score of the input pipeline.
//   AdamW:   torch.optim.AdamW(model.parameters(), lr, weight_decay) with torch defaults —
//            reference scripts/03_train_ecg_baseline.py:130-133, 04:158-162, 05:130.
//   z-score: PTBXLDataset._normalize("per_lead") — reference src/datasets/ptbxl.py:122-127.
// Both are pure HBM streams (AdamW: 16 B read + 12 B written per parameter).
#include "common.h"

namespace ecg {

struct AdamArgs {
    float decay;      // 1 - lr*wd
    float b1, b2;
    float step_size;  // lr / (1 - b1^t)
    float bc2_sqrt;   // sqrt(1 - b2^t)
    float eps;
    float gscale;
};

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, const AdamArgs &a) {
    g *= a.gscale;
    p *= a.decay;
    m = m + (g - m) * (1.0f - a.b1);                 // exp_avg.lerp_(grad, 1-beta1)
    v = __fmaf_rn(v, a.b2, (1.0f - a.b2) * g * g);   // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(float *__restrict__ p,
                                                    const float *__restrict__ g,
                                                    float *__restrict__ m, float *__restrict__ v,
                                                    size_t n, AdamArgs a) {
    size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        float4 pp = *reinterpret_cast<float4 *>(p + i4);
        float4 gg = *reinterpret_cast<const float4 *>(g + i4);
        float4 mm = *reinterpret_cast<float4 *>(m + i4);
        float4 vv = *reinterpret_cast<float4 *>(v + i4);
        adam1(pp.x, gg.x, mm.x, vv.x, a);
        adam1(pp.y, gg.y, mm.y, vv.y, a);
        adam1(pp.z, gg.z, mm.z, vv.z, a);
        adam1(pp.w, gg.w, mm.w, vv.w, a);
        *reinterpret_cast<float4 *>(p + i4) = pp;
        *reinterpret_cast<float4 *>(m + i4) = mm;
        *reinterpret_cast<float4 *>(v + i4) = vv;
    } else {
        for (size_t i = i4; i < n; ++i) adam1(p[i], g[i], m[i], v[i], a);
    }
}

// Graph-capturable variant: the step counter lives on the device (incremented by the update
// itself) and the bias corrections are computed in-kernel, so a captured hipGraph replays the
// correct AdamW step every time without any host-side argument changing.
__global__ __launch_bounds__(256) void adamw_graph_kernel(float *__restrict__ p,
                                                          const float *__restrict__ g,
                                                          float *__restrict__ m, float *__restrict__ v,
                                                          size_t n, const int *__restrict__ step_dev,
                                                          float lr, float b1, float b2, float eps,
                                                          float wd, float gscale) {
    const int step = step_dev[0] + 1;                 // the counter is bumped by adamw_step_bump_kernel afterwards
    AdamArgs a;
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    a.decay = (float)(1.0 - (double)lr * (double)wd);
    a.b1 = b1; a.b2 = b2;
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = eps; a.gscale = gscale;
    size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        float4 pp = *reinterpret_cast<float4 *>(p + i4);
        float4 gg = *reinterpret_cast<const float4 *>(g + i4);
        float4 mm = *reinterpret_cast<float4 *>(m + i4);
        float4 vv = *reinterpret_cast<float4 *>(v + i4);
        adam1(pp.x, gg.x, mm.x, vv.x, a);
        adam1(pp.y, gg.y, mm.y, vv.y, a);
        adam1(pp.z, gg.z, mm.z, vv.z, a);
        adam1(pp.w, gg.w, mm.w, vv.w, a);
        *reinterpret_cast<float4 *>(p + i4) = pp;
        *reinterpret_cast<float4 *>(m + i4) = mm;
        *reinterpret_cast<float4 *>(v + i4) = vv;
    } else {
        for (size_t i = i4; i < n; ++i) adam1(p[i], g[i], m[i], v[i], a);
    }
}
__global__ void adamw_step_bump_kernel(int *step_dev) { step_dev[0] += 1; }

}  // namespace ecg

using namespace ecg;

ECG_API int ecg_adamw_step(float *p, const float *g, float *m, float *v, size_t n, int step,
                           float lr, float beta1, float beta2, float eps, float weight_decay,
                           float grad_scale, ecg_stream_t stream) {
    ECG_REQUIRE(p && g && m && v, "adamw_step: null pointer");
    ECG_REQUIRE(step >= 1, "adamw_step: step=%d must be >= 1", step);
    ECG_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0,
                "adamw_step: buffers must be 16-byte aligned");
    if (n == 0) return ECG_OK;
    AdamArgs a;
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.decay = (float)(1.0 - (double)lr * (double)weight_decay);
    a.b1 = beta1; a.b2 = beta2;
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = eps; a.gscale = grad_scale;
    hipLaunchKernelGGL(adamw_kernel, dim3(cdiv((long long)((n + 3) / 4), 256)), dim3(256), 0,
                       as_stream(stream), p, g, m, v, n, a);
    return check_launch("adamw_kernel");
}

ECG_API int ecg_adamw_step_graph(float *p, const float *g, float *m, float *v, size_t n,
                                 int *step_dev, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, float grad_scale, ecg_stream_t stream) {
    ECG_REQUIRE(p && g && m && v && step_dev, "adamw_step_graph: null pointer");
    ECG_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0,
                "adamw_step_graph: buffers must be 16-byte aligned");
    if (n == 0) return ECG_OK;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(adamw_graph_kernel, dim3(cdiv((long long)((n + 3) / 4), 256)), dim3(256), 0, st,
                       p, g, m, v, n, step_dev, lr, beta1, beta2, eps, weight_decay, grad_scale);
    int rc = check_launch("adamw_graph_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(adamw_step_bump_kernel, dim3(1), dim3(1), 0, st, step_dev);
    return check_launch("adamw_step_bump_kernel");
}
