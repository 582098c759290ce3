// tail.hip — the small end of the network: global average pool, Linear layers, FiLM fusion,
// BCE-with-logits, sigmoid.  Every tensor here is at most [B, 512]; these kernels are
// latency-bound, so they are written for few launches and coalesced rows, not for MFMA.
//
// Replaces ATen mean / addmm / tanh / mul / add / binary_cross_entropy_with_logits behind
//   ECGCNN.gap/proj/head           reference src/models/ecg_cnn.py:46-50,62-64
//   DemoEncoder / film_gen / head  reference src/models/ecg_multimodal.py:51-59,85-98
//   the loss                        reference src/training/loop.py:32, loop_demo.py:10,33
#include "common.h"

namespace ecg {

// ---- AdaptiveAvgPool1d(1): one wave per row -----------------------------------------------
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float *__restrict__ p,
                                                      float *__restrict__ g, int rows, int L) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int lane = threadIdx.x & 63;
    const float *r = p + (size_t)row * L;
    float a = 0.f;
    for (int t = lane; t < L; t += 64) a += r[t];
    a = wave_sum(a);
    if (lane == 0) g[row] = a / (float)L;
}

__global__ void gap_bwd_kernel(const float *__restrict__ dg, float *__restrict__ dp, int L,
                               float invL, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dp[i] = dg[i / L] * invL;
}

// ---- small strided GEMM: C[m][n] = act(bias[n] + sum_k A(m,k) * B(k,n)) -----------------------
// A(m,k) = A[m*sam + k*sak] (optionally masked: value kept only where amask[same index] > 0),
// B(k,n) = B[k*sbk + n*sbn].  16x16 output tile per 256-thread workgroup, K walked in 16-chunks
// through LDS.  Used for Linear forward, input-grad and weight-grad (sizes <= 256x512x256).
constexpr int kT = 16;
__global__ __launch_bounds__(256) void gemm_small_kernel(
    const float *__restrict__ A, const float *__restrict__ amask, const float *__restrict__ B,
    const float *__restrict__ bias, float *__restrict__ Cm, int M, int Nn, int K, long sam,
    long sak, long sbk, long sbn, int relu) {
    __shared__ float As[kT][kT + 1], Bs[kT][kT + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * kT, n0 = blockIdx.x * kT;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += kT) {
        {
            int m = m0 + ty, k = k0 + tx;
            float v = 0.f;
            if (m < M && k < K) {
                size_t ia = (size_t)m * sam + (size_t)k * sak;
                v = A[ia];
                if (amask && !(amask[ia] > 0.f)) v = 0.f;
            }
            As[ty][tx] = v;
            int kb = k0 + ty, n = n0 + tx;
            Bs[ty][tx] = (kb < K && n < Nn) ? B[(size_t)kb * sbk + (size_t)n * sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kT; ++kk) acc = __fmaf_rn(As[ty][kk], Bs[kk][tx], acc);
        __syncthreads();
    }
    int m = m0 + ty, n = n0 + tx;
    if (m < M && n < Nn) {
        if (bias) acc += bias[n];
        if (relu && acc < 0.f) acc = 0.f;
        Cm[(size_t)m * Nn + n] = acc;
    }
}

// db[o] = sum_m g[m][o] (masked like above); one wave per column.
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ g,
                                                     const float *__restrict__ mask,
                                                     float *__restrict__ out, int M, int Out) {
    int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= Out) return;
    int lane = threadIdx.x & 63;
    float a = 0.f;
    for (int m = lane; m < M; m += 64) {
        size_t i = (size_t)m * Out + o;
        float v = g[i];
        if (mask && !(mask[i] > 0.f)) v = 0.f;
        a += v;
    }
    a = wave_sum(a);
    if (lane == 0) out[o] = a;
}

// ---- FiLM -----------------------------------------------------------------------------------
__global__ void film_fwd_kernel(const float *__restrict__ z, const float *__restrict__ film,
                                float *__restrict__ zc, int F, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    size_t m = i / F;
    int f = (int)(i - m * F);
    float g = 1.0f + tanhf(film[m * 2 * F + f]);
    zc[i] = __fmaf_rn(g, z[i], film[m * 2 * F + F + f]);
}
__global__ void film_bwd_kernel(const float *__restrict__ z, const float *__restrict__ film,
                                const float *__restrict__ dzc, float *__restrict__ dz,
                                float *__restrict__ dfilm, int F, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    size_t m = i / F;
    int f = (int)(i - m * F);
    float th = tanhf(film[m * 2 * F + f]);
    float d = dzc[i];
    dz[i] = d * (1.0f + th);
    dfilm[m * 2 * F + f] = d * z[i] * (1.0f - th * th);
    dfilm[m * 2 * F + F + f] = d;
}

// ---- BCE with logits (mean) -------------------------------------------------------------------
// One workgroup of 1024 threads: at the model's sizes (B x C = 1280 logits) every element is loaded in the first trip — the
// kernel is one dependent chain load -> exp / log1p -> reduce -> store, so its time is that chain's latency (5.8 us with 256
// threads walking five trips; the sums are fixed-order per thread, then lanes by butterfly, then waves in order).
constexpr int kBceThreads = 1024;
__global__ __launch_bounds__(kBceThreads) void bce_kernel(const float *__restrict__ x,
                                                          const float *__restrict__ t,
                                                          float *__restrict__ loss, float *__restrict__ dx,
                                                          int numel, double *__restrict__ running,
                                                          double weight) {
    __shared__ double red[kBceThreads / 64];
    double a = 0.0;
    const float inv = 1.0f / (float)numel;
    for (int i0 = threadIdx.x; i0 < numel; i0 += 2 * kBceThreads) {
        const int i1 = i0 + kBceThreads;
        const bool two = i1 < numel;
        const float x0 = x[i0], t0 = t[i0], x1 = two ? x[i1] : 0.f, t1 = two ? t[i1] : 0.f;      // both trips' loads in flight
        a += (double)(fmaxf(x0, 0.f) - x0 * t0 + log1pf(expf(-fabsf(x0))));
        if (dx) dx[i0] = (1.0f / (1.0f + expf(-x0)) - t0) * inv;
        if (two) {
            a += (double)(fmaxf(x1, 0.f) - x1 * t1 + log1pf(expf(-fabsf(x1))));
            if (dx) dx[i1] = (1.0f / (1.0f + expf(-x1)) - t1) * inv;
        }
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kBceThreads / 64; ++w) tot += red[w];
        const float l = (float)(tot / (double)numel);
        loss[0] = l;
        if (running) running[0] += (double)l * weight;    // epoch bookkeeping without extra launches
    }
}

__global__ void sigmoid_kernel(const float *__restrict__ x, float *__restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 1.0f / (1.0f + expf(-x[i]));
}

static int gemm_small(const float *A, const float *amask, const float *B, const float *bias,
                      float *C, int M, int N, int K, long sam, long sak, long sbk, long sbn,
                      int relu, hipStream_t st) {
    dim3 grid(cdiv(N, kT), cdiv(M, kT));
    hipLaunchKernelGGL(gemm_small_kernel, grid, dim3(256), 0, st, A, amask, B, bias, C, M, N, K,
                       sam, sak, sbk, sbn, relu);
    return check_launch("gemm_small_kernel");
}

}  // namespace ecg

using namespace ecg;

ECG_API int ecg_gap_fwd(const float *p, float *g, int rows, int L, ecg_stream_t stream) {
    ECG_REQUIRE(p && g && rows > 0 && L > 0, "gap_fwd: bad argument");
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), p, g,
                       rows, L);
    return check_launch("gap_fwd_kernel");
}

ECG_API int ecg_gap_bwd(const float *dg, float *dp, int rows, int L, ecg_stream_t stream) {
    ECG_REQUIRE(dg && dp && rows > 0 && L > 0, "gap_bwd: bad argument");
    size_t total = (size_t)rows * L;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), dg,
                       dp, L, 1.0f / (float)L, total);
    return check_launch("gap_bwd_kernel");
}

ECG_API int ecg_linear_fwd(const float *x, const float *w, const float *b, float *y, int M, int In,
                           int Out, int relu, ecg_stream_t stream) {
    ECG_REQUIRE(x && w && y && M > 0 && In > 0 && Out > 0, "linear_fwd: bad argument");
    // y[m][o] = sum_i x[m][i] * w[o][i]
    return gemm_small(x, nullptr, w, b, y, M, Out, In, In, 1, 1, In, relu, as_stream(stream));
}

ECG_API size_t ecg_linear_bwd_ws_floats(int M, int In, int Out) { (void)M; (void)In; (void)Out; return 0; }

ECG_API int ecg_linear_bwd(const float *x, const float *w, const float *y, const float *dy,
                           float *dx, float *dw, float *db, float *ws, int M, int In, int Out,
                           int relu, ecg_stream_t stream) {
    (void)ws;
    ECG_REQUIRE(x && w && dy && M > 0 && In > 0 && Out > 0, "linear_bwd: bad argument");
    ECG_REQUIRE(!relu || y, "linear_bwd: y is required when relu != 0");
    const float *mask = relu ? y : nullptr;
    hipStream_t st = as_stream(stream);
    int rc = ECG_OK;
    if (dx)   // dx[m][i] = sum_o g[m][o] * w[o][i]
        rc = gemm_small(dy, mask, w, nullptr, dx, M, In, Out, Out, 1, In, 1, 0, st);
    if (!rc && dw)   // dw[o][i] = sum_m g[m][o] * x[m][i]
        rc = gemm_small(dy, mask, x, nullptr, dw, Out, In, M, 1, Out, In, 1, 0, st);
    if (!rc && db) {
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(Out, 4)), dim3(256), 0, st, dy, mask, db, M, Out);
        rc = check_launch("colsum_kernel");
    }
    return rc;
}

ECG_API int ecg_film_fwd(const float *z, const float *film, float *zc, int M, int F,
                         ecg_stream_t stream) {
    ECG_REQUIRE(z && film && zc && M > 0 && F > 0, "film_fwd: bad argument");
    size_t total = (size_t)M * F;
    hipLaunchKernelGGL(film_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), z,
                       film, zc, F, total);
    return check_launch("film_fwd_kernel");
}

ECG_API int ecg_film_bwd(const float *z, const float *film, const float *dzc, float *dz,
                         float *dfilm, int M, int F, ecg_stream_t stream) {
    ECG_REQUIRE(z && film && dzc && dz && dfilm && M > 0 && F > 0, "film_bwd: bad argument");
    size_t total = (size_t)M * F;
    hipLaunchKernelGGL(film_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), z,
                       film, dzc, dz, dfilm, F, total);
    return check_launch("film_bwd_kernel");
}

ECG_API int ecg_bce_logits_fwd(const float *x, const float *target, float *loss, float *dx,
                               int numel, double *running_sum, double weight,
                               ecg_stream_t stream) {
    ECG_REQUIRE(x && target && loss && numel > 0, "bce_logits_fwd: bad argument");
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(kBceThreads), 0, as_stream(stream), x, target, loss, dx,
                       numel, running_sum, weight);
    return check_launch("bce_kernel");
}

ECG_API int ecg_sigmoid_fwd(const float *x, float *prob, size_t n, ecg_stream_t stream) {
    ECG_REQUIRE(x && prob, "sigmoid_fwd: null pointer");
    if (n == 0) return ECG_OK;
    hipLaunchKernelGGL(sigmoid_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), x, prob, n);
    return check_launch("sigmoid_kernel");
}
