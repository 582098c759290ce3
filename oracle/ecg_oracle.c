/*
 * ecg_oracle.c — CPU restatement of the arithmetic on the ptbxl-multimodal 1D-CNN
 * train/eval path.  TEST INFRASTRUCTURE ONLY: imported by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker.
 * Nothing under ptbxl-multimodal_amd/ may link, import or call this file.
 *
 * The reference (cyu0330/ptbxl-multimodal) is pure Python on stock PyTorch; the
 * arithmetic itself lives in a third-party dependency that is NOT vendored under
 * /root/reference: PyTorch ATen, pinned `torch==2.8.0` (reference
 * requirements.txt:54; this container has 2.10.0+rocm7.0).  Each function below
 * restates the published definition of the torch.nn op the reference calls and
 * cites the reference call site it stands for.  Parity is pinned by
 * the .npz fixtures under tests/golden/, generated here by importing the reference modules
 * (tests/golden/make_golden.py) — see tests/test_oracle_golden.py.
 *
 * All tensors are contiguous float32, activations NCL.  Reductions accumulate in
 * double and round once, so the oracle is at least as accurate as the fp32
 * reference (which agrees with its own fp64 evaluation to 9.5e-7).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---- nn.Conv1d(in,out,k,padding=k//2), stride 1, dilation 1, groups 1 --------
 * reference: src/models/ecg_cnn.py:13, src/models/ecg_multimodal.py:9
 * y[n,co,t] = b[co] + sum_ci sum_k w[co,ci,k] * x[n,ci,t+k-pad]   (zero padding) */
ORC_API void orc_conv1d_fwd(const float *x, const float *w, const float *b, float *y,
                            int N, int Ci, int Co, int L, int K, int pad)
{
    int Lo = L + 2 * pad - K + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co) {
            float *yr = y + ((size_t)n * Co + co) * Lo;
            for (int t = 0; t < Lo; ++t) {
                double acc = b ? (double)b[co] : 0.0;
                for (int ci = 0; ci < Ci; ++ci) {
                    const float *xr = x + ((size_t)n * Ci + ci) * L;
                    const float *wr = w + ((size_t)co * Ci + ci) * K;
                    for (int k = 0; k < K; ++k) {
                        int s = t + k - pad;
                        if (s >= 0 && s < L) acc += (double)wr[k] * (double)xr[s];
                    }
                }
                yr[t] = (float)acc;
            }
        }
}

/* ---- autograd of the above w.r.t. its input (loss.backward(), src/training/loop.py:33)
 * dx[n,ci,s] = sum_co sum_k dy[n,co,s-k+pad] * w[co,ci,k] */
ORC_API void orc_conv1d_bwd_data(const float *dy, const float *w, float *dx,
                                 int N, int Ci, int Co, int L, int K, int pad)
{
    int Lo = L + 2 * pad - K + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < Ci; ++ci) {
            float *dxr = dx + ((size_t)n * Ci + ci) * L;
            for (int s = 0; s < L; ++s) {
                double acc = 0.0;
                for (int co = 0; co < Co; ++co) {
                    const float *dyr = dy + ((size_t)n * Co + co) * Lo;
                    const float *wr = w + ((size_t)co * Ci + ci) * K;
                    for (int k = 0; k < K; ++k) {
                        int t = s - k + pad;
                        if (t >= 0 && t < Lo) acc += (double)dyr[t] * (double)wr[k];
                    }
                }
                dxr[s] = (float)acc;
            }
        }
}

/* ---- autograd w.r.t. weight and bias
 * dw[co,ci,k] = sum_n sum_t dy[n,co,t] * x[n,ci,t+k-pad];  db[co] = sum_n sum_t dy[n,co,t] */
ORC_API void orc_conv1d_bwd_weight(const float *dy, const float *x, float *dw, float *db,
                                   int N, int Ci, int Co, int L, int K, int pad)
{
    int Lo = L + 2 * pad - K + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int k = 0; k < K; ++k) {
                double acc = 0.0;
                for (int n = 0; n < N; ++n) {
                    const float *dyr = dy + ((size_t)n * Co + co) * Lo;
                    const float *xr = x + ((size_t)n * Ci + ci) * L;
                    for (int t = 0; t < Lo; ++t) {
                        int s = t + k - pad;
                        if (s >= 0 && s < L) acc += (double)dyr[t] * (double)xr[s];
                    }
                }
                dw[((size_t)co * Ci + ci) * K + k] = (float)acc;
            }
    if (db) {
        for (int co = 0; co < Co; ++co) {
            double acc = 0.0;
            for (int n = 0; n < N; ++n) {
                const float *dyr = dy + ((size_t)n * Co + co) * Lo;
                for (int t = 0; t < Lo; ++t) acc += dyr[t];
            }
            db[co] = (float)acc;
        }
    }
}

/* ---- nn.BatchNorm1d(C) training-mode statistics (src/models/ecg_cnn.py:14)
 * mean_c, biased var_c over (N,L); invstd = 1/sqrt(var+eps);
 * running_mean = (1-m)*running_mean + m*mean; running_var uses the UNBIASED var;
 * num_batches_tracked += 1. */
ORC_API void orc_bn_stats(const float *y, float *mean, float *invstd,
                          float *running_mean, float *running_var, int64_t *nbt,
                          int N, int C, int L, float momentum, float eps)
{
    double cnt = (double)N * L;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int n = 0; n < N; ++n) {
            const float *r = y + ((size_t)n * C + c) * L;
            for (int t = 0; t < L; ++t) s += r[t];
        }
        double mu = s / cnt, v = 0.0;
        for (int n = 0; n < N; ++n) {
            const float *r = y + ((size_t)n * C + c) * L;
            for (int t = 0; t < L; ++t) { double d = r[t] - mu; v += d * d; }
        }
        double var = v / cnt;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            double unb = cnt > 1 ? v / (cnt - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    }
    if (nbt) *nbt += 1;
}

/* BN affine exactly as the HIP kernels and this oracle both evaluate it:
 * a = (y - mean) * (invstd * gamma) + beta  (single precision, fused multiply-add) */
static inline float bn_apply1(float y, float mean, float scale, float beta)
{
    return fmaf(y - mean, scale, beta);
}

/* ---- BatchNorm1d (given mean/invstd: batch stats in train, running stats in eval) */
ORC_API void orc_bn_apply(const float *y, const float *gamma, const float *beta,
                          const float *mean, const float *invstd, float *out,
                          int N, int C, int L)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            float sc = invstd[c] * gamma[c];
            const float *r = y + ((size_t)n * C + c) * L;
            float *o = out + ((size_t)n * C + c) * L;
            for (int t = 0; t < L; ++t) o[t] = bn_apply1(r[t], mean[c], sc, beta[c]);
        }
}

/* ---- fused BatchNorm1d -> ReLU(inplace) -> MaxPool1d(2)
 * reference: src/models/ecg_cnn.py:14-16.  Lp = floor(L/2): an odd tail sample is dropped. */
ORC_API void orc_bn_relu_pool_fwd(const float *y, const float *gamma, const float *beta,
                                  const float *mean, const float *invstd, float *p,
                                  int N, int C, int L)
{
    int Lp = L / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            float sc = invstd[c] * gamma[c];
            const float *r = y + ((size_t)n * C + c) * L;
            float *o = p + ((size_t)n * C + c) * Lp;
            for (int j = 0; j < Lp; ++j) {
                float a0 = bn_apply1(r[2 * j], mean[c], sc, beta[c]);
                float a1 = bn_apply1(r[2 * j + 1], mean[c], sc, beta[c]);
                float m = a1 > a0 ? a1 : a0;
                o[j] = m > 0.f ? m : 0.f;
            }
        }
}

/* ---- backward of the fused block tail.
 * max-pool routes dp to the arg-max of the pair (first element on a tie —
 * max_pool1d keeps the first maximal index), ReLU passes it only where the output
 * is > 0, then native_batch_norm_backward (training):
 *   dbeta = sum da, dgamma = sum da*xhat,
 *   dy = gamma*invstd * (da - dbeta/M - xhat*dgamma/M),  M = N*L.
 * In eval mode (train=0): dy = da*gamma*invstd, dgamma/dbeta as above. */
ORC_API void orc_bn_relu_pool_bwd(const float *y, const float *dp, const float *gamma,
                                  const float *beta, const float *mean, const float *invstd,
                                  float *dy, float *dgamma, float *dbeta,
                                  int N, int C, int L, int train)
{
    int Lp = L / 2;
    double M = (double)N * L;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        float sc = invstd[c] * gamma[c];
        double sda = 0.0, sdax = 0.0;
        for (int n = 0; n < N; ++n) {
            const float *r = y + ((size_t)n * C + c) * L;
            const float *g = dp + ((size_t)n * C + c) * Lp;
            for (int j = 0; j < Lp; ++j) {
                float a0 = bn_apply1(r[2 * j], mean[c], sc, beta[c]);
                float a1 = bn_apply1(r[2 * j + 1], mean[c], sc, beta[c]);
                int am = a1 > a0 ? 1 : 0;
                float m = am ? a1 : a0;
                if (m > 0.f) {
                    double xh = ((double)r[2 * j + am] - mean[c]) * invstd[c];
                    sda += g[j];
                    sdax += (double)g[j] * xh;
                }
            }
        }
        if (dgamma) dgamma[c] = (float)sdax;
        if (dbeta) dbeta[c] = (float)sda;
        double k1 = train ? sda / M : 0.0, k2 = train ? sdax / M : 0.0;
        for (int n = 0; n < N; ++n) {
            const float *r = y + ((size_t)n * C + c) * L;
            const float *g = dp + ((size_t)n * C + c) * Lp;
            float *d = dy + ((size_t)n * C + c) * L;
            for (int t = 0; t < L; ++t) {
                double da = 0.0;
                int j = t >> 1;
                if (j < Lp) {
                    float a0 = bn_apply1(r[2 * j], mean[c], sc, beta[c]);
                    float a1 = bn_apply1(r[2 * j + 1], mean[c], sc, beta[c]);
                    int am = a1 > a0 ? 1 : 0;
                    float m = am ? a1 : a0;
                    if (m > 0.f && (t & 1) == am) da = g[j];
                }
                double xh = ((double)r[t] - mean[c]) * invstd[c];
                d[t] = (float)((double)gamma[c] * invstd[c] * (da - k1 - xh * k2));
            }
        }
    }
}

/* ---- nn.AdaptiveAvgPool1d(1) + squeeze(-1)  (src/models/ecg_cnn.py:46,62) */
ORC_API void orc_gap_fwd(const float *p, float *g, int N, int C, int L)
{
    for (size_t i = 0; i < (size_t)N * C; ++i) {
        double s = 0.0;
        for (int t = 0; t < L; ++t) s += p[i * L + t];
        g[i] = (float)(s / L);
    }
}
ORC_API void orc_gap_bwd(const float *dg, float *dp, int N, int C, int L)
{
    for (size_t i = 0; i < (size_t)N * C; ++i)
        for (int t = 0; t < L; ++t) dp[i * L + t] = dg[i] / (float)L;
}

/* ---- nn.Linear (src/models/ecg_cnn.py:47,50; ecg_multimodal.py:35,52,54,85,86)
 * y[m,o] = b[o] + sum_i x[m,i]*w[o,i]; optional ReLU (ecg_multimodal.py:53,55) */
ORC_API void orc_linear_fwd(const float *x, const float *w, const float *b, float *y,
                            int M, int In, int Out, int relu)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int o = 0; o < Out; ++o) {
            double acc = b ? (double)b[o] : 0.0;
            for (int i = 0; i < In; ++i) acc += (double)x[(size_t)m * In + i] * w[(size_t)o * In + i];
            float v = (float)acc;
            y[(size_t)m * Out + o] = (relu && v < 0.f) ? 0.f : v;
        }
}
/* dy is the gradient w.r.t. the (post-ReLU if relu) output y. */
ORC_API void orc_linear_bwd(const float *x, const float *w, const float *y, const float *dy,
                            float *dx, float *dw, float *db, int M, int In, int Out, int relu)
{
    float *g = (float *)malloc(sizeof(float) * (size_t)M * Out);
    for (size_t i = 0; i < (size_t)M * Out; ++i) g[i] = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
    if (dx)
        for (int m = 0; m < M; ++m)
            for (int i = 0; i < In; ++i) {
                double acc = 0.0;
                for (int o = 0; o < Out; ++o) acc += (double)g[(size_t)m * Out + o] * w[(size_t)o * In + i];
                dx[(size_t)m * In + i] = (float)acc;
            }
    if (dw)
        for (int o = 0; o < Out; ++o)
            for (int i = 0; i < In; ++i) {
                double acc = 0.0;
                for (int m = 0; m < M; ++m) acc += (double)g[(size_t)m * Out + o] * x[(size_t)m * In + i];
                dw[(size_t)o * In + i] = (float)acc;
            }
    if (db)
        for (int o = 0; o < Out; ++o) {
            double acc = 0.0;
            for (int m = 0; m < M; ++m) acc += g[(size_t)m * Out + o];
            db[o] = (float)acc;
        }
    free(g);
}

/* ---- FiLM fusion (src/models/ecg_multimodal.py:92-96)
 * film[m, 0:F] = gamma-raw, film[m, F:2F] = beta (torch.chunk(…, 2, dim=-1));
 * zc = (1 + tanh(gamma_raw)) * z + beta */
ORC_API void orc_film_fwd(const float *z, const float *film, float *zc, int M, int F)
{
    for (int m = 0; m < M; ++m)
        for (int f = 0; f < F; ++f) {
            float g = 1.0f + tanhf(film[(size_t)m * 2 * F + f]);
            zc[(size_t)m * F + f] = g * z[(size_t)m * F + f] + film[(size_t)m * 2 * F + F + f];
        }
}
ORC_API void orc_film_bwd(const float *z, const float *film, const float *dzc,
                          float *dz, float *dfilm, int M, int F)
{
    for (int m = 0; m < M; ++m)
        for (int f = 0; f < F; ++f) {
            float th = tanhf(film[(size_t)m * 2 * F + f]);
            float d = dzc[(size_t)m * F + f];
            dz[(size_t)m * F + f] = d * (1.0f + th);
            dfilm[(size_t)m * 2 * F + f] = d * z[(size_t)m * F + f] * (1.0f - th * th);
            dfilm[(size_t)m * 2 * F + F + f] = d;
        }
}

/* ---- F.binary_cross_entropy_with_logits / BCEWithLogitsLoss(), reduction='mean'
 * (src/training/loop.py:32, loop_demo.py:10,33)
 * loss = mean( max(x,0) - x*y + log1p(exp(-|x|)) );  dx = (sigmoid(x)-y)/numel */
ORC_API float orc_bce_fwd(const float *x, const float *y, int numel)
{
    double s = 0.0;
    for (int i = 0; i < numel; ++i) {
        double xi = x[i];
        s += (xi > 0 ? xi : 0.0) - xi * y[i] + log1p(exp(-fabs(xi)));
    }
    return (float)(s / numel);
}
ORC_API void orc_bce_bwd(const float *x, const float *y, float *dx, int numel, float gscale)
{
    for (int i = 0; i < numel; ++i) {
        double sg = 1.0 / (1.0 + exp(-(double)x[i]));
        dx[i] = (float)((sg - y[i]) / numel * gscale);
    }
}

/* ---- torch.optim.AdamW single-tensor step, defaults betas=(0.9,0.999), eps=1e-8,
 * amsgrad=False, maximize=False (scripts/03_train_ecg_baseline.py:130-133).
 * p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
 * p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps),  bc = 1 - beta^step */
ORC_API void orc_adamw(float *p, const float *g, float *m, float *v, size_t n, int step,
                       float lr, float b1, float b2, float eps, float wd)
{
    double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    double step_size = lr / bc1, bc2s = sqrt(bc2);
    for (size_t i = 0; i < n; ++i) {
        double pi = (double)p[i] * (1.0 - (double)lr * wd);
        double mi = (double)b1 * m[i] + (1.0 - b1) * g[i];
        double vi = (double)b2 * v[i] + (1.0 - b2) * (double)g[i] * g[i];
        m[i] = (float)mi; v[i] = (float)vi;
        double denom = sqrt(vi) / bc2s + eps;
        p[i] = (float)(pi - step_size * mi / denom);
    }
}
