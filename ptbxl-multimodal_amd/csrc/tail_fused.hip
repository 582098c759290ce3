// tail_fused.hip — the whole network tail in three launches instead of ~25:
//   ecg_tail_fwd          proj Linear, demographic MLP (2 x Linear+ReLU), film_gen Linear, FiLM
//                         fusion and the head Linear, for SB samples per workgroup;
//   ecg_tail_bwd_chain    the per-sample backward chain (d logits -> d zc -> d z / d film ->
//                         d h2 -> d h1 -> d x_demo, d g), emitting the masked gradient matrices;
//   ecg_linear_wgrad_grouped   every weight / bias gradient of the tail (5 small GEMMs over the
//                         batch dimension) as ONE grouped launch.
// All tensors are <= [B, 512]; the arithmetic is negligible (53 MFLOP at B=256), so the design
// target is launch count and dependent-load latency, not FLOP/s.
//
// Replaces the ATen addmm/relu/tanh/mul/add chain behind
//   ECGCNN.proj/head                  reference src/models/ecg_cnn.py:47-50,63-64
//   ECGBackbone.proj, DemoEncoder.mlp, film_gen, FiLM, head
//                                     reference src/models/ecg_multimodal.py:35,51-59,85-98
#include "common.h"

namespace ecg {

constexpr int kSB = 2;   // samples per workgroup
typedef float vsb __attribute__((ext_vector_type(kSB)));
constexpr int kLB = 32;  // global loads issued back-to-back before their first use

// acc[s] += sum_{i<n} v_s[i][s] * w[i * stride]   (v_s in LDS as [n][kSB], w walks global memory).
// These loops are chains of dependent-looking L2 loads; hipcc keeps each load next to its FMA and
// waits vmcnt(0) per element unless the loads are first gathered into a register batch.
__device__ __forceinline__ void dot_rows(float (&acc)[kSB], const float *__restrict__ v_s,
                                         const float *__restrict__ w, size_t stride, int n) {
    int i = 0;
    for (; i + kLB <= n; i += kLB) {
        float wv[kLB];
#pragma unroll
        for (int u = 0; u < kLB; ++u) wv[u] = w[(size_t)(i + u) * stride];
#pragma unroll
        for (int u = 0; u < kLB; ++u) {
            const vsb v = *reinterpret_cast<const vsb *>(v_s + (i + u) * kSB);
#pragma unroll
            for (int q = 0; q < kSB; ++q) acc[q] = __fmaf_rn(v[q], wv[u], acc[q]);
        }
    }
    for (; i < n; ++i) {
        const float wv = w[(size_t)i * stride];
        const vsb v = *reinterpret_cast<const vsb *>(v_s + i * kSB);
#pragma unroll
        for (int q = 0; q < kSB; ++q) acc[q] = __fmaf_rn(v[q], wv, acc[q]);
    }
}

// acc += sum_{i<n} v_s[i][s] * w[i * stride]  for ONE sample column s
__device__ __forceinline__ float dot_col(const float *__restrict__ v_s, int s,
                                         const float *__restrict__ w, size_t stride, int n) {
    float acc = 0.f;
    int i = 0;
    for (; i + kLB <= n; i += kLB) {
        float wv[kLB];
#pragma unroll
        for (int u = 0; u < kLB; ++u) wv[u] = w[(size_t)(i + u) * stride];
#pragma unroll
        for (int u = 0; u < kLB; ++u) acc = __fmaf_rn(v_s[(i + u) * kSB + s], wv[u], acc);
    }
    for (; i < n; ++i) acc = __fmaf_rn(v_s[i * kSB + s], w[(size_t)i * stride], acc);
    return acc;
}

struct TailFwdArgs {
    const float *g, *xd;
    const float *WpT, *bp, *W0, *b0, *W2, *b2, *WfT, *bf, *Wh, *bh;   // W0 [H1][D], W2 [H][H1]: state_dict layout
    float *z, *h1, *h2, *film, *zc, *logits;
    int M, F0, F, D, H1, H, C;
};

// dynamic LDS: g_s[F0][SB] | xd_s[D][SB] | h1_s[H1][SB] | h2_s[H][SB] | zc_s[F][SB]
__global__ __launch_bounds__(256) void tail_fwd_kernel(TailFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *g_s = smem, *xd_s = g_s + a.F0 * kSB, *h1_s = xd_s + a.D * kSB, *h2_s = h1_s + a.H1 * kSB,
          *zc_s = h2_s + a.H * kSB;
    const int tid = threadIdx.x, m0 = blockIdx.x * kSB;
    const bool demo = a.xd != nullptr;

    for (int e = tid; e < a.F0 * kSB; e += 256) {
        int s = e / a.F0, i = e - s * a.F0;
        g_s[i * kSB + s] = (m0 + s < a.M) ? a.g[(size_t)(m0 + s) * a.F0 + i] : 0.f;
    }
    if (demo)
        for (int e = tid; e < a.D * kSB; e += 256) {
            int s = e / a.D, i = e - s * a.D;
            xd_s[i * kSB + s] = (m0 + s < a.M) ? a.xd[(size_t)(m0 + s) * a.D + i] : 0.f;
        }
    __syncthreads();

    if (demo) {
        // h1 = relu(xd W0^T + b0), h2 = relu(h1 W2^T + b2): thread <-> (sample, unit)
        for (int e = tid; e < a.H1 * kSB; e += 256) {
            int s = e / a.H1, j = e - s * a.H1;
            float acc = a.b0[j] + dot_col(xd_s, s, a.W0 + (size_t)j * a.D, 1, a.D);
            acc = acc > 0.f ? acc : 0.f;
            h1_s[j * kSB + s] = acc;
            if (m0 + s < a.M) a.h1[(size_t)(m0 + s) * a.H1 + j] = acc;
        }
        __syncthreads();
        for (int e = tid; e < a.H * kSB; e += 256) {
            int s = e / a.H, j = e - s * a.H;
            float acc = a.b2[j] + dot_col(h1_s, s, a.W2 + (size_t)j * a.H1, 1, a.H1);
            acc = acc > 0.f ? acc : 0.f;
            h2_s[j * kSB + s] = acc;
            if (m0 + s < a.M) a.h2[(size_t)(m0 + s) * a.H + j] = acc;
        }
        __syncthreads();
    }

    // z = g Wp^T + bp;  film = h2 Wf^T + bf;  zc = (1 + tanh(film_g)) * z + film_b: thread <-> feature o
    for (int o = tid; o < a.F; o += 256) {
        float zacc[kSB];
        const float bpo = a.bp[o];
#pragma unroll
        for (int s = 0; s < kSB; ++s) zacc[s] = bpo;
        dot_rows(zacc, g_s, a.WpT + o, (size_t)a.F, a.F0);
        float zc[kSB];
        if (demo) {
            float fg[kSB], fb[kSB];
            const float bg = a.bf[o], bb = a.bf[a.F + o];
#pragma unroll
            for (int s = 0; s < kSB; ++s) { fg[s] = bg; fb[s] = bb; }
            dot_rows(fg, h2_s, a.WfT + o, (size_t)2 * a.F, a.H);
            dot_rows(fb, h2_s, a.WfT + a.F + o, (size_t)2 * a.F, a.H);
#pragma unroll
            for (int s = 0; s < kSB; ++s) {
                zc[s] = __fmaf_rn(1.0f + tanhf(fg[s]), zacc[s], fb[s]);
                if (m0 + s < a.M) {
                    a.film[(size_t)(m0 + s) * 2 * a.F + o] = fg[s];
                    a.film[(size_t)(m0 + s) * 2 * a.F + a.F + o] = fb[s];
                    a.zc[(size_t)(m0 + s) * a.F + o] = zc[s];
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < kSB; ++s) zc[s] = zacc[s];
        }
#pragma unroll
        for (int s = 0; s < kSB; ++s) {
            zc_s[o * kSB + s] = zc[s];
            if (m0 + s < a.M) a.z[(size_t)(m0 + s) * a.F + o] = zacc[s];
        }
    }
    __syncthreads();

    // logits = zc Wh^T + bh: one wave per (sample, label) pair
    const int wave = tid >> 6, lane = tid & 63;
    for (int p = wave; p < kSB * a.C; p += 4) {
        const int s = p / a.C, c = p - s * a.C;
        float acc = 0.f;
        for (int o = lane; o < a.F; o += 64) acc = __fmaf_rn(zc_s[o * kSB + s], a.Wh[(size_t)c * a.F + o], acc);
        acc = wave_sum(acc);
        if (lane == 0 && m0 + s < a.M) a.logits[(size_t)(m0 + s) * a.C + c] = acc + a.bh[c];
    }
}

struct TailBwdArgs {
    const float *dlogits, *dz_extra, *z, *h1, *h2, *film;
    const float *Wp, *W0, *W2, *Wf, *Wh;
    float *dzc, *dz, *dfilm, *dh2m, *dh1m, *dg, *dxd;
    int M, F0, F, D, H1, H, C, demo;
};

// dynamic LDS: dlog_s[C][SB] | dz_s[F][SB] | dfilm_s[2F][SB] | dh2_s[H][SB] | dh1_s[H1][SB]
__global__ __launch_bounds__(256) void tail_bwd_chain_kernel(TailBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *dlog_s = smem, *dz_s = dlog_s + a.C * kSB, *dfilm_s = dz_s + a.F * kSB,
          *dh2_s = dfilm_s + 2 * a.F * kSB, *dh1_s = dh2_s + a.H * kSB;
    const int tid = threadIdx.x, m0 = blockIdx.x * kSB;

    for (int e = tid; e < a.C * kSB; e += 256) {
        int s = e / a.C, c = e - s * a.C;
        dlog_s[c * kSB + s] = (m0 + s < a.M) ? a.dlogits[(size_t)(m0 + s) * a.C + c] : 0.f;
    }
    __syncthreads();

    // d zc = d logits Wh;  FiLM backward: d z = d zc (1+th), d film_g = d zc z (1-th^2), d film_b = d zc
    for (int o = tid; o < a.F; o += 256) {
        float d[kSB];
#pragma unroll
        for (int q = 0; q < kSB; ++q) d[q] = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float w = a.Wh[(size_t)c * a.F + o];
            const vsb dl = *reinterpret_cast<const vsb *>(dlog_s + c * kSB);
#pragma unroll
            for (int q = 0; q < kSB; ++q) d[q] = __fmaf_rn(dl[q], w, d[q]);
        }
#pragma unroll
        for (int s = 0; s < kSB; ++s) {
            const bool live = m0 + s < a.M;
            const size_t m = live ? (size_t)(m0 + s) : 0;
            float dzv = d[s];
            if (a.demo) {
                const float th = tanhf(a.film[m * 2 * a.F + o]);
                const float zv = a.z[m * a.F + o];
                const float dfg = d[s] * zv * (1.0f - th * th);
                dzv = d[s] * (1.0f + th);
                dfilm_s[o * kSB + s] = live ? dfg : 0.f;
                dfilm_s[(a.F + o) * kSB + s] = live ? d[s] : 0.f;
                if (live) {
                    a.dzc[m * a.F + o] = d[s];
                    a.dfilm[m * 2 * a.F + o] = dfg;
                    a.dfilm[m * 2 * a.F + a.F + o] = d[s];
                }
            }
            if (a.dz_extra && live) dzv += a.dz_extra[m * a.F + o];
            dz_s[o * kSB + s] = live ? dzv : 0.f;
            if (live) a.dz[m * a.F + o] = dzv;
        }
    }
    __syncthreads();

    if (a.demo) {
        // d h2 = (d film Wf) masked by h2 > 0
        for (int e = tid; e < a.H * kSB; e += 256) {
            int s = e / a.H, i = e - s * a.H;
            float acc = dot_col(dfilm_s, s, a.Wf + i, (size_t)a.H, 2 * a.F);
            const bool live = m0 + s < a.M;
            if (!(live && a.h2[(size_t)(m0 + s) * a.H + i] > 0.f)) acc = 0.f;
            dh2_s[i * kSB + s] = acc;
            if (live) a.dh2m[(size_t)(m0 + s) * a.H + i] = acc;
        }
        __syncthreads();
        for (int e = tid; e < a.H1 * kSB; e += 256) {
            int s = e / a.H1, i = e - s * a.H1;
            float acc = dot_col(dh2_s, s, a.W2 + i, (size_t)a.H1, a.H);
            const bool live = m0 + s < a.M;
            if (!(live && a.h1[(size_t)(m0 + s) * a.H1 + i] > 0.f)) acc = 0.f;
            dh1_s[i * kSB + s] = acc;
            if (live) a.dh1m[(size_t)(m0 + s) * a.H1 + i] = acc;
        }
        __syncthreads();
        if (a.dxd)
            for (int e = tid; e < a.D * kSB; e += 256) {
                int s = e / a.D, i = e - s * a.D;
                float acc = dot_col(dh1_s, s, a.W0 + i, (size_t)a.D, a.H1);
                if (m0 + s < a.M) a.dxd[(size_t)(m0 + s) * a.D + i] = acc;
            }
    }

    // d g = d z Wp: thread <-> input feature i
    for (int i = tid; i < a.F0; i += 256) {
        float acc[kSB];
#pragma unroll
        for (int q = 0; q < kSB; ++q) acc[q] = 0.f;
        dot_rows(acc, dz_s, a.Wp + i, (size_t)a.F0, a.F);
#pragma unroll
        for (int s = 0; s < kSB; ++s)
            if (m0 + s < a.M) a.dg[(size_t)(m0 + s) * a.F0 + i] = acc[s];
    }
}

// ---------------------------------------------------------------------------------------
// grouped weight gradient: for each problem p, dW_p[o][i] = sum_m G_p[m][o] X_p[m][i],
// db_p[o] = sum_m G_p[m][o].  32x32 output tile per workgroup.
// ---------------------------------------------------------------------------------------
constexpr int kMaxProb = 8;
struct WgProb {
    const float *G, *X;
    float *dW, *db;
    int Out, In, tile0, tiles_i;
};
struct WgArgs {
    WgProb p[kMaxProb];
    int count, M;
};

// One 32 x 32 output tile per workgroup on the fp32 matrix core (v_mfma_f32_32x32x2_f32: an exact fp32 fma chain over
// the samples, two per instruction).  The four waves split the samples into four contiguous ranges and are summed
// through LDS in wave order (fixed order: deterministic).  Operands come straight from global memory — lane (o | i, k)
// holds G[m + k][o0 + o] and X[m + k][i0 + i], 128 contiguous bytes per half wave — in batches of 16 instructions whose
// loads are all in flight before the first MFMA issues: the kernel is a latency chain (72 workgroups for the whole
// tail), the VALU version with its 32-sample LDS chunks took 12.4 us at B = 256, this one takes ~6.
typedef float f32x16t __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void linear_wgrad_grouped_kernel(WgArgs a) {
    __shared__ float red[3][16][64];
    __shared__ float bred[4][32];
    int pi = 0;
#pragma unroll
    for (int q = 1; q < kMaxProb; ++q)
        if (q < a.count && (int)blockIdx.x >= a.p[q].tile0) pi = q;
    const WgProb pr = a.p[pi];
    const int tile = blockIdx.x - pr.tile0;
    const int o0 = (tile / pr.tiles_i) * 32, i0 = (tile % pr.tiles_i) * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int steps = (a.M + 1) >> 1, per = (steps + 3) >> 2;           // MFMA steps (2 samples each) in all / per wave
    const int s_begin = wave * per, s_end = min(steps, s_begin + per);
    const bool ok_o = o0 + l31 < pr.Out, ok_i = i0 + l31 < pr.In;
    const float *gp = pr.G + min(o0 + l31, pr.Out - 1), *xp = pr.X + min(i0 + l31, pr.In - 1);
    f32x16t acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float bacc = 0.f;
    constexpr int kBatch = 16;
    for (int s0 = s_begin; s0 < s_end; s0 += kBatch) {
        float gv[kBatch], xv[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {                  // unconditional, clamped loads; masked below
            const int m = 2 * (s0 + u) + half, mc = min(m, a.M - 1);
            gv[u] = gp[(size_t)mc * pr.Out];
            xv[u] = xp[(size_t)mc * pr.In];
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const bool live = (s0 + u < s_end) && (2 * (s0 + u) + half < a.M);
            const float g = (live && ok_o) ? gv[u] : 0.f, x = (live && ok_i) ? xv[u] : 0.f;
            bacc += g;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g, x, acc, 0, 0, 0);
        }
    }
    bacc += __shfl_xor(bacc, 32, 64);                       // the two samples of a step
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) red[wave - 1][q][lane] = acc[q];
    }
    if (half == 0) bred[wave][l31] = bacc;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] += red[w][q][lane];
#pragma unroll
        for (int q = 0; q < 16; ++q) {                      // D[row = o][col = i]: row (q & 3) + 8 (q >> 2) + 4 half, col l31
            const int o = o0 + (q & 3) + 8 * (q >> 2) + 4 * half, i = i0 + l31;
            if (o < pr.Out && i < pr.In) pr.dW[(size_t)o * pr.In + i] = acc[q];
        }
        if (i0 == 0 && half == 0 && pr.db && ok_o)
            pr.db[o0 + l31] = ((bred[0][l31] + bred[1][l31]) + bred[2][l31]) + bred[3][l31];
    }
}

__global__ void transpose_kernel(const float *__restrict__ w, float *__restrict__ wT, int rows,
                                 int cols) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < rows && c0 + tx < cols) t[r][tx] = w[(size_t)(r0 + r) * cols + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < cols && r0 + tx < rows) wT[(size_t)(c0 + c) * rows + r0 + tx] = t[tx][c];
}

}  // namespace ecg

using namespace ecg;

ECG_API int ecg_transpose(const float *w, float *wT, int rows, int cols, ecg_stream_t stream) {
    ECG_REQUIRE(w && wT && rows > 0 && cols > 0, "transpose: bad argument");
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0,
                       as_stream(stream), w, wT, rows, cols);
    return check_launch("transpose_kernel");
}

static int check_tail_dims(const char *who, int M, int F0, int F, int D, int H1, int H, int C, int demo) {
    ECG_REQUIRE(M > 0 && F0 > 0 && F > 0 && C > 0, "%s: M=%d F0=%d F=%d C=%d must be > 0", who, M, F0, F, C);
    ECG_REQUIRE(!demo || (D > 0 && H1 > 0 && H > 0), "%s: demographic path needs D, H1, H > 0", who);
    ECG_REQUIRE(F0 <= 4096 && F <= 2048 && H1 <= 1024 && H <= 1024 && D <= 256 && C <= 1024,
                "%s: layer too wide for the fused tail (F0=%d F=%d H1=%d H=%d D=%d C=%d)", who, F0, F, H1, H, D, C);
    return ECG_OK;
}

ECG_API int ecg_tail_fwd(const float *g, const float *xd, const float *WpT, const float *bp,
                         const float *W0, const float *b0, const float *W2, const float *b2,
                         const float *WfT, const float *bf, const float *Wh, const float *bh,
                         float *z, float *h1, float *h2, float *film, float *zc, float *logits,
                         int M, int F0, int F, int D, int H1, int H, int C, ecg_stream_t stream) {
    const int demo = xd != nullptr;
    int rc = check_tail_dims("tail_fwd", M, F0, F, D, H1, H, C, demo);
    if (rc) return rc;
    ECG_REQUIRE(g && WpT && bp && Wh && bh && z && logits, "tail_fwd: null pointer");
    ECG_REQUIRE(!demo || (W0 && b0 && W2 && b2 && WfT && bf && h1 && h2 && film && zc),
                "tail_fwd: null pointer on the demographic path");
    TailFwdArgs a{g, xd, WpT, bp, W0, b0, W2, b2, WfT, bf, Wh, bh, z, h1, h2, film, zc, logits,
                  M, F0, F, demo ? D : 0, demo ? H1 : 0, demo ? H : 0, C};
    size_t lds = sizeof(float) * kSB * (size_t)(F0 + a.D + a.H1 + a.H + F);
    ECG_REQUIRE(lds <= 64 * 1024, "tail_fwd: %zu bytes of LDS needed", lds);
    hipLaunchKernelGGL(tail_fwd_kernel, dim3(cdiv(M, kSB)), dim3(256), lds, as_stream(stream), a);
    return check_launch("tail_fwd_kernel");
}

ECG_API int ecg_tail_bwd_chain(const float *dlogits, const float *dz_extra, const float *z,
                               const float *h1, const float *h2, const float *film,
                               const float *Wp, const float *W0, const float *W2, const float *Wf,
                               const float *Wh, float *dzc, float *dz, float *dfilm, float *dh2m,
                               float *dh1m, float *dg, float *dxd, int M, int F0, int F, int D,
                               int H1, int H, int C, int demo, ecg_stream_t stream) {
    int rc = check_tail_dims("tail_bwd_chain", M, F0, F, D, H1, H, C, demo);
    if (rc) return rc;
    ECG_REQUIRE(dlogits && Wp && Wh && dz && dg, "tail_bwd_chain: null pointer");
    ECG_REQUIRE(!demo || (z && h1 && h2 && film && W0 && W2 && Wf && dzc && dfilm && dh2m && dh1m),
                "tail_bwd_chain: null pointer on the demographic path");
    TailBwdArgs a{dlogits, dz_extra, z, h1, h2, film, Wp, W0, W2, Wf, Wh, dzc, dz, dfilm, dh2m,
                  dh1m, dg, dxd, M, F0, F, demo ? D : 0, demo ? H1 : 0, demo ? H : 0, C, demo};
    size_t lds = sizeof(float) * kSB * (size_t)(C + F + 2 * F + a.H + a.H1);
    ECG_REQUIRE(lds <= 64 * 1024, "tail_bwd_chain: %zu bytes of LDS needed", lds);
    hipLaunchKernelGGL(tail_bwd_chain_kernel, dim3(cdiv(M, kSB)), dim3(256), lds, as_stream(stream), a);
    return check_launch("tail_bwd_chain_kernel");
}

ECG_API int ecg_linear_wgrad_grouped(const float *const *G, const float *const *X, float *const *dW,
                                     float *const *db, const int *Out, const int *In, int count,
                                     int M, ecg_stream_t stream) {
    ECG_REQUIRE(G && X && dW && db && Out && In, "linear_wgrad_grouped: null table");
    ECG_REQUIRE(count >= 1 && count <= kMaxProb, "linear_wgrad_grouped: count=%d outside [1,%d]", count, kMaxProb);
    ECG_REQUIRE(M > 0, "linear_wgrad_grouped: M=%d", M);
    WgArgs a;
    a.count = count; a.M = M;
    int tiles = 0;
    for (int q = 0; q < count; ++q) {
        ECG_REQUIRE(G[q] && X[q] && dW[q] && Out[q] > 0 && In[q] > 0, "linear_wgrad_grouped: problem %d is malformed", q);
        a.p[q] = WgProb{G[q], X[q], dW[q], db[q], Out[q], In[q], tiles, cdiv(In[q], 32)};
        tiles += cdiv(Out[q], 32) * cdiv(In[q], 32);
    }
    for (int q = count; q < kMaxProb; ++q) a.p[q] = WgProb{nullptr, nullptr, nullptr, nullptr, 0, 0, 1 << 30, 1};
    hipLaunchKernelGGL(linear_wgrad_grouped_kernel, dim3(tiles), dim3(256), 0, as_stream(stream), a);
    return check_launch("linear_wgrad_grouped_kernel");
}
