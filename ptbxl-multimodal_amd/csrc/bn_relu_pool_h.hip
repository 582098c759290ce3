// bn_relu_pool_h.hip — the BatchNorm + ReLU + MaxPool(2) passes of the mixed-precision train step on bf16 [N][C][ld]
// tensors ONLY (round 4): 16 bytes per lane in and out, no "n16" copies.
//
// Rounds 2-3 ran these passes organised around the weight gradient's operand layout (thread <-> 16 samples at one position:
// 4-byte loads from 16 rows, a 32-byte n16 store plus a second copy as bf16 rows): 0.49 GB of the 2.3 GB per step of
// BASELINE config 5 were the second copies, and the passes ran at 3.5-5.3 TB/s.  With the time-on-K weight gradient
// (conv1d_wgrad_bf16_tk.hip) every consumer reads plain rows, so these kernels stream rows:
//   ecg_bn_stats_relu_pool_fwd_h   y bf16 [N][C][ldy] -> p bf16 [N][C][ldp] (rows zero-filled from L/2 to ldp: the next conv's
//                                  forward and weight gradient read whole 16-byte chunks), statistics combine folded in
//                                  exactly as ecg_bn_stats_relu_pool_fwd does (same arithmetic, same bits);
//   ecg_bn_relu_pool_bwd_h         reduction pass + dx pass (the combine of the reduction partials folded into the second):
//                                  y bf16, dp bf16 [N][C][ldp] (or fp32 dg [N][C] of the fused global average pool) ->
//                                  dY bf16 [N][C][ldy] with rows zero-filled to ldy (a multiple of 128: what the weight
//                                  gradient's LDS-DMA wants), dgamma, dbeta.
// Per element the formulas are those of bn_relu_pool.hip (bn_apply1, first element wins a tie, ReLU on the pooled value):
// on the same bf16 inputs the outputs are bit-identical to the n16 producers' bf16 outputs; the partial sums of the
// reduction associate differently (both deterministic).
// Replaces autograd of ConvBlock.net[1..3] (reference src/models/ecg_cnn.py:14-16) in the opt-in bf16 mode.
#include "common.h"

namespace ecg {

typedef unsigned short u16h;
typedef unsigned u32x4h __attribute__((ext_vector_type(4)));
typedef unsigned u32x2h __attribute__((ext_vector_type(2)));
constexpr int kBlockH = 256;
constexpr int kStreamBlocksH = 2048;      // one round of resident workgroups (bn_relu_pool.hip: kStreamBlocks)

__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pack2h(float lo, float hi) {
    const u16h a = __builtin_bit_cast(u16h, (__bf16)lo), b = __builtin_bit_cast(u16h, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

// the statistics combine of bn_relu_pool.hip (double, fixed order), restated here because __shared__ helpers do not cross
// translation units; `writer` stores mean / invstd and updates the running statistics and the counter
struct BnFinH {
    const float *partials; int P; double count; float *mean, *invstd, *running_mean, *running_var;
    long long *nbt; float momentum, eps;
};
__device__ __forceinline__ void bn_finalize_block_h(const BnFinH &f, int c, bool writer, float &mu_out, float &is_out) {
    __shared__ double red[4][2];
    __shared__ float res[2];
    const int tl = threadIdx.x;
    const float2 *pc = reinterpret_cast<const float2 *>(f.partials) + (size_t)c * f.P;
    double a = 0.0, q = 0.0;
    for (int p = tl; p < f.P; p += 256) {
        const float2 v = pc[p];
        a += (double)v.x;
        q += (double)v.y;
    }
    a = wave_sum(a); q = wave_sum(q);
    if ((tl & 63) == 0) { red[tl >> 6][0] = a; red[tl >> 6][1] = q; }
    __syncthreads();
    if (tl == 0) {
        a = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        q = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
        const double mu = a / f.count;
        double var = q / f.count - mu * mu;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)f.eps));
        res[0] = mf; res[1] = isf;
        if (writer) {
            f.mean[c] = mf;
            f.invstd[c] = isf;
            if (f.running_mean) {
                const double unb = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
                f.running_mean[c] = (float)((1.0 - f.momentum) * f.running_mean[c] + f.momentum * mu);
                f.running_var[c] = (float)((1.0 - f.momentum) * f.running_var[c] + f.momentum * unb);
            }
            if (f.nbt && c == 0) *f.nbt += 1;
        }
    }
    __syncthreads();
    mu_out = res[0]; is_out = res[1];
}

__device__ __forceinline__ bool pool_route_h(float y0, float y1, float mu, float sc, float be, int &am) {
    const float a0 = bn_apply1(y0, mu, sc, be), a1 = bn_apply1(y1, mu, sc, be);
    am = a1 > a0 ? 1 : 0;          // first element wins a tie (max_pool1d keeps the first index)
    return (am ? a1 : a0) > 0.f;   // ReLU backward: output > 0
}

// ---------------------------------------------------------------------------------------
// forward: grid = (C, S2); a workgroup owns channel c and the samples of split s2 and walks (row, chunk of 8 pooled
// outputs) flat, U chunks in flight per thread: 32 bytes of y in, 16 bytes of p out per chunk
// ---------------------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(kBlockH) void bn_relu_pool_fwd_h_kernel(
    const u16h *__restrict__ y, const float *__restrict__ gamma, const float *__restrict__ beta, u16h *__restrict__ p,
    int N, int C, int Lp, int ldy, int ldp, int S2, BnFinH fin) {
    const int c = blockIdx.x, s2 = blockIdx.y, tl = threadIdx.x;
    float mu, is;
    bn_finalize_block_h(fin, c, s2 == 0, mu, is);
    const float sc = is * gamma[c], be = beta[c];
    const int n0 = (int)((long long)N * s2 / S2), n1 = (int)((long long)N * (s2 + 1) / S2);
    const int CH = ldp >> 3, total = (n1 - n0) * CH;
    for (int base = tl; base < total; base += U * kBlockH) {
        u32x4h lo[U], hi[U];
        size_t out[U];
        int q8[U];
        bool live[U], vlo[U], vhi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * kBlockH;
            live[u] = idx < total;
            const int ic = live[u] ? idx : tl;
            const int nl = ic / CH, q = ic - nl * CH;
            const size_t row = (size_t)(n0 + nl) * C + c;
            const int e0 = 16 * q;                          // y elements e0 .. e0 + 15 of the row (ldy is a multiple of 8)
            vlo[u] = e0 + 8 <= ldy; vhi[u] = e0 + 16 <= ldy;
            const u16h *yr = y + row * ldy;
            lo[u] = *reinterpret_cast<const u32x4h *>(yr + min(e0, ldy - 8));
            hi[u] = *reinterpret_cast<const u32x4h *>(yr + min(e0 + 8, ldy - 8));
            out[u] = row * ldp + 8 * q;
            q8[u] = 8 * q;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            float m[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned v = i < 4 ? lo[u][i] : hi[u][i - 4];
                const float a0 = bn_apply1(bf_lo(v), mu, sc, be), a1 = bn_apply1(bf_hi(v), mu, sc, be);
                float r = a1 > a0 ? a1 : a0;
                r = r > 0.f ? r : 0.f;
                // pooled position q8 + i exists iff < Lp (then both its inputs are inside the row); the rest is the zero fill
                m[i] = (q8[u] + i < Lp && (i < 4 ? vlo[u] : vhi[u])) ? r : 0.f;
            }
            u32x4h o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = pack2h(m[2 * i], m[2 * i + 1]);
            *reinterpret_cast<u32x4h *>(p + out[u]) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------
// backward, pass 1: partials[c][s][2] = (sum da, sum da*xhat) over the samples of split s; thread <-> chunk of 8 y
// elements (4 pooling pairs): 16 bytes of y + 8 bytes of dp
// ---------------------------------------------------------------------------------------
// DK: 0 = dp bf16 [N][C][ldp]; 1 = fp32 dg [N][C] of the fused global average pool (dp = dg * bcast everywhere);
//     2 = dp fp32 [N][C][ldp] (a consumer that hands back an fp32 gradient: frozen statistics downstream, unfused leaves)
template <int DK, int U>
__global__ __launch_bounds__(kBlockH) void bn_bwd_reduce_h_kernel(
    const u16h *__restrict__ y, const void *__restrict__ g, const float *__restrict__ gamma,
    const float *__restrict__ beta, const float *__restrict__ mean, const float *__restrict__ invstd,
    float *__restrict__ partials, int N, int C, int L, int S, float bcast, int ldyy, int ldp) {
    __shared__ float red[4][2];
    const int c = blockIdx.x, s = blockIdx.y, tl = threadIdx.x;
    const int n0 = (int)((long long)N * s / S), n1 = (int)((long long)N * (s + 1) / S);
    const float mu = mean[c], is = invstd[c], sc = is * gamma[c], be = beta[c];
    const int Lp = L >> 1, CH = (Lp + 3) >> 2, total = (n1 - n0) * CH;     // chunks that hold at least one pooling pair
    float a = 0.f, q = 0.f;
    for (int base = tl; base < total; base += U * kBlockH) {
        u32x4h yv[U];
        float d[U][4];
        int j0[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * kBlockH;
            live[u] = idx < total;
            const int ic = live[u] ? idx : tl;
            const int nl = ic / CH, qc = ic - nl * CH;
            const size_t row = (size_t)(n0 + nl) * C + c;
            yv[u] = *reinterpret_cast<const u32x4h *>(y + row * ldyy + 8 * qc);            // (8 qc < L <= ldyy, ldyy % 8 == 0)
            j0[u] = 4 * qc;
            if (DK == 1) {
                const float gv = __fmul_rn(static_cast<const float *>(g)[row], bcast);
#pragma unroll
                for (int i = 0; i < 4; ++i) d[u][i] = gv;
            } else if (DK == 2) {
                const float *gr = static_cast<const float *>(g) + row * ldp;
#pragma unroll
                for (int i = 0; i < 4; ++i) d[u][i] = gr[min(4 * qc + i, ldp - 1)];       // (a pair past Lp is never used)
            } else {
                const u32x2h dv = *reinterpret_cast<const u32x2h *>(static_cast<const u16h *>(g) + row * ldp + 4 * qc);
                d[u][0] = bf_lo(dv[0]); d[u][1] = bf_hi(dv[0]); d[u][2] = bf_lo(dv[1]); d[u][3] = bf_hi(dv[1]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (j0[u] + i >= Lp) continue;             // (an odd tail sample never reaches the pool)
                const float y0 = bf_lo(yv[u][i]), y1 = bf_hi(yv[u][i]);
                int am;
                if (pool_route_h(y0, y1, mu, sc, be, am)) {
                    a += d[u][i];
                    q = __fmaf_rn(d[u][i], ((am ? y1 : y0) - mu) * is, q);
                }
            }
        }
    }
    a = wave_sum(a); q = wave_sum(q);
    if ((tl & 63) == 0) { red[tl >> 6][0] = a; red[tl >> 6][1] = q; }
    __syncthreads();
    if (tl < 2) partials[((size_t)c * S + s) * 2 + tl] = ((red[0][tl] + red[1][tl]) + red[2][tl]) + red[3][tl];
}

// ---------------------------------------------------------------------------------------
// backward, pass 2: dY = gamma*invstd * (da - mean(da) - xhat * mean(da*xhat)), rows zero-filled to ldy; the combine of the
// S reduction partials (double, fixed order) is folded in; workgroup (c, 0) also stores dgamma / dbeta.
// thread <-> chunk of 8 outputs: 16 bytes of y + 8 of dp in, 16 out
// ---------------------------------------------------------------------------------------
template <int DK, int U>
__global__ __launch_bounds__(kBlockH) void bn_bwd_dx_h_kernel(
    const u16h *__restrict__ y, const void *__restrict__ g, const float *__restrict__ gamma,
    const float *__restrict__ beta, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ partials, int S, double M, float *__restrict__ dgamma, float *__restrict__ dbeta,
    u16h *__restrict__ dy, int ldy, int N, int C, int L, int S2, float bcast, int train, int ldyy, int ldp) {
    __shared__ double red[4][2];
    __shared__ float kk[2];
    const int c = blockIdx.x, s2 = blockIdx.y, tl = threadIdx.x;
    {
        double a = 0.0, q = 0.0;
        for (int pp = tl; pp < S; pp += kBlockH) {
            a += (double)partials[((size_t)c * S + pp) * 2];
            q += (double)partials[((size_t)c * S + pp) * 2 + 1];
        }
        a = wave_sum(a); q = wave_sum(q);
        if ((tl & 63) == 0) { red[tl >> 6][0] = a; red[tl >> 6][1] = q; }
        __syncthreads();
        if (tl == 0) {
            a = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
            q = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
            if (s2 == 0) {
                if (dbeta) dbeta[c] = (float)a;
                if (dgamma) dgamma[c] = (float)q;
            }
            kk[0] = train ? (float)(a / M) : 0.f;
            kk[1] = train ? (float)(q / M) : 0.f;
        }
        __syncthreads();
    }
    const float k1 = kk[0], k2 = kk[1];
    const float mu = mean[c], is = invstd[c], ga = gamma[c], sc = is * ga, gi = ga * is, be = beta[c];
    const int n0 = (int)((long long)N * s2 / S2), n1 = (int)((long long)N * (s2 + 1) / S2);
    const int Lp = L >> 1, CH = ldy >> 3, total = (n1 - n0) * CH;
    const int lastq = (L - 1) >> 3;                          // last chunk that holds a sample of the row
    for (int base = tl; base < total; base += U * kBlockH) {
        u32x4h yv[U];
        float d[U][4];
        size_t out[U];
        int t0[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * kBlockH;
            live[u] = idx < total;
            const int ic = live[u] ? idx : tl;
            const int nl = ic / CH, qc = ic - nl * CH;
            const size_t row = (size_t)(n0 + nl) * C + c;
            const int ql = min(qc, lastq);                  // chunks past the row are pure zero fill: their loads are clamped
            yv[u] = *reinterpret_cast<const u32x4h *>(y + row * ldyy + 8 * ql);
            out[u] = row * ldy + 8 * qc;
            t0[u] = 8 * qc;
            if (DK == 1) {
                const float gv = __fmul_rn(static_cast<const float *>(g)[row], bcast);
#pragma unroll
                for (int i = 0; i < 4; ++i) d[u][i] = gv;
            } else if (DK == 2) {
                const float *gr = static_cast<const float *>(g) + row * ldp;
#pragma unroll
                for (int i = 0; i < 4; ++i) d[u][i] = gr[min(4 * ql + i, ldp - 1)];
            } else {
                const int jq = min(4 * ql, max(ldp - 4, 0));
                const u32x2h dv = *reinterpret_cast<const u32x2h *>(static_cast<const u16h *>(g) + row * ldp + jq);
                d[u][0] = bf_lo(dv[0]); d[u][1] = bf_hi(dv[0]); d[u][2] = bf_lo(dv[1]); d[u][3] = bf_hi(dv[1]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            u32x4h o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = t0[u] + 2 * i;
                const bool in0 = t < L, has1 = t + 1 < L;
                const float y0 = bf_lo(yv[u][i]), y1 = bf_hi(yv[u][i]);
                float da0 = 0.f, da1 = 0.f;
                if (has1 && (t >> 1) < Lp) {                // an odd tail sample never reaches the pool: da = 0
                    int am;
                    if (pool_route_h(y0, y1, mu, sc, be, am)) {
                        if (am) da1 = d[u][i]; else da0 = d[u][i];
                    }
                }
                const float v0 = in0 ? gi * (da0 - k1 - (y0 - mu) * is * k2) : 0.f;
                const float v1 = has1 ? gi * (da1 - k1 - (y1 - mu) * is * k2) : 0.f;
                o[i] = pack2h(v0, v1);
            }
            *reinterpret_cast<u32x4h *>(dy + out[u]) = o;
        }
    }
}

}  // namespace ecg

using namespace ecg;

// Chunks in flight per thread.  A workgroup walks its `items` chunks in trips of U x 256; with one round of workgroups running
// in lock-step a half-empty last trip is idle bandwidth (12x5000, 2048 workgroups: 1252-2560 chunks per workgroup — U = 4
// left 17-39 % of the slots of the last trip empty).  Take the U in 3..6 that wastes the fewest slots (ties: the larger).
static int pick_u(long long items) {
    int best = 4;
    long long best_slots = -1;
    for (int u = 3; u <= 6; ++u) {
        const long long trips = (items + 256LL * u - 1) / (256LL * u), slots = trips * u;
        if (best_slots < 0 || slots <= best_slots) { best = u; best_slots = slots; }
    }
    return best;
}
#define ECG_PICK_U(U_, ...) \
    do { switch (U_) { case 3: { constexpr int UU = 3; __VA_ARGS__; } break; case 5: { constexpr int UU = 5; __VA_ARGS__; } break; \
                       case 6: { constexpr int UU = 6; __VA_ARGS__; } break; default: { constexpr int UU = 4; __VA_ARGS__; } } } while (0)

static int splits_h(int N, int C) {
    int s = cdiv(2048, C);
    if (s > N) s = N;
    return s < 1 ? 1 : s;
}

ECG_API int ecg_bn_stats_relu_pool_fwd_h(const float *stat_partials, int P, long long count, float *running_mean,
                                         float *running_var, long long *num_batches_tracked, float momentum, float eps,
                                         const void *y_bf16, int ldy, const float *gamma, const float *beta, float *mean,
                                         float *invstd, void *p_bf16, int ldp, int N, int C, int L,
                                         ecg_stream_t stream) {
    ECG_REQUIRE(N > 0 && C > 0 && L >= 2 && C <= 65535, "bn_stats_relu_pool_fwd_h: N=%d C=%d L=%d", N, C, L);
    ECG_REQUIRE(stat_partials && mean && invstd && y_bf16 && gamma && beta && p_bf16, "bn_stats_relu_pool_fwd_h: null pointer");
    ECG_REQUIRE(P > 0 && count > 0, "bn_stats_relu_pool_fwd_h: P=%d count=%lld", P, count);
    ECG_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
                "bn_stats_relu_pool_fwd_h: running_mean/var must both be given or both NULL");
    ECG_REQUIRE(ldy % 8 == 0 && ldy >= L && ldp % 8 == 0 && ldp >= L / 2,
                "bn_stats_relu_pool_fwd_h: row strides must be multiples of 8 (ldy=%d >= L, ldp=%d >= L/2)", ldy, ldp);
    ECG_REQUIRE(((reinterpret_cast<uintptr_t>(y_bf16) | reinterpret_cast<uintptr_t>(p_bf16)) & 15) == 0,
                "bn_stats_relu_pool_fwd_h: tensors must be 16-byte aligned");
    const BnFinH f{stat_partials, P, (double)count, mean, invstd, running_mean, running_var, num_batches_tracked,
                   momentum, eps};
    int S2 = cdiv(kStreamBlocksH, C);
    if (S2 > N) S2 = N;
    const int u = pick_u((long long)cdiv(N, S2) * (ldp / 8));
    ECG_PICK_U(u, hipLaunchKernelGGL(bn_relu_pool_fwd_h_kernel<UU>, dim3(C, S2), dim3(kBlockH), 0, as_stream(stream),
                                     static_cast<const u16h *>(y_bf16), gamma, beta, static_cast<u16h *>(p_bf16), N, C, L / 2,
                                     ldy, ldp, S2, f));
    return check_launch("bn_relu_pool_fwd_h_kernel");
}

ECG_API int ecg_bn_relu_pool_bwd_h(const void *y_bf16, int ldyy, const void *dp, int dp_kind, int ldp, const float *gamma,
                                   const float *beta, const float *mean, const float *invstd, void *dy_bf16, int ldy,
                                   float *dgamma, float *dbeta, float *ws, int N, int C, int L, int train,
                                   ecg_stream_t stream) {
    ECG_REQUIRE(N > 0 && C > 0 && L >= 2 && C <= 65535, "bn_relu_pool_bwd_h: N=%d C=%d L=%d", N, C, L);
    ECG_REQUIRE(y_bf16 && dp && gamma && beta && mean && invstd && dy_bf16 && ws, "bn_relu_pool_bwd_h: null pointer");
    ECG_REQUIRE(dp_kind >= 0 && dp_kind <= 2, "bn_relu_pool_bwd_h: dp_kind %d (0 bf16 rows, 1 fp32 dg, 2 fp32 rows)", dp_kind);
    ECG_REQUIRE(ldyy % 8 == 0 && ldyy >= L && ldy % 8 == 0 && ldy >= L,
                "bn_relu_pool_bwd_h: row strides must be multiples of 8 and >= L (y %d, dY %d)", ldyy, ldy);
    ECG_REQUIRE(dp_kind != 0 || (ldp % 4 == 0 && ldp >= (L / 2 + 3) / 4 * 4), "bn_relu_pool_bwd_h: bf16 dp row stride %d", ldp);
    ECG_REQUIRE(dp_kind != 2 || ldp >= L / 2, "bn_relu_pool_bwd_h: fp32 dp row stride %d", ldp);
    ECG_REQUIRE(((reinterpret_cast<uintptr_t>(y_bf16) | reinterpret_cast<uintptr_t>(dy_bf16)) & 15) == 0 &&
                (dp_kind != 0 || (reinterpret_cast<uintptr_t>(dp) & 7) == 0), "bn_relu_pool_bwd_h: tensors must be 16-byte aligned");
    ECG_REQUIRE(ecg_bn_relu_pool_bwd_ws_floats(N, C, L) >= (size_t)C * splits_h(N, C) * 2, "bn_relu_pool_bwd_h: workspace");
    hipStream_t st = as_stream(stream);
    const int S = splits_h(N, C);
    const float bcast = dp_kind == 1 ? 1.0f / (float)(L / 2) : 0.f;
    const u16h *y = static_cast<const u16h *>(y_bf16);
    const int ur = pick_u((long long)cdiv(N, S) * ((L / 2 + 3) / 4));
#define ECG_RED(DK) ECG_PICK_U(ur, hipLaunchKernelGGL((bn_bwd_reduce_h_kernel<DK, UU>), dim3(C, S), dim3(kBlockH), 0, st, y, dp, gamma, \
                                                      beta, mean, invstd, ws, N, C, L, S, bcast, ldyy, ldp))
    if (dp_kind == 1) ECG_RED(1); else if (dp_kind == 2) ECG_RED(2); else ECG_RED(0);
#undef ECG_RED
    int rc = check_launch("bn_bwd_reduce_h_kernel");
    if (rc) return rc;
    int S2 = cdiv(kStreamBlocksH, C);
    if (S2 > N) S2 = N;
    const int ud = pick_u((long long)cdiv(N, S2) * (ldy / 8));
#define ECG_DX(DK) ECG_PICK_U(ud, hipLaunchKernelGGL((bn_bwd_dx_h_kernel<DK, UU>), dim3(C, S2), dim3(kBlockH), 0, st, y, dp, gamma, beta, \
                                                     mean, invstd, ws, S, (double)N * L, dgamma, dbeta, static_cast<u16h *>(dy_bf16), \
                                                     ldy, N, C, L, S2, bcast, train, ldyy, ldp))
    if (dp_kind == 1) ECG_DX(1); else if (dp_kind == 2) ECG_DX(2); else ECG_DX(0);
#undef ECG_DX
    return check_launch("bn_bwd_dx_h_kernel");
}
