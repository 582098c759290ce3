// bn_relu_pool.hip — BatchNorm1d statistics, the fused BatchNorm -> ReLU -> MaxPool1d(2) tail of
// a ConvBlock and its backward, plus the unfused leaves.  All HBM-streaming kernels: one read of
// the conv output, one write of the pooled activation (forward); backward recomputes the pool
// arg-max and ReLU mask from the saved conv output instead of storing indices.
//
// Replaces ATen native_batch_norm(_backward), threshold(_backward), max_pool2d_with_indices
// (_backward) behind ConvBlock.net[1..3] (reference src/models/ecg_cnn.py:14-16).
#include "common.h"
#include <atomic>
#include <cstdlib>

namespace ecg {

constexpr int kBlock = 256;

// the pooling pair (r[0], r[1]) as ONE 8-byte load when the pair is 8-byte aligned (even row length: every row of
// the 12x1000 / 12x5000 models except the last block's), two 4-byte loads otherwise (AL8 is chosen per launch by the host)
template <bool AL8>
__device__ __forceinline__ void ld_pair(const float *r, float &a, float &b) {
    if (AL8) {
        const float2 v = *reinterpret_cast<const float2 *>(r);
        a = v.x; b = v.y;
    } else {
        a = r[0]; b = r[1];
    }
}

// The pooling pair (t, t+1), t even, of row `row` of y: fp32 rows of stride ld, or (YH) bf16 rows of EVEN stride ld
// (bf16 activation storage: one aligned dword holds the pair).
template <bool AL8, bool YH>
__device__ __forceinline__ void ld_pair_y(const float *y, size_t row, int ld, int t, float &a, float &b) {
    if (YH) {
        const unsigned v = *reinterpret_cast<const unsigned *>(reinterpret_cast<const unsigned short *>(y) + row * ld + t);
        a = __uint_as_float(v << 16);
        b = __uint_as_float(v & 0xFFFF0000u);
    } else ld_pair<AL8>(y + row * ld + t, a, b);
}

// ---------------------------------------------------------------------------------------
// statistics
// ---------------------------------------------------------------------------------------
// Workgroups of the streaming passes: 2048 = 256 CUs x 8 resident 256-thread workgroups, i.e. exactly one round (every
// workgroup pays the folded statistics combine once, none waits for a slot).  Measured per train step at B=256 12x1000:
// 1024: +9 us, 1536: -3, 2048: -10, 2560: +6, 3072: +3, 4096: 0 (round 1's value), 8192: +42.
constexpr int kStreamBlocks = 2048;
// workgroups of the reduce passes per channel.  fp32 operands: 1024 in all (2048 measured the same at 12x1000, 512 slower).
// bf16 operands (the bf16-storage backward, `wide`): 2048 — that pass is bound by the latency of its dependent load rounds
// (6 bytes per position: the number of loads in flight per thread, the per-position instruction count and the walk order
// were all varied without effect), and twice the workgroups took 4-11 us off every backward call of 12x5000 (config-5 step
// 1.449 -> 1.415 ms); 4096 adds nothing.
static int stat_splits(int N, int C, bool wide = false) {
    int s = cdiv(wide ? 2048 : 1024, C);
    if (s > N) s = N;
    if (s < 1) s = 1;
    return s;
}

// partials[c][s][2] = (sum y, sum y^2) over n in split s.  grid = (C, S)
__global__ __launch_bounds__(kBlock) void bn_stat_partials_kernel(
    const float *__restrict__ y, float *__restrict__ partials, int N, int C, int L, int S) {
    __shared__ float red[4][2];
    const int c = blockIdx.x, s = blockIdx.y, tl = threadIdx.x;
    const int n0 = (int)((long long)N * s / S), n1 = (int)((long long)N * (s + 1) / S);
    float a = 0.f, q = 0.f;
    const int total = (n1 - n0) * L;
    for (int idx = tl; idx < total; idx += kBlock) {
        const int nl = idx / L, t = idx - nl * L;
        const float v = y[((size_t)(n0 + nl) * C + c) * L + t];
        a += v; q = __fmaf_rn(v, v, q);
    }
    a = wave_sum(a); q = wave_sum(q);
    if ((tl & 63) == 0) { red[tl >> 6][0] = a; red[tl >> 6][1] = q; }
    __syncthreads();
    if (tl < 2)
        partials[((size_t)c * S + s) * 2 + tl] = red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl];
}

// Double-precision, fixed-order combine of the P statistics partials of channel c by one 256-thread workgroup
// (thread t sums partials t, t+256, ...; lanes by butterfly; waves 0..3 in order) -> batch mean / invstd, returned
// to every thread.  `writer` (one workgroup per channel) also stores them and updates the running statistics
// (momentum, unbiased variance) and, for channel 0, the batch counter.  Used by bn_finalize_kernel (its own launch)
// and, FOLDED IN, by the BN+ReLU+pool forward kernels: every workgroup re-derives the numbers of its channel from
// L2 instead of waiting for a 4.7 us launch per layer; same arithmetic, same bits.
struct BnFin {
    const float *partials; int P; double count; float *mean, *invstd, *running_mean, *running_var;
    long long *nbt; float momentum, eps;
};
__device__ __forceinline__ void bn_finalize_block(const BnFin &f, int c, bool writer, float &mu_out, float &is_out) {
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

__global__ __launch_bounds__(256) void bn_finalize_kernel(BnFin f) {
    float mu, is;
    bn_finalize_block(f, blockIdx.x, true, mu, is);
}

__global__ void bn_invstd_kernel(const float *__restrict__ var, float *__restrict__ invstd, int C,
                                 float eps) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) invstd[c] = (float)(1.0 / sqrt((double)var[c] + (double)eps));
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
// grid = (C, S2): a block owns channel c and the samples of split s2 and walks their pooled positions flat, four
// iterations in flight (loads unconditional, clamped); a wave reads 512 contiguous bytes of y and writes 256 of p.
template <bool FIN, bool AL8>
__global__ __launch_bounds__(kBlock) void bn_relu_pool_fwd_kernel(
    const float *__restrict__ y, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ p,
    int N, int C, int L, int Lp, int S2, BnFin fin) {
    const int c = blockIdx.x, s2 = blockIdx.y, tl = threadIdx.x;
    float mu, is;
    if (FIN) bn_finalize_block(fin, c, s2 == 0, mu, is);      // statistics combine folded in (no finalize launch)
    else { mu = mean[c]; is = invstd[c]; }
    const float sc = is * gamma[c], be = beta[c];
    const int n0 = (int)((long long)N * s2 / S2), n1 = (int)((long long)N * (s2 + 1) / S2);
    const int total = (n1 - n0) * Lp;
    constexpr int U = 4;
    for (int base = tl; base < total; base += U * kBlock) {
        float y0[U], y1[U];
        size_t out[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * kBlock;
            live[u] = idx < total;
            const int ic = live[u] ? idx : tl;
            const int nl = ic / Lp, j = ic - nl * Lp;
            const size_t row = (size_t)(n0 + nl) * C + c;
            ld_pair<AL8>(y + row * L + 2 * j, y0[u], y1[u]);
            out[u] = row * Lp + j;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            const float a0 = bn_apply1(y0[u], mu, sc, be), a1 = bn_apply1(y1[u], mu, sc, be);
            const float m = a1 > a0 ? a1 : a0;
            p[out[u]] = m > 0.f ? m : 0.f;
        }
    }
}

// Last block of the backbone: BatchNorm -> ReLU -> MaxPool(2) -> AdaptiveAvgPool1d(1) without
// materialising the pooled tensor.  grid = (C, S2): a workgroup owns channel c and the samples of split s2, one wave
// per (n, c) row at a time; g[row] = mean_j pooled[row][j].
template <bool FIN, bool YH = false>
__global__ __launch_bounds__(kBlock) void bn_relu_pool_gap_fwd_kernel(
    const float *__restrict__ y, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ g,
    int N, int C, int L, int Lp, int S2, BnFin fin, int ldy) {
    const int c = blockIdx.x, s2 = blockIdx.y;
    float mu, is;
    if (FIN) bn_finalize_block(fin, c, s2 == 0, mu, is);
    else { mu = mean[c]; is = invstd[c]; }
    const float sc = is * gamma[c], be = beta[c];
    const int n0 = (int)((long long)N * s2 / S2), n1 = (int)((long long)N * (s2 + 1) / S2);
    const int lane = threadIdx.x & 63;
    // A wave owns every fourth sample of the split and takes them FOUR ROWS AT A TIME: the pair loads of four rows are in
    // flight together and their four lane reductions interleave.  (One row per trip left this pass latency-bound: 32.8 MB
    // in 21 us on the 62-pair rows of 12x1000, 82 MB in 25 us on the 312-pair rows of 12x5000.)  Per row the additions
    // and their order are unchanged.
#ifndef ECG_GAP_RB
#define ECG_GAP_RB 4
#endif
    constexpr int RB = ECG_GAP_RB;
    for (int nb = n0 + (threadIdx.x >> 6); nb < n1; nb += 4 * RB) {
        float a[RB];
        size_t rows[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            a[r] = 0.f;
            rows[r] = (size_t)min(nb + 4 * r, n1 - 1) * C + c;          // clamped: a tail row is recomputed, stored once
        }
        for (int j = lane; j < Lp; j += 64) {
            float r0[RB], r1[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) ld_pair_y<false, YH>(y, rows[r], ldy, 2 * j, r0[r], r1[r]);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float a0 = bn_apply1(r0[r], mu, sc, be), a1 = bn_apply1(r1[r], mu, sc, be);
                const float m = a1 > a0 ? a1 : a0;
                a[r] += m > 0.f ? m : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) a[r] = wave_sum(a[r]);
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (lane == 0 && nb + 4 * r < n1) g[rows[r]] = a[r] / (float)Lp;
    }
}

__global__ __launch_bounds__(kBlock) void bn_apply_fwd_kernel(
    const float *__restrict__ y, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ out,
    int C, int L, size_t total) {
    size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= total) return;
    int c = (int)((idx / L) % C);
    out[idx] = bn_apply1(y[idx], mean[c], invstd[c] * gamma[c], beta[c]);
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
// da for the pair (2j, 2j+1): returns the arg-max slot am (0/1) and whether the gradient passes.
__device__ __forceinline__ bool pool_route(float y0, float y1, float mu, float sc, float be, int &am) {
    float a0 = bn_apply1(y0, mu, sc, be), a1 = bn_apply1(y1, mu, sc, be);
    am = a1 > a0 ? 1 : 0;          // first element wins a tie (max_pool1d keeps the first index)
    return (am ? a1 : a0) > 0.f;   // ReLU backward: output > 0
}

// partials[c][s][2] = (sum da, sum da*xhat).  FUSED: da routed from dp through pool+ReLU;
// otherwise da = dout (plain BatchNorm backward).  grid = (C, S).
// bcast != 0 (FUSED only): dp[row][j] = g[row] * bcast for every j — the gradient of a global
// average pool that was fused behind the max-pool (g = dG [N*C], bcast = 1/Lp).
template <bool FUSED, bool AL8>
__global__ __launch_bounds__(kBlock) void bn_bwd_reduce_kernel(
    const float *__restrict__ y, const float *__restrict__ g, const float *__restrict__ gamma,
    const float *__restrict__ beta, const float *__restrict__ mean,
    const float *__restrict__ invstd, float *__restrict__ partials, int N, int C, int L, int S,
    float bcast) {
    __shared__ float red[4][2];
    const int c = blockIdx.x, s = blockIdx.y, tl = threadIdx.x;
    const int n0 = (int)((long long)N * s / S), n1 = (int)((long long)N * (s + 1) / S);
    const float mu = mean[c], is = invstd[c], sc = is * gamma[c], be = FUSED ? beta[c] : 0.f;
    const int Lp = L >> 1;
    float a = 0.f, q = 0.f;
    // flat walk over (n, position) of this block's n-range: all 256 lanes stay busy even when a
    // row is shorter than the workgroup (Lp = 62 for the last block)
    const int per = FUSED ? Lp : L;
    const int total = (n1 - n0) * per;
    // Loads are unconditional (a load under the routing branch gets a vmcnt(0) of its own) and four
    // iterations are in flight at once; the per-thread accumulation order is unchanged.
    constexpr int U = 4;
    for (int base = tl; base < total; base += U * kBlock) {
        float y0[U], y1[U], d[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * kBlock;
            live[u] = idx < total;
            const int ic = live[u] ? idx : tl;                     // clamp to this thread's first (valid) element
            const int nl = ic / per, j = ic - nl * per;
            const size_t row = (size_t)(n0 + nl) * C + c;
            if (FUSED) {
                ld_pair<AL8>(y + row * L + 2 * j, y0[u], y1[u]);
                d[u] = bcast != 0.f ? g[row] * bcast : g[row * Lp + j];
            } else {
                y0[u] = y[row * L + j]; y1[u] = 0.f;
                d[u] = g[row * L + j];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            if (FUSED) {
                int am;
                if (pool_route(y0[u], y1[u], mu, sc, be, am)) {
                    a += d[u];
                    q = __fmaf_rn(d[u], ((am ? y1[u] : y0[u]) - mu) * is, q);
                }
            } else {
                a += d[u];
                q = __fmaf_rn(d[u], (y0[u] - mu) * is, q);
            }
        }
    }
    a = wave_sum(a); q = wave_sum(q);
    if ((tl & 63) == 0) { red[tl >> 6][0] = a; red[tl >> 6][1] = q; }
    __syncthreads();
    if (tl < 2)
        partials[((size_t)c * S + s) * 2 + tl] = red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl];
}

// dy[t] = gamma*invstd * (da[t] - k1 - xhat[t]*k2), (k1, k2) = (dbeta, dgamma) / M in train mode, (0, 0) in eval mode.
// grid = (C, S2): a block owns channel c and the samples of split s2 and walks their (sample, output pair) positions
// flat, four iterations in flight (all loads unconditional, as in the reduce kernel above).
// The combine of the S reduce partials is FOLDED in: every block re-derives (dbeta, dgamma) of its channel in double,
// in a fixed order (thread t sums partials t, t+256, ...; lanes by butterfly; waves 0..3 in order) — S <= 1024/C + 1
// numbers out of L2 — instead of a separate 4.9 us launch per layer; block s2 == 0 writes dbeta / dgamma.
// dy rows have stride ldy >= L; the pad [L, ldy) is written as zeros (Lh = ceil(ldy/2) pairs per row): the
// weight-gradient kernel streams such rows by LDS-DMA and needs the zeros.
template <bool FUSED, bool AL8>
__global__ __launch_bounds__(kBlock) void bn_bwd_dx_kernel(
    const float *__restrict__ y, const float *__restrict__ g, const float *__restrict__ gamma,
    const float *__restrict__ beta, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ partials, int S, double M,
    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dy,
    int N, int C, int L, int ldy, int Lh, int S2, float bcast, int train) {
    __shared__ double red[4][2];
    __shared__ float kk[2];
    const int c = blockIdx.x, s2 = blockIdx.y, tl = threadIdx.x;
    {
        double a = 0.0, q = 0.0;
        for (int p = tl; p < S; p += kBlock) {
            a += (double)partials[((size_t)c * S + p) * 2];
            q += (double)partials[((size_t)c * S + p) * 2 + 1];
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
    const float mu = mean[c], is = invstd[c], ga = gamma[c], sc = is * ga, gi = ga * is;
    const float be = FUSED ? beta[c] : 0.f;
    const int n0 = (int)((long long)N * s2 / S2), n1 = (int)((long long)N * (s2 + 1) / S2);
    const int Lp = L >> 1;
    const int total = (n1 - n0) * Lh;
    constexpr int U = 4;
    for (int base = tl; base < total; base += U * kBlock) {
        float y0[U], y1[U], d0[U], d1[U];
        int t0[U];
        size_t rowv[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * kBlock;
            live[u] = idx < total;
            const int ic = live[u] ? idx : tl;                     // clamp to this thread's first (valid) pair
            const int nl = ic / Lh, j = ic - nl * Lh;
            const size_t row = (size_t)(n0 + nl) * C + c;
            rowv[u] = row; t0[u] = 2 * j;
            const float *r = y + row * L;
            const int ta = min(2 * j, L - 1), tb = min(2 * j + 1, L - 1);
            if (AL8) ld_pair<true>(r + min(2 * j, L - 2), y0[u], y1[u]);      // (pad pairs read the row's last pair: unused)
            else { y0[u] = r[ta]; y1[u] = r[tb]; }
            if (FUSED) {
                d0[u] = bcast != 0.f ? g[row] * bcast : (Lp > 0 ? g[row * (size_t)Lp + min(j, Lp - 1)] : 0.f);
                d1[u] = 0.f;
            } else {
                d0[u] = g[row * L + ta]; d1[u] = g[row * L + tb];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            float *d = dy + rowv[u] * (size_t)ldy;
            const int t = t0[u];
            if (t >= L) {                  // pad pair
                d[t] = 0.f;
                if (t + 1 < ldy) d[t + 1] = 0.f;
                continue;
            }
            const bool has1 = t + 1 < L;
            float da0 = 0.f, da1 = 0.f;
            if (FUSED) {
                if (has1) {   // an odd tail sample never reaches the pool: da = 0
                    int am;
                    if (pool_route(y0[u], y1[u], mu, sc, be, am)) {
                        if (am) da1 = d0[u]; else da0 = d0[u];
                    }
                }
            } else {
                da0 = d0[u];
                if (has1) da1 = d1[u];
            }
            d[t] = gi * (da0 - k1 - (y0[u] - mu) * is * k2);
            if (has1) d[t + 1] = gi * (da1 - k1 - (y1[u] - mu) * is * k2);
            else if (t + 1 < ldy) d[t + 1] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same backward in ONE launch with its operands RESIDENT IN REGISTERS (round 3).  The two-pass form reads dp and y
// twice (reduce, then dx: 131 MB for 82 MB of tensors per block at B=256 12x1000) behind two launches.  When a block's
// (dp, y) fits the register file of the chip — 49 MB against 131 MB, every block of the headline configuration — a grid
// of C x S <= #CUs workgroups of 1024 threads loads its slice ONCE (up to 16 pooling pairs per thread, all loads in flight
// together), reduces it, publishes its (sum da, sum da*xhat) partial, waits for the S - 1 workgroups that share its
// channel, and writes dY from the registers.
//   * The exchange needs NO ordering between different memory locations: a partial sum travels in ONE 64-bit word
//     together with its own "valid" tag — {float bits, 1} stored by a single agent-scope atomic store (performed at the
//     memory side, past the XCD's non-coherent L2; single-copy atomic for an aligned 8-byte word).  A waiter polls the 2 S
//     words of its channel with agent-scope atomic loads, one word per thread (s_sleep between polls), until every tag is
//     set: whoever sees the tag has the value.  (Round 3 published plain partials + a counter and relied on
//     s_waitcnt vmcnt(0) between them — an argument from hardware behaviour, not from the memory model.)
//     Every workgroup of the grid is resident (host: grid <= CU count; one 1024-thread workgroup per CU needs 128
//     registers and no LDS to speak of), so the wait ends at once in the normal case; it is BOUNDED (spin_polls), after
//     which the workgroup stops waiting and recomputes the missing partials itself (see SELF-SERVICE below): a co-tenant
//     that keeps siblings from becoming resident costs time, never correctness, and no wait can be circular.
//   * The words live in CALLER-OWNED memory (`counters`: ecg_bn_relu_pool_bwd_one_launch_counter_uints() uint32, zero before
//     the first launch) and return to zero by themselves: every workgroup counts itself out (`left[c]`, after the values it
//     polled have been consumed into LDS) and the last of the S to leave clears the 2 S words and the count.  The library
//     allocates nothing and keeps nothing between calls; two launches that may run concurrently need two buffers.
//   * S == 1 (C >= #CUs: the last block) needs no exchange at all and never touches `counters`.
// Arithmetic: the per-element formulas of the two kernels above; the partial sums associate differently (1024 threads,
// 16 waves), which the parity tests' tolerances cover like any other split count.
#ifndef ECG_BN_RES_MAX_S
#define ECG_BN_RES_MAX_S 8       // workgroups per channel the one-launch form accepts (compile-time A/B knob)
#endif
constexpr int kResThreads = 1024, kResPairs = 16, kResMaxC = 1024, kResMaxS = ECG_BN_RES_MAX_S, kResSpin = 512;

__device__ __forceinline__ unsigned long long res_word(float v) {      // {value bits, tag = 1}
    return (1ull << 32) | (unsigned long long)__float_as_uint(v);
}

template <bool AL8>
__global__ __launch_bounds__(kResThreads) void bn_bwd_resident_kernel(
    const float *__restrict__ y, const float *__restrict__ g, const float *__restrict__ gamma,
    const float *__restrict__ beta, const float *__restrict__ mean, const float *__restrict__ invstd,
    int S, double M, float *__restrict__ dgamma, float *__restrict__ dbeta,
    float *__restrict__ dy, int N, int C, int L, int ldy, float bcast, int train,
    unsigned long long *__restrict__ words, unsigned *__restrict__ left, int spin_limit) {
    __shared__ float redf[kResThreads / 64][2];
    __shared__ float kk[2];
    const int c = blockIdx.x, s = blockIdx.y, tl = threadIdx.x;
    const int n0 = (int)((long long)N * s / S), n1 = (int)((long long)N * (s + 1) / S);
    const float mu = mean[c], is = invstd[c], ga = gamma[c], sc = is * ga, gi = ga * is, be = beta[c];
    const int Lp = L >> 1, Lr = (L + 1) >> 1;            // pooled positions; pairs per row incl. the odd tail sample
    const int total = (n1 - n0) * Lr;

    // ---- the slice, once: pair i of this thread = flat pair tl + 1024 i of (sample, pair) ----
    float y0[kResPairs], y1[kResPairs], d[kResPairs];
#pragma unroll
    for (int i = 0; i < kResPairs; ++i) {
        const int idx = tl + kResThreads * i;
        const int ic = idx < total ? idx : 0;            // clamped: loads are unconditional (total >= 1)
        const int nl = ic / Lr, j = ic - nl * Lr;
        const size_t row = (size_t)(n0 + nl) * C + c;
        const float *r = y + row * L;
        if (AL8) ld_pair<true>(r + 2 * j, y0[i], y1[i]);
        else { y0[i] = r[2 * j]; y1[i] = r[min(2 * j + 1, L - 1)]; }
        d[i] = bcast != 0.f ? __fmul_rn(g[row], bcast) : (Lp > 0 ? g[row * (size_t)Lp + min(j, Lp - 1)] : 0.f);   // (a product that can
        // never be contracted into the sums below: the self-service path must reproduce these numbers bit for bit)
    }
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < kResPairs; ++i) {
        const int idx = tl + kResThreads * i;
        const int nl = idx / Lr, j = idx - nl * Lr;
        if (idx < total && j < Lp) {                     // (an odd tail sample never reaches the pool)
            int am;
            if (pool_route(y0[i], y1[i], mu, sc, be, am)) {
                a += d[i];
                q = __fmaf_rn(d[i], ((am ? y1[i] : y0[i]) - mu) * is, q);
            }
        }
    }
    a = wave_sum(a); q = wave_sum(q);
    if ((tl & 63) == 0) { redf[tl >> 6][0] = a; redf[tl >> 6][1] = q; }
    __syncthreads();
    __shared__ float sib[2 * kResMaxS], own[2];
    if (tl == 0) {
        float pa = 0.f, pq = 0.f;
#pragma unroll
        for (int w = 0; w < kResThreads / 64; ++w) { pa += redf[w][0]; pq += redf[w][1]; }
        own[0] = pa; own[1] = pq;
        if (S > 1) {            // value and tag in one word: nothing has to be ordered against anything else
            __hip_atomic_store(&words[((size_t)c * S + s) * 2], res_word(pa), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&words[((size_t)c * S + s) * 2 + 1], res_word(pq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    int mine_ok = 1;
    if (S > 1 && tl < 2 * S) {
        // one word per thread, all 2 S polled at once: ONE round trip in the normal case
        const unsigned long long *wp = &words[(size_t)c * S * 2 + tl];
        unsigned long long v = 0;
        int polls = 0;
        mine_ok = 0;
        while (polls < spin_limit) {                     // (spin_limit = 0, the test hook: nobody waits — always self-service)
            v = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) != 0u) { mine_ok = 1; break; }
            __builtin_amdgcn_s_sleep(2);
            ++polls;
        }
        if (mine_ok) sib[tl] = __uint_as_float((unsigned)v);
    }
    const int okf = __syncthreads_and(mine_ok);          // (also publishes own[] and sib[] to the workgroup)
    if (okf) {
        if (S == 1 && tl == 0) { sib[0] = own[0]; sib[1] = own[1]; }
    } else {
        // SELF-SERVICE (uniform branch): a sibling has not arrived within ~1 ms — it is not resident (another process's
        // workgroups, or a communication kernel waiting for a late peer, hold its CU).  Instead of waiting for it this
        // workgroup recomputes the siblings' partials from memory itself: the same pair -> thread mapping, the same order
        // of additions, the same combine — bit for bit the numbers the siblings will publish.  It then finishes and frees
        // its CU, so any circular wait between co-tenants dissolves; the late siblings find the counter where they need
        // it (every workgroup still arrives and leaves exactly once) and the counters reset as usual.
        for (int s2 = 0; s2 < S; ++s2) {
            if (s2 == s) { if (tl == 0) { sib[2 * s2] = own[0]; sib[2 * s2 + 1] = own[1]; } continue; }
            const int m0 = (int)((long long)N * s2 / S), m1 = (int)((long long)N * (s2 + 1) / S);
            const int tot2 = (m1 - m0) * Lr;
            float a2 = 0.f, q2 = 0.f;
            for (int i = 0; i < kResPairs; ++i) {
                const int idx = tl + kResThreads * i;
                if (idx >= tot2) break;
                const int nl = idx / Lr, j = idx - nl * Lr;
                if (j >= Lp) continue;
                const size_t row = (size_t)(m0 + nl) * C + c;
                const float *r = y + row * L;
                float u0, u1;
                if (AL8) ld_pair<true>(r + 2 * j, u0, u1);
                else { u0 = r[2 * j]; u1 = r[min(2 * j + 1, L - 1)]; }
                const float dd = bcast != 0.f ? __fmul_rn(g[row], bcast) : g[row * (size_t)Lp + j];
                int am;
                if (pool_route(u0, u1, mu, sc, be, am)) {
                    a2 += dd;
                    q2 = __fmaf_rn(dd, ((am ? u1 : u0) - mu) * is, q2);
                }
            }
            a2 = wave_sum(a2); q2 = wave_sum(q2);
            __syncthreads();
            if ((tl & 63) == 0) { redf[tl >> 6][0] = a2; redf[tl >> 6][1] = q2; }
            __syncthreads();
            if (tl == 0) {
                float pa = 0.f, pq = 0.f;
#pragma unroll
                for (int w = 0; w < kResThreads / 64; ++w) { pa += redf[w][0]; pq += redf[w][1]; }
                sib[2 * s2] = pa; sib[2 * s2 + 1] = pq;
            }
        }
    }
    __syncthreads();
    if (tl == 0) {
        double ta = 0.0, tq = 0.0;
        for (int p = 0; p < S; ++p) { ta += (double)sib[2 * p]; tq += (double)sib[2 * p + 1]; }     // split order: deterministic
        if (S > 1) {
            // count this workgroup out.  Its polled values have been CONSUMED (they sit in LDS, behind a barrier), and every
            // workgroup publishes before it leaves: when the count reaches S all 2 S words are set and nobody will read them
            // again, so the last one out clears them (and the count) for the next launch that is handed this buffer.
            // The clear must not be overtaken by this workgroup's OWN publish: on the normal path its threads 2s / 2s + 1 have
            // seen the tags (behind the barrier above); on the self-service path nobody has, so the publisher reads its two
            // words back (agent scope, the same location: the read returns once the store is visible where the clear will
            // land) before it counts out — no ordering between DIFFERENT locations is assumed anywhere.
            if (!okf)
                for (int w = 0; w < 2; ++w)
                    while ((unsigned)(__hip_atomic_load(&words[((size_t)c * S + s) * 2 + w], __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT) >> 32) == 0u)
                        __builtin_amdgcn_s_sleep(1);
            const unsigned gone = __hip_atomic_fetch_add(&left[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gone == (unsigned)S - 1) {
                for (int p = 0; p < 2 * S; ++p)
                    __hip_atomic_store(&words[(size_t)c * S * 2 + p], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&left[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (s == 0) {
            if (dbeta) dbeta[c] = (float)ta;
            if (dgamma) dgamma[c] = (float)tq;
        }
        kk[0] = train ? (float)(ta / M) : 0.f;
        kk[1] = train ? (float)(tq / M) : 0.f;
    }
    __syncthreads();
    const float k1 = kk[0], k2 = kk[1];

    // ---- dY from the registers ----
#pragma unroll
    for (int i = 0; i < kResPairs; ++i) {
        const int idx = tl + kResThreads * i;
        if (idx >= total) continue;
        const int nl = idx / Lr, j = idx - nl * Lr;
        float *dr = dy + ((size_t)(n0 + nl) * C + c) * (size_t)ldy;
        const int t = 2 * j;
        const bool has1 = t + 1 < L;
        float da0 = 0.f, da1 = 0.f;
        if (has1) {
            int am;
            if (pool_route(y0[i], y1[i], mu, sc, be, am)) {
                if (am) da1 = d[i]; else da0 = d[i];
            }
        }
        const float o0 = gi * (da0 - k1 - (y0[i] - mu) * is * k2);
        if (has1) {
            const float o1 = gi * (da1 - k1 - (y1[i] - mu) * is * k2);
            if (AL8 && (ldy & 1) == 0) *reinterpret_cast<float2 *>(dr + t) = make_float2(o0, o1);
            else { dr[t] = o0; dr[t + 1] = o1; }
        } else dr[t] = o0;
    }
    const int padw = ldy - L;                            // zero pad of the row-padded form
    for (int e = tl; e < (n1 - n0) * padw; e += kResThreads) {
        const int nl = e / padw, k = e - nl * padw;
        dy[((size_t)(n0 + nl) * C + c) * (size_t)ldy + L + k] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------
// unfused ReLU / MaxPool leaves
// ---------------------------------------------------------------------------------------
__global__ void relu_fwd_kernel(const float *__restrict__ x, float *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float v = x[i]; out[i] = v > 0.f ? v : 0.f; }
}
__global__ void relu_bwd_kernel(const float *__restrict__ out, const float *__restrict__ dout,
                                float *__restrict__ dx, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = out[i] > 0.f ? dout[i] : 0.f;
}
__global__ void maxpool2_fwd_kernel(const float *__restrict__ x, float *__restrict__ p, int L,
                                    int Lp, size_t total) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t row = idx / Lp;
    int j = (int)(idx - row * Lp);
    const float *r = x + row * L + 2 * j;
    p[idx] = r[1] > r[0] ? r[1] : r[0];
}
__global__ void maxpool2_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dp,
                                    float *__restrict__ dx, int L, int Lh, size_t total) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t row = idx / Lh;
    int j = (int)(idx - row * Lh);
    int t0 = 2 * j;
    const float *r = x + row * L;
    float *d = dx + row * L;
    if (t0 + 1 < L) {
        float v = dp[row * (size_t)(L >> 1) + j];
        bool am = r[t0 + 1] > r[t0];
        d[t0] = am ? 0.f : v;
        d[t0 + 1] = am ? v : 0.f;
    } else {
        d[t0] = 0.f;
    }
}

}  // namespace ecg

using namespace ecg;

// every pooling pair (2j, 2j+1) of every row is 8-byte aligned: even rows from an 8-byte aligned base
static bool pairs_aligned(const float *y, int L) { return (L & 1) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0; }

static int check_ncl(const char *who, int N, int C, int L) {
    ECG_REQUIRE(N > 0 && C > 0 && L > 0, "%s: N=%d C=%d L=%d must be > 0", who, N, C, L);
    ECG_REQUIRE(C <= 65535 && N <= 65535, "%s: N/C exceed grid limits", who);
    return ECG_OK;
}

ECG_API int ecg_bn_stat_partials_count(int N, int C, int L) { (void)L; return stat_splits(N, C); }

ECG_API int ecg_bn_stat_partials(const float *y, float *stat_partials, int N, int C, int L,
                                 ecg_stream_t stream) {
    int rc = check_ncl("bn_stat_partials", N, C, L);
    if (rc) return rc;
    ECG_REQUIRE(y && stat_partials, "bn_stat_partials: null pointer");
    int S = stat_splits(N, C);
    hipLaunchKernelGGL(bn_stat_partials_kernel, dim3(C, S), dim3(kBlock), 0, as_stream(stream), y,
                       stat_partials, N, C, L, S);
    return check_launch("bn_stat_partials_kernel");
}

ECG_API int ecg_bn_finalize(const float *stat_partials, int P, long long count, float *mean,
                            float *invstd, float *running_mean, float *running_var,
                            long long *num_batches_tracked, int C, float momentum, float eps,
                            ecg_stream_t stream) {
    ECG_REQUIRE(stat_partials && mean && invstd, "bn_finalize: null pointer");
    ECG_REQUIRE(P > 0 && C > 0 && count > 0, "bn_finalize: P=%d C=%d count=%lld", P, C, count);
    ECG_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
                "bn_finalize: running_mean/var must both be given or both NULL");
    const BnFin f{stat_partials, P, (double)count, mean, invstd, running_mean, running_var, num_batches_tracked,
                  momentum, eps};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, as_stream(stream), f);
    return check_launch("bn_finalize_kernel");
}

ECG_API int ecg_bn_invstd(const float *var, float *invstd, int C, float eps, ecg_stream_t stream) {
    ECG_REQUIRE(var && invstd && C > 0, "bn_invstd: bad argument");
    hipLaunchKernelGGL(bn_invstd_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), var,
                       invstd, C, eps);
    return check_launch("bn_invstd_kernel");
}

// ---- BN-apply + ReLU + MaxPool(2) forward, optionally with the statistics combine folded in (fin != NULL) ----------
static int check_fin(const char *who, const BnFin &f, int C) {
    ECG_REQUIRE(f.partials && f.mean && f.invstd, "%s: null pointer", who);
    ECG_REQUIRE(f.P > 0 && C > 0 && f.count > 0, "%s: P=%d C=%d count=%g", who, f.P, C, f.count);
    ECG_REQUIRE((f.running_mean == nullptr) == (f.running_var == nullptr),
                "%s: running_mean/var must both be given or both NULL", who);
    return ECG_OK;
}

static int pool_fwd_impl(const BnFin *fin, const float *y, const float *gamma, const float *beta, const float *mean,
                         const float *invstd, float *p, int N, int C, int L, hipStream_t st) {
    int rc = check_ncl("bn_relu_pool_fwd", N, C, L);
    if (rc) return rc;
    ECG_REQUIRE(y && gamma && beta && mean && invstd, "bn_relu_pool_fwd: null pointer");
    const int Lp = L / 2;
    if (Lp == 0 && !fin) return ECG_OK;   // MaxPool1d(2) of a length-1 row is empty
    ECG_REQUIRE(p || Lp == 0, "bn_relu_pool_fwd: null output");
    int S2 = cdiv(kStreamBlocks, C);
    if (S2 > N) S2 = N;
    const bool al8 = pairs_aligned(y, L);
    const BnFin f = fin ? *fin : BnFin{};
#define ECG_POOL(FIN, AL8) hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<FIN, AL8>), dim3(C, S2), dim3(kBlock), 0, st, y, gamma, \
                                              beta, mean, invstd, p, N, C, L, Lp, S2, f)
    if (fin) { if (al8) ECG_POOL(true, true); else ECG_POOL(true, false); }
    else { if (al8) ECG_POOL(false, true); else ECG_POOL(false, false); }
#undef ECG_POOL
    return check_launch("bn_relu_pool_fwd_kernel");
}

static int pool_gap_fwd_impl(const BnFin *fin, const float *y, const float *gamma, const float *beta,
                             const float *mean, const float *invstd, float *g, int N, int C, int L, hipStream_t st,
                             bool yh = false, int ldy = 0) {
    int rc = check_ncl("bn_relu_pool_gap_fwd", N, C, L);
    if (rc) return rc;
    ECG_REQUIRE(y && gamma && beta && mean && invstd && g, "bn_relu_pool_gap_fwd: null pointer");
    ECG_REQUIRE(L >= 2, "bn_relu_pool_gap_fwd: L=%d leaves an empty pooled row", L);
    int S2 = cdiv(kStreamBlocks, C);
    if (S2 > cdiv(N, 4)) S2 = cdiv(N, 4);          // four rows (one per wave) in flight per workgroup
    if (yh)         // bf16 activation storage (train mode: always with the statistics combine)
        hipLaunchKernelGGL((bn_relu_pool_gap_fwd_kernel<true, true>), dim3(C, S2), dim3(kBlock), 0, st, y, gamma, beta,
                           mean, invstd, g, N, C, L, L / 2, S2, *fin, ldy);
    else if (fin)
        hipLaunchKernelGGL(bn_relu_pool_gap_fwd_kernel<true>, dim3(C, S2), dim3(kBlock), 0, st, y, gamma, beta, mean,
                           invstd, g, N, C, L, L / 2, S2, *fin, L);
    else
        hipLaunchKernelGGL(bn_relu_pool_gap_fwd_kernel<false>, dim3(C, S2), dim3(kBlock), 0, st, y, gamma, beta, mean,
                           invstd, g, N, C, L, L / 2, S2, BnFin{}, L);
    return check_launch("bn_relu_pool_gap_fwd_kernel");
}

ECG_API int ecg_bn_relu_pool_fwd(const float *y, const float *gamma, const float *beta,
                                 const float *mean, const float *invstd, float *p, int N, int C,
                                 int L, ecg_stream_t stream) {
    return pool_fwd_impl(nullptr, y, gamma, beta, mean, invstd, p, N, C, L, as_stream(stream));
}

ECG_API int ecg_bn_relu_pool_gap_fwd(const float *y, const float *gamma, const float *beta,
                                     const float *mean, const float *invstd, float *g, int N,
                                     int C, int L, ecg_stream_t stream) {
    return pool_gap_fwd_impl(nullptr, y, gamma, beta, mean, invstd, g, N, C, L, as_stream(stream));
}

// ecg_bn_finalize + the pass above in ONE launch: mean / invstd are OUTPUTS here (and the running statistics and the
// counter are updated), written by one workgroup per channel while every workgroup re-derives them for itself.
// mode: 0 = pool -> out [N][C][L/2]; 1 = pool + global average -> out [N][C].
ECG_API int ecg_bn_stats_relu_pool_fwd(const float *stat_partials, int P, long long count, float *running_mean,
                                       float *running_var, long long *num_batches_tracked, float momentum, float eps,
                                       const float *y, const float *gamma, const float *beta, float *mean,
                                       float *invstd, float *out, int N, int C, int L, int mode, ecg_stream_t stream) {
    const BnFin f{stat_partials, P, (double)count, mean, invstd, running_mean, running_var, num_batches_tracked,
                  momentum, eps};
    int rc = check_fin("bn_stats_relu_pool_fwd", f, C);
    if (rc) return rc;
    ECG_REQUIRE(mode == 0 || mode == 1, "bn_stats_relu_pool_fwd: mode %d", mode);
    if (mode == 0) return pool_fwd_impl(&f, y, gamma, beta, mean, invstd, out, N, C, L, as_stream(stream));
    return pool_gap_fwd_impl(&f, y, gamma, beta, mean, invstd, out, N, C, L, as_stream(stream));
}

// pool + global average (the last block of the backbone) for a y that ecg_conv1d_fwd_bf16_yh wrote as bf16 [N][C][ldy]
// (the mixed-precision form; the pool-only pass on bf16 rows is ecg_bn_stats_relu_pool_fwd_h, bn_relu_pool_h.hip)
ECG_API int ecg_bn_stats_relu_pool_gap_fwd_yh(const float *stat_partials, int P, long long count, float *running_mean,
                                              float *running_var, long long *num_batches_tracked, float momentum,
                                              float eps, const void *y_bf16, int ldy, const float *gamma,
                                              const float *beta, float *mean, float *invstd, float *out, int N, int C,
                                              int L, ecg_stream_t stream) {
    const BnFin f{stat_partials, P, (double)count, mean, invstd, running_mean, running_var, num_batches_tracked,
                  momentum, eps};
    int rc = check_fin("bn_stats_relu_pool_gap_fwd_yh", f, C);
    if (rc) return rc;
    ECG_REQUIRE(y_bf16 && ldy >= L && ldy % 2 == 0 && (reinterpret_cast<uintptr_t>(y_bf16) & 3) == 0,
                "bn_stats_relu_pool_gap_fwd_yh: bf16 y needs an even row stride >= L and a 4-byte aligned base");
    return pool_gap_fwd_impl(&f, static_cast<const float *>(y_bf16), gamma, beta, mean, invstd, out, N, C, L,
                             as_stream(stream), true, ldy);
}

ECG_API size_t ecg_bn_relu_pool_bwd_ws_floats(int N, int C, int L) {
    (void)L;
    return (size_t)C * stat_splits(N, C, true) * 2 + (size_t)C * 2;      // (room for the bf16-storage pass's split count)
}
ECG_API size_t ecg_bn_bwd_ws_floats(int N, int C, int L) { return ecg_bn_relu_pool_bwd_ws_floats(N, C, L); }

// ---- the register-resident one-launch form: when a shape takes it (the caller owns the decision and the counters) ----
// Splits per channel, 0 = the shape does not qualify.  The only state is the CU count of each device, cached on first use
// (a device attribute, not a setting).
static int resident_splits(int N, int C, int L, int ldy) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    static std::atomic<int> cus[16];
    int ncu = cus[dev].load(std::memory_order_relaxed);
    if (!ncu) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        ncu = v > 0 ? v : -1;
        cus[dev].store(ncu, std::memory_order_relaxed);
    }
    if (ncu < C || C > kResMaxC) return 0;
    int S = ncu / C;
    if (S > N) S = N;
    if (S < 1) return 0;
    // measured per call at B=256 12x1000 (two-pass -> one launch; tools/bn_bwd_bench.py, round 4): S = 1 31.6 -> 24.9 us, S = 2
    // 30.1 -> 24.4, S = 4 29.3 -> 24.7, S = 8 (block 0) 28.6 -> 25.0.  (With round 3's counter protocol — one thread spinning, then
    // everybody loading the partials: two serial round trips — S = 8 gained nothing, 27.6 -> 27.2; with every word polled by its
    // own thread it is one round trip.)  At B=32 (an eighth of the bytes) the two short passes win (12-15 us against 14-19):
    // the one-launch form is for slices that fill at least half of its registers
    if (S > kResMaxS) return 0;
    const int Lr = (L + 1) / 2;
    const long long pairs = (long long)cdiv(N, S) * Lr;
    if (pairs > (long long)kResThreads * kResPairs || pairs < (long long)kResThreads * kResPairs / 2) return 0;
    if ((long long)N * C * (ldy > L ? ldy : L) >= (1LL << 31)) return 0;
    return S;
}

template <bool FUSED>
static int bn_bwd_impl(const float *y, const float *g, const float *gamma, const float *beta,
                       const float *mean, const float *invstd, float *dy, float *dgamma,
                       float *dbeta, float *ws, int N, int C, int L, int ldy, int train,
                       hipStream_t st, float bcast = 0.f) {
    const int S = stat_splits(N, C);
    float *partials = ws;
    const bool al8 = FUSED && pairs_aligned(y, L);
    if (al8)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<FUSED, true>), dim3(C, S), dim3(kBlock), 0, st, y, g, gamma,
                           beta, mean, invstd, partials, N, C, L, S, bcast);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<FUSED, false>), dim3(C, S), dim3(kBlock), 0, st, y, g, gamma,
                           beta, mean, invstd, partials, N, C, L, S, bcast);
    int rc = check_launch("bn_bwd_reduce_kernel");
    if (rc) return rc;
    // dx pass with the combine of the reduce partials folded in (no finalize launch)
    const int Lh = (ldy + 1) / 2;
    int S2 = cdiv(kStreamBlocks, C);            // one round of resident workgroups
    if (S2 > N) S2 = N;
    if (pairs_aligned(y, L) && L >= 2)
        hipLaunchKernelGGL((bn_bwd_dx_kernel<FUSED, true>), dim3(C, S2), dim3(kBlock), 0, st, y, g, gamma, beta, mean,
                           invstd, partials, S, (double)N * L, dgamma, dbeta, dy, N, C, L, ldy, Lh, S2, bcast, train);
    else
        hipLaunchKernelGGL((bn_bwd_dx_kernel<FUSED, false>), dim3(C, S2), dim3(kBlock), 0, st, y, g, gamma, beta, mean,
                           invstd, partials, S, (double)N * L, dgamma, dbeta, dy, N, C, L, ldy, Lh, S2, bcast, train);
    return check_launch("bn_bwd_dx_kernel");
}

ECG_API int ecg_bn_relu_pool_bwd_one_launch_splits(int N, int C, int L, int ldy) {
    if (N <= 0 || C <= 0 || L <= 0 || ldy < L) return 0;
    return resident_splits(N, C, L, ldy);
}

ECG_API size_t ecg_bn_relu_pool_bwd_one_launch_counter_uints(int N, int C, int L, int ldy) {
    const int S = ecg_bn_relu_pool_bwd_one_launch_splits(N, C, L, ldy);
    return S > 1 ? (size_t)C * (4 * (size_t)S + 1) : 0;      // 2 S 64-bit words per channel + one leave count per channel
}

ECG_API int ecg_bn_relu_pool_bwd_one_launch(const float *y, const float *dp, const float *gamma, const float *beta,
                                            const float *mean, const float *invstd, float *dy, int ldy,
                                            float *dgamma, float *dbeta, uint32_t *counters, int N, int C, int L,
                                            int train, int gap, int spin_polls, ecg_stream_t stream) {
    int rc = check_ncl("bn_relu_pool_bwd_one_launch", N, C, L);
    if (rc) return rc;
    ECG_REQUIRE(y && dp && gamma && beta && mean && invstd && dy, "bn_relu_pool_bwd_one_launch: null pointer");
    ECG_REQUIRE(ldy >= L, "bn_relu_pool_bwd_one_launch: dY row stride %d < row length %d", ldy, L);
    const int S = resident_splits(N, C, L, ldy);
    ECG_REQUIRE(S > 0, "bn_relu_pool_bwd_one_launch: N=%d C=%d L=%d does not take the one-launch form on this device "
                "(ask ecg_bn_relu_pool_bwd_one_launch_splits first)", N, C, L);
    ECG_REQUIRE(S == 1 || (counters && (reinterpret_cast<uintptr_t>(counters) & 7) == 0),
                "bn_relu_pool_bwd_one_launch: %d workgroups per channel need the caller's 8-byte aligned counter buffer", S);
    unsigned long long *words = reinterpret_cast<unsigned long long *>(counters);
    unsigned *left = S > 1 ? reinterpret_cast<unsigned *>(counters) + (size_t)C * 4 * S : nullptr;
    const int spin = spin_polls < 0 ? kResSpin : spin_polls;
    const float bcast = gap ? 1.0f / (float)(L / 2) : 0.f;
    ECG_REQUIRE(!gap || L >= 2, "bn_relu_pool_bwd_one_launch: L=%d leaves an empty pooled row", L);
    hipStream_t st = as_stream(stream);
    if (pairs_aligned(y, L))
        hipLaunchKernelGGL((bn_bwd_resident_kernel<true>), dim3(C, S), dim3(kResThreads), 0, st, y, dp, gamma, beta, mean,
                           invstd, S, (double)N * L, dgamma, dbeta, dy, N, C, L, ldy, bcast, train, words, left, spin);
    else
        hipLaunchKernelGGL((bn_bwd_resident_kernel<false>), dim3(C, S), dim3(kResThreads), 0, st, y, dp, gamma, beta, mean,
                           invstd, S, (double)N * L, dgamma, dbeta, dy, N, C, L, ldy, bcast, train, words, left, spin);
    return check_launch("bn_bwd_resident_kernel");
}

ECG_API int ecg_bn_relu_pool_bwd_ld(const float *y, const float *dp, const float *gamma,
                                    const float *beta, const float *mean, const float *invstd,
                                    float *dy, int ldy, float *dgamma, float *dbeta, float *ws,
                                    int N, int C, int L, int train, ecg_stream_t stream) {
    int rc = check_ncl("bn_relu_pool_bwd", N, C, L);
    if (rc) return rc;
    ECG_REQUIRE(y && gamma && beta && mean && invstd && dy && ws, "bn_relu_pool_bwd: null pointer");
    ECG_REQUIRE(dp || L < 2, "bn_relu_pool_bwd: dp is NULL");
    ECG_REQUIRE(ldy >= L, "bn_relu_pool_bwd: dY row stride %d < row length %d", ldy, L);
    return bn_bwd_impl<true>(y, dp, gamma, beta, mean, invstd, dy, dgamma, dbeta, ws, N, C, L, ldy,
                             train, as_stream(stream));
}

ECG_API int ecg_bn_relu_pool_bwd(const float *y, const float *dp, const float *gamma,
                                 const float *beta, const float *mean, const float *invstd,
                                 float *dy, float *dgamma, float *dbeta, float *ws, int N, int C,
                                 int L, int train, ecg_stream_t stream) {
    return ecg_bn_relu_pool_bwd_ld(y, dp, gamma, beta, mean, invstd, dy, L, dgamma, dbeta, ws, N, C, L,
                                   train, stream);
}

ECG_API int ecg_bn_apply_fwd(const float *y, const float *gamma, const float *beta,
                             const float *mean, const float *invstd, float *out, int N, int C,
                             int L, ecg_stream_t stream) {
    int rc = check_ncl("bn_apply_fwd", N, C, L);
    if (rc) return rc;
    ECG_REQUIRE(y && gamma && beta && mean && invstd && out, "bn_apply_fwd: null pointer");
    size_t total = (size_t)N * C * L;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(cdiv(total, kBlock)), dim3(kBlock), 0,
                       as_stream(stream), y, gamma, beta, mean, invstd, out, C, L, total);
    return check_launch("bn_apply_fwd_kernel");
}

ECG_API int ecg_bn_bwd(const float *y, const float *dout, const float *gamma, const float *mean,
                       const float *invstd, float *dy, float *dgamma, float *dbeta, float *ws,
                       int N, int C, int L, int train, ecg_stream_t stream) {
    int rc = check_ncl("bn_bwd", N, C, L);
    if (rc) return rc;
    ECG_REQUIRE(y && dout && gamma && mean && invstd && dy && ws, "bn_bwd: null pointer");
    return bn_bwd_impl<false>(y, dout, gamma, nullptr, mean, invstd, dy, dgamma, dbeta, ws, N, C, L, L,
                              train, as_stream(stream));
}

ECG_API int ecg_relu_fwd(const float *x, float *out, size_t n, ecg_stream_t stream) {
    ECG_REQUIRE(x && out, "relu_fwd: null pointer");
    if (n == 0) return ECG_OK;
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), x, out, n);
    return check_launch("relu_fwd_kernel");
}
ECG_API int ecg_relu_bwd(const float *out, const float *dout, float *dx, size_t n,
                         ecg_stream_t stream) {
    ECG_REQUIRE(out && dout && dx, "relu_bwd: null pointer");
    if (n == 0) return ECG_OK;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), out,
                       dout, dx, n);
    return check_launch("relu_bwd_kernel");
}
ECG_API int ecg_maxpool2_fwd(const float *x, float *p, int rows, int L, ecg_stream_t stream) {
    ECG_REQUIRE(x && rows > 0 && L > 0, "maxpool2_fwd: bad argument");
    int Lp = L / 2;
    if (Lp == 0) return ECG_OK;
    ECG_REQUIRE(p, "maxpool2_fwd: null output");
    size_t total = (size_t)rows * Lp;
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream),
                       x, p, L, Lp, total);
    return check_launch("maxpool2_fwd_kernel");
}
ECG_API int ecg_maxpool2_bwd(const float *x, const float *dp, float *dx, int rows, int L,
                             ecg_stream_t stream) {
    ECG_REQUIRE(x && dx && rows > 0 && L > 0, "maxpool2_bwd: bad argument");
    ECG_REQUIRE(dp || L < 2, "maxpool2_bwd: dp is NULL");
    int Lh = (L + 1) / 2;
    size_t total = (size_t)rows * Lh;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream),
                       x, dp, dx, L, Lh, total);
    return check_launch("maxpool2_bwd_kernel");
}

ECG_API int ecg_bn_relu_pool_gap_bwd_ld(const float *y, const float *dg, const float *gamma,
                                        const float *beta, const float *mean, const float *invstd,
                                        float *dy, int ldy, float *dgamma, float *dbeta, float *ws,
                                        int N, int C, int L, int train, ecg_stream_t stream) {
    int rc = check_ncl("bn_relu_pool_gap_bwd", N, C, L);
    if (rc) return rc;
    ECG_REQUIRE(y && dg && gamma && beta && mean && invstd && dy && ws,
                "bn_relu_pool_gap_bwd: null pointer");
    ECG_REQUIRE(L >= 2, "bn_relu_pool_gap_bwd: L=%d leaves an empty pooled row", L);
    ECG_REQUIRE(ldy >= L, "bn_relu_pool_gap_bwd: dY row stride %d < row length %d", ldy, L);
    return bn_bwd_impl<true>(y, dg, gamma, beta, mean, invstd, dy, dgamma, dbeta, ws, N, C, L, ldy,
                             train, as_stream(stream), 1.0f / (float)(L / 2));
}

ECG_API int ecg_bn_relu_pool_gap_bwd(const float *y, const float *dg, const float *gamma,
                                     const float *beta, const float *mean, const float *invstd,
                                     float *dy, float *dgamma, float *dbeta, float *ws, int N,
                                     int C, int L, int train, ecg_stream_t stream) {
    return ecg_bn_relu_pool_gap_bwd_ld(y, dg, gamma, beta, mean, invstd, dy, L, dgamma, dbeta, ws, N, C,
                                       L, train, stream);
}
