// input.hip — the step BEFORE the model: WFDB format-16 samples -> per-lead z-scored fp32 windows
// (reference: src/datasets/ptbxl.py:14-41 `_load_ecg` = wfdb.rdsamp + float32 cast + transpose,
//  :122-127 `_normalize`; same code in ptbxl_ecg_multimodal.py:15-36,98-103 and ptbxl_af.py).
//
// The reference normalises a TRANSPOSED VIEW ([12,T] over a [T,12] buffer), so numpy reduces along
// the strided axis: a plain left-to-right float32 sum per lead, no pairwise tree.  These kernels
// reproduce exactly that arithmetic — results are bit-identical to the reference's arrays
// (tests/golden/g8_input_pipeline.npz holds three of its committed demo windows):
//     p    = float32( (double)(d - baseline) / gain )          wfdb 4.3.0 Record.dac, then the cast
//     mean = float32(sum_seq(p)) / float32(T)
//     std  = sqrt( float32(sum_seq((p-mean)*(p-mean))) / float32(T) ) + 1e-6f
//     out  = (p - mean) / std
// Every operation is a separately rounded IEEE fp32 op: contraction is switched off for this file
// (pragma below and -ffp-contract=off in the Makefile), and hipcc never reassociates without
// fast-math; division and sqrt are the correctly rounded expansions (HIP default).
//
// Three HBM-streaming launches per batch:
//   wfdb16_physical_kernel   int16 [B][T][leads] -> fp32 [B][leads][T]   (LDS-tiled transpose)
//   zscore_stats_kernel      one LANE per (window, lead) row walks its row twice (the sequential
//                            sums are the semantics; 3072 rows at B=256 -> latency-, not HBM-bound)
//   zscore_apply_kernel      elementwise, float4
#include "common.h"

// hipcc contracts a*b+c into an fma by default (-ffp-contract=fast) — and does so even through the
// __fmul_rn/__fadd_rn wrappers, whose bodies are compiled under the header's own state: one rounding
// where numpy does two.  Plain operators under this pragma stay separate.
#pragma clang fp contract(off)

namespace ecg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTT = 256;        // time samples per transpose tile
constexpr int kMaxLeads = 16;

__global__ __launch_bounds__(256) void wfdb16_physical_kernel(
    const int16_t *__restrict__ d, const double *__restrict__ gain, const int *__restrict__ baseline,
    float *__restrict__ out, int T, int leads) {
    __shared__ int16_t tile[kTT * kMaxLeads];
    const int b = blockIdx.y, t0 = blockIdx.x * kTT, tid = threadIdx.x;
    const int nt = min(kTT, T - t0);
    const int16_t *src = d + ((size_t)b * T + t0) * leads;     // nt*leads contiguous samples
    const int count = nt * leads;
    for (int e = tid; e < count; e += 256) tile[e] = src[e];
    __syncthreads();
    if (tid < nt) {
        for (int l = 0; l < leads; ++l) {
            const int v = tile[tid * leads + l];
            const double g = gain[(size_t)b * leads + l];
            const int base = baseline[(size_t)b * leads + l];
            // format 16 reserves -32768 as "invalid sample": wfdb returns NaN for it
            const float p = (v == -32768) ? __builtin_nanf("") : (float)((double)(v - base) / g);
            out[((size_t)b * leads + l) * T + t0 + tid] = p;
        }
    }
}

// Left-to-right walk over one row with the loads kept well ahead of the dependent add chain: the
// row streams through two register batches of kNB float4 (the next batch is in flight while the
// chain consumes the current one).  Loads are unconditional with clamped addresses; elements past
// T are skipped at the chain.
constexpr int kNB = 8;

template <typename F>
__device__ __forceinline__ void walk_row_vec(const float *__restrict__ r, int T, F &&step) {
    const int n4 = T >> 2;                      // T % 4 == 0 on this path
    const f32x4 *r4 = reinterpret_cast<const f32x4 *>(r);
    f32x4 cur[kNB], nxt[kNB];
#pragma unroll
    for (int j = 0; j < kNB; ++j) cur[j] = r4[min(j, n4 - 1)];
    for (int i = 0; i < n4; i += kNB) {
#pragma unroll
        for (int j = 0; j < kNB; ++j) nxt[j] = r4[min(i + kNB + j, n4 - 1)];
#pragma unroll
        for (int j = 0; j < kNB; ++j) {
            if (i + j < n4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) step(cur[j][k]);
            }
        }
#pragma unroll
        for (int j = 0; j < kNB; ++j) cur[j] = nxt[j];
    }
}

// LANES active lanes per wave, one row each (fewer lanes per wave = more waves = more CUs busy and
// fewer distinct cache lines per load instruction when there are few rows).
template <int LANES>
__global__ __launch_bounds__(64) void zscore_stats_kernel(const float *__restrict__ x,
                                                         float *__restrict__ stats, int rows, int T) {
    const int lane = threadIdx.x;
    const int row = blockIdx.x * LANES + lane;
    if (lane >= LANES || row >= rows) return;
    const float *r = x + (size_t)row * T;
    const float n = (float)T;
    const bool vec = ((T & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    float acc = 0.f;
    if (vec) walk_row_vec(r, T, [&](float v) { acc = acc + v; });
    else for (int t = 0; t < T; ++t) acc = acc + r[t];
    const float mean = acc / n;
    float q = 0.f;
    auto sq_step = [&](float v) {
        const float dv = v - mean;
        const float sq = dv * dv;
        q = q + sq;
    };
    if (vec) walk_row_vec(r, T, sq_step);
    else for (int t = 0; t < T; ++t) sq_step(r[t]);
    stats[2 * (size_t)row] = mean;
    // sqrt through fp64: rounding a double sqrt to float is the correctly rounded float sqrt
    const float sd = (float)sqrt((double)(q / n));
    stats[2 * (size_t)row + 1] = sd + 1e-6f;
}

__global__ __launch_bounds__(256) void zscore_apply_kernel(const float *__restrict__ x,
                                                          const float *__restrict__ stats,
                                                          float *__restrict__ out, int T, int T4) {
    // grid = (ceil(T4/256), rows); T4 = T/4 when rows are float4-addressable, else 0 (scalar walk)
    const size_t row = blockIdx.y;
    const float mean = stats[2 * row], sd = stats[2 * row + 1];
    const float *r = x + row * T;
    float *o = out + row * T;
    if (T4) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= T4) return;
        const f32x4 v = reinterpret_cast<const f32x4 *>(r)[i];
        f32x4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (v[k] - mean) / sd;
        reinterpret_cast<f32x4 *>(o)[i] = w;
    } else {
        const int t = blockIdx.x * 256 + threadIdx.x;
        if (t < T) o[t] = (r[t] - mean) / sd;
    }
}

}  // namespace ecg

using namespace ecg;

ECG_API int ecg_wfdb16_physical(const int16_t *d, const double *gain, const int *baseline, float *out,
                                int B, int T, int leads, ecg_stream_t stream) {
    ECG_REQUIRE(d && gain && baseline && out, "wfdb16_physical: null pointer");
    ECG_REQUIRE(B > 0 && T > 0, "wfdb16_physical: B=%d T=%d must be > 0", B, T);
    ECG_REQUIRE(leads >= 1 && leads <= kMaxLeads, "wfdb16_physical: leads=%d outside [1,%d]", leads, kMaxLeads);
    ECG_REQUIRE(B <= 65535, "wfdb16_physical: B=%d exceeds grid.y limit 65535", B);
    hipLaunchKernelGGL(wfdb16_physical_kernel, dim3(cdiv(T, kTT), B), dim3(256), 0, as_stream(stream), d,
                       gain, baseline, out, T, leads);
    return check_launch("wfdb16_physical_kernel");
}

ECG_API int ecg_zscore_rows(const float *x, float *out, float *stats, int rows, int T,
                            ecg_stream_t stream) {
    ECG_REQUIRE(x && out && stats, "zscore_rows: null pointer");
    ECG_REQUIRE(rows > 0 && T > 0, "zscore_rows: rows=%d T=%d must be > 0", rows, T);
    ECG_REQUIRE(rows <= 65535, "zscore_rows: rows=%d exceeds grid.y limit 65535 (split the batch)", rows);
    hipStream_t st = as_stream(stream);
    if (rows >= 64 * 1024)
        hipLaunchKernelGGL((zscore_stats_kernel<64>), dim3(cdiv(rows, 64)), dim3(64), 0, st, x, stats, rows, T);
    else
        hipLaunchKernelGGL((zscore_stats_kernel<16>), dim3(cdiv(rows, 16)), dim3(64), 0, st, x, stats, rows, T);
    int rc = check_launch("zscore_stats_kernel");
    if (rc) return rc;
    const bool vec = (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    const int T4 = vec ? T / 4 : 0;
    hipLaunchKernelGGL(zscore_apply_kernel, dim3(cdiv(vec ? T4 : T, 256), rows), dim3(256), 0, st, x, stats,
                       out, T, T4);
    return check_launch("zscore_apply_kernel");
}
