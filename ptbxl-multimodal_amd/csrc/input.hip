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
    const int n4 = T >> 2, tail = T & 3;        // rows are readable up to the next multiple of 4
    const int m4 = n4 + (tail ? 1 : 0);
    const f32x4 *r4 = reinterpret_cast<const f32x4 *>(r);
    f32x4 a[kNB], b[kNB];                       // ping-pong: no register copies between batches (a copy
                                                // would wait for the loads it is meant to hide)
    auto fill = [&](f32x4 *buf, int i) {
#pragma unroll
        for (int j = 0; j < kNB; ++j) buf[j] = r4[min(i + j, m4 - 1)];
    };
    auto consume_full = [&](const f32x4 *buf) {     // a whole batch: straight-line chain, no guards
#pragma unroll
        for (int j = 0; j < kNB; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) step(buf[j][k]);
    };
    auto consume_guarded = [&](const f32x4 *buf, int i) {   // the last, partial batch only
#pragma unroll
        for (int j = 0; j < kNB; ++j) {
            if (i + j < n4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) step(buf[j][k]);
            } else if (i + j == n4) {           // ragged last float4 (only when T % 4 != 0)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (k < tail) step(buf[j][k]);
            }
        }
    };
    const int nb = n4 / kNB;                    // full batches
    fill(a, 0);
    int bi = 0;
    for (; bi + 2 <= nb; bi += 2) {
        fill(b, (bi + 1) * kNB);
        consume_full(a);
        fill(a, (bi + 2) * kNB);
        consume_full(b);
    }
    if (bi < nb) {                              // one full batch left, already in a
        fill(b, (bi + 1) * kNB);
        consume_full(a);
        ++bi;
        if (bi * kNB < m4) consume_guarded(b, bi * kNB);
    } else if (bi * kNB < m4) {
        consume_guarded(a, bi * kNB);
    }
}

// (mean, std + 1e-6) of one row exactly as numpy computes them for the reference (see the header)
__device__ __forceinline__ void row_stats_vec(const float *__restrict__ r, int T, float &mean, float &sd) {
    const float n = (float)T;
    float acc = 0.f;
    walk_row_vec(r, T, [&](float v) { acc = acc + v; });
    mean = acc / n;
    const float mu = mean;
    float q = 0.f;
    walk_row_vec(r, T, [&](float v) {
        const float dv = v - mu;
        const float sq = dv * dv;
        q = q + sq;
    });
    // sqrt through fp64: rounding a double sqrt to float is the correctly rounded float sqrt
    sd = (float)sqrt((double)(q / n)) + 1e-6f;
}

// ---------------------------------------------------------------------------------------
// Fused path: one workgroup per (window, group of G leads) keeps its G physical rows in LDS
// ([G][Tpad] fp32, Tpad % 64 == 4 so that the G chain lanes' ds_read_b128 hit disjoint banks):
//   phase 1  all threads: int16 samples -> physical fp32 -> LDS (global reads of this group's leads)
//   phase 2  lanes 0..G-1: the two left-to-right chains per row, out of LDS
//   phase 3  all threads: (p - mean)/std -> out, coalesced
// HBM traffic is the algorithmic minimum: 2 B/sample in, 4 B/sample out.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wfdb16_zscore_fused_kernel(
    const int16_t *__restrict__ d, const double *__restrict__ gain, const int *__restrict__ baseline,
    float *__restrict__ out, float *__restrict__ stats, int T, int leads, int G, int Tpad) {
    extern __shared__ __attribute__((aligned(16))) float phys[];      // [G][Tpad] (+ 2*G stats behind)
    const int b = blockIdx.y, l0 = blockIdx.x * G, tid = threadIdx.x;
    const int nl = min(G, leads - l0);
    float *st = phys + (size_t)G * Tpad;
    __shared__ double sg[kMaxLeads];
    __shared__ int sb[kMaxLeads];
    if (tid < nl) {
        sg[tid] = gain[(size_t)b * leads + l0 + tid];
        sb[tid] = baseline[(size_t)b * leads + l0 + tid];
    }
    __syncthreads();
    const int16_t *src = d + (size_t)b * T * leads + l0;
    for (int t = tid; t < T; t += 256) {               // thread <-> time sample: nl consecutive int16
        const int16_t *p = src + (size_t)t * leads;
        for (int l = 0; l < nl; ++l) {
            const int v = p[l];
            phys[(size_t)l * Tpad + t] = (v == -32768) ? __builtin_nanf("") : (float)((double)(v - sb[l]) / sg[l]);
        }
    }
    __syncthreads();
    if (tid < nl) {
        float mean, sd;
        row_stats_vec(phys + (size_t)tid * Tpad, T, mean, sd);
        st[2 * tid] = mean;
        st[2 * tid + 1] = sd;
        stats[2 * ((size_t)b * leads + l0 + tid)] = mean;
        stats[2 * ((size_t)b * leads + l0 + tid) + 1] = sd;
    }
    __syncthreads();
    const bool vec = ((T & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    for (int l = 0; l < nl; ++l) {
        const float mean = st[2 * l], sd = st[2 * l + 1];
        const float *row = phys + (size_t)l * Tpad;
        float *o = out + ((size_t)b * leads + l0 + l) * T;
        if (vec) {
            for (int i = tid; i < (T >> 2); i += 256) {
                const f32x4 v = reinterpret_cast<const f32x4 *>(row)[i];
                f32x4 w;
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = (v[k] - mean) / sd;
                reinterpret_cast<f32x4 *>(o)[i] = w;
            }
        } else {
            for (int t = tid; t < T; t += 256) o[t] = (row[t] - mean) / sd;
        }
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
    // float4 walk needs 16-byte aligned rows: T % 4 == 0 and an aligned base
    const bool vec = ((T & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    float mean, sd;
    if (vec) {
        row_stats_vec(r, T, mean, sd);
    } else {
        const float n = (float)T;
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc = acc + r[t];
        mean = acc / n;
        float q = 0.f;
        for (int t = 0; t < T; ++t) {
            const float dv = r[t] - mean;
            const float sq = dv * dv;
            q = q + sq;
        }
        sd = (float)sqrt((double)(q / n)) + 1e-6f;
    }
    stats[2 * (size_t)row] = mean;
    stats[2 * (size_t)row + 1] = sd;
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

// Fused int16 -> z-scored fp32 (the LDS-resident kernel above) when a whole window fits in LDS,
// otherwise the three streaming launches.  stats [B*leads][2] receives (mean, std + 1e-6).
ECG_API int ecg_wfdb16_zscore(const int16_t *d, const double *gain, const int *baseline, float *out,
                              float *stats, int B, int T, int leads, ecg_stream_t stream) {
    ECG_REQUIRE(d && gain && baseline && out && stats, "wfdb16_zscore: null pointer");
    ECG_REQUIRE(B > 0 && T > 0, "wfdb16_zscore: B=%d T=%d must be > 0", B, T);
    ECG_REQUIRE(leads >= 1 && leads <= kMaxLeads, "wfdb16_zscore: leads=%d outside [1,%d]", leads, kMaxLeads);
    ECG_REQUIRE(B <= 65535, "wfdb16_zscore: B=%d exceeds grid.y limit 65535", B);
    int Tpad = (T + 3) / 4 * 4;
    Tpad += (4 - Tpad % 64 + 64) % 64;                 // row stride == 4 (mod 64 banks)
    const size_t row_bytes = (size_t)Tpad * 4;
    // Plan (measured on MI355X, B=256): the fused kernel wins while ALL leads of a window fit in one
    // workgroup's <= 64 KB of LDS (12x1000: 30 us vs 54 us streamed).  Longer windows would have to
    // split the leads over workgroups; LDS then caps the rows resident per CU below what the chains
    // need to overlap (12x5000: 220 us fused with 3 leads per workgroup vs 138 us streamed), so they
    // take the three streaming launches instead.
    const int G = (row_bytes * leads + (size_t)leads * 8 <= 64u * 1024u) ? leads : 0;
    if (G == 0) {
        int rc = ecg_wfdb16_physical(d, gain, baseline, out, B, T, leads, stream);
        if (rc) return rc;
        ECG_REQUIRE((long long)B * leads <= 65535, "wfdb16_zscore: B*leads=%lld exceeds 65535 rows for T=%d",
                    (long long)B * leads, T);
        return ecg_zscore_rows(out, out, stats, B * leads, T, stream);
    }
    const size_t lds = (size_t)G * row_bytes + (size_t)G * 2 * sizeof(float);
    hipLaunchKernelGGL(wfdb16_zscore_fused_kernel, dim3(cdiv(leads, G), B), dim3(256), lds, as_stream(stream), d,
                       gain, baseline, out, stats, T, leads, G, Tpad);
    return check_launch("wfdb16_zscore_fused_kernel");
}
