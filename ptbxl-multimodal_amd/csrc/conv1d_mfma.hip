// conv1d_mfma.hip — Conv1d forward / input-grad / weight-grad as im2col-FREE implicit GEMMs on the
// gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 157 TFLOP/s peak — the
// same peak as the fp32 VALU but at 1/16 of the instruction issue and one operand VGPR per MFMA).
//
// No im2col buffer ever exists: the x tile (with its K-1 halo) sits in LDS once and the "im2col
// column" for tap k is just the same LDS row read at offset +k.
//
//   forward / input-grad:  D[co][t] += sum_{ci,k} W[k][ci][co] * X[ci][t+k-pad]
//       MFMA A = W fragment  (lane l: co = l&31, ci-pair member l>>5, fixed tap)
//       MFMA B = X fragment  (lane l: t  = l&31, ci-pair member l>>5, shifted by tap)
//     the reduction runs over (tap, ci-pair); both operand reads are 32 consecutive floats per
//     half-wave -> conflict-free ds_read_b32.
//   weight-grad:           D[co][r] += sum_{n,t} dY[co][t] * X[ci(r)][t+k(r)-pad],  r = ci*K + k
//       MFMA A = dY fragment (lane l: co = l&31, t-pair member l>>5)   row stride odd -> conflict-free
//       MFMA B = X  fragment (lane l: r  = l&31, t-pair member l>>5)   row stride == K mod 32 makes
//                 LDS address == r + const (mod 32) -> conflict-free
//     split over N into slabs that a fixed-order reduce sums (deterministic, no atomics).
//
// Replaces the ATen work behind ConvBlock.net[0] (reference src/models/ecg_cnn.py:13) and its
// backward (src/training/loop.py:33).
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace ecg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in VGPRs (HIP's float4 struct did not)

constexpr int kKM = 15;   // largest kernel size the MFMA path stages (the reference uses 15)

// In-kernel stamps (diagnostic build only: make STAMP=1 -> tools/_build/libecg_hip_stamp.so; the product library has none).
// Wave 0 of every workgroup writes s_memtime at a few points into a buffer no other code reads.
#ifdef ECG_STAMP
__device__ unsigned long long *g_stamps = nullptr;
#define ECG_STAMP_AT(slot) do { if (g_stamps && threadIdx.x == 0) { \
    g_stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    if ((slot) == 0) g_stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime(); \
    if ((slot) == 4) g_stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define ECG_STAMP_AT(slot) do { } while (0)
#endif

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// s_waitcnt lgkmcnt(n), vmcnt / expcnt left at "no wait" (n is a compile-time constant after unrolling).  hipcc's own
// placement drains lgkmcnt(0) on every second step of the fragment double-buffering below — i.e. it waits for the LDS
// reads just issued for the NEXT step in front of MFMAs that only need the previous ones; an explicit counted wait
// placed before the MFMAs tells its scoreboard the operands are complete (a count larger than the reads really in
// flight is harmless: the compiler still adds whatever wait correctness needs).
// One LDS-DMA piece (1 KB per wave) from INLINE ASM: wave-uniform 64-bit base in SGPRs + this lane's 32-bit byte offset ->
// LDS byte address `dst` (wave-uniform, + 16 * lane implied).  hipcc books __builtin_amdgcn_global_load_lds like a FLAT
// access: every LDS read that is pending when one issues is later waited for with lgkmcnt(0), together with the fragments
// read since (22 of the 30 steps of a forward chunk).  The asm statement has no register result, so there is nothing for
// the compiler to protect; M0 is saved and restored inside it; completion is waited for explicitly (vmcnt) before the
// barrier that publishes the image.
__device__ __forceinline__ void glds16(const void *base_in, unsigned voff, unsigned dst) {
    // (uniform by construction; readfirstlane makes it PROVABLY so for the "s" operand — the diagnostic STAMP build could
    // not prove it on its own)
    const unsigned long long b = (unsigned long long)base_in;
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    const void *base = (const void *)(((unsigned long long)bhi << 32) | blo);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

__device__ __forceinline__ void wait_lgkm_f(int n) {
    switch (n) {
        case 1: __builtin_amdgcn_s_waitcnt(0xC17F); break;
        case 2: __builtin_amdgcn_s_waitcnt(0xC27F); break;
        case 3: __builtin_amdgcn_s_waitcnt(0xC37F); break;
        case 4: __builtin_amdgcn_s_waitcnt(0xC47F); break;
        default: break;
    }
}

// row of element r of a 32x32 accumulator held by this lane: (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// =======================================================================================
// forward (and input-grad with flipped/transposed packed weights)
// =======================================================================================
// grid = (ceil(Lo/T_T), Cout/CO_T, N), 256 threads = 4 waves laid out WCO x WT over the tile.
//
// Pipeline (per chunk of CI_C = 4 input channels = 30 reduction steps = 120 MFMAs per wave):
//   LDS holds TWO chunk images {weights [K][4][CO_T] | x tile [4][T_T+16]}.  While the MFMAs of
//   chunk c run out of image c&1,
//     * the weight slice of chunk c+1 streams global -> LDS directly (global_load_lds_dwordx4:
//       no VGPRs, no ds_write; one 1 KB wave-instruction every few steps),
//     * the x tile of chunk c+1 (3 floats per thread, loaded one chunk earlier) is written with
//       its zero padding applied by an AND mask, and the x tile of chunk c+2 is loaded,
//   and ONE barrier (with the vmcnt(0) the compiler attaches to it) closes the chunk.
//   Every global offset is loop-invariant per thread and precomputed; a chunk only advances a
//   uniform base pointer.
// Loads are UNCONDITIONAL (clamped addresses, zeroing by mask): a load that is only used under
// a condition gets sunk into a branch by hipcc and followed by s_waitcnt vmcnt(0).
// Epilogue modes: EPI_PLAIN stores y; EPI_STATS also emits the train-mode BN (sum, sum^2)
// partials; EPI_EVAL folds the eval-mode BatchNorm affine, ReLU and MaxPool1d(2) into the store —
// the inference path writes only the pooled activation (one launch per ConvBlock); EPI_EVAL_GAP also folds
// the global average pool behind it (last block, whole row inside one t tile): only g [N][C_out] is written.
enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_EVAL = 2, EPI_EVAL_GAP = 3 };
#ifndef ECG_FWD_FL
#define ECG_FWD_FL 1         // two-level accumulation (below); tools build -DECG_FWD_FL=0 to A/B its cost and effect
#endif

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2),
// so neighbouring block ids — which here would be the tiles that read the SAME input panel — land on
// eight different L2s and each fetches the panel again.  This bijective remap gives every XCD a
// contiguous chunk of the logical tile order instead (cdna_hip_programming.md T1, any grid size):
// tiles that share a panel sit next to each other in the chunk, are dispatched back to back and hit
// in their XCD's L2.  Placement is a speed matter only; nothing depends on it for correctness.
__device__ __forceinline__ int xcd_chunked(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
struct EvalEpi { const float *gamma, *beta, *mean, *var; float eps; int gap; };

template <int CO_T, int T_T, int WCO, int WT, int EPI>
__global__ __launch_bounds__(256, 2) void conv1d_mfma_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias,
    float *__restrict__ y, float *__restrict__ partials, int Cin, int Cout, int L, int ldx, int Lo,
    int pad, int P, int tiles_t, EvalEpi ev) {
    constexpr bool STATS = (EPI == EPI_STATS), GAP = (EPI == EPI_EVAL_GAP);
    static_assert(WCO * WT == 4, "4 waves per workgroup");
    constexpr int KK = kKM, CI_C = 4, NST = KK * CI_C / 2;
    constexpr int MC = CO_T / WCO / 32, MT = T_T / WT / 32;
    static_assert(MC >= 1 && MT >= 1, "wave tile must hold at least one 32x32 accumulator");
    constexpr int XS = T_T + 16;                 // x-tile row stride (span T_T + 14)
    constexpr int WSZ = KK * CI_C * CO_T;        // floats of one weight chunk [K][CI_C][CO_T]
    constexpr int NDMA = (WSZ + 255) / 256;      // 1 KB wave-instructions per chunk
    constexpr int WPAD = NDMA * 256;             // weight region padded to whole DMA pieces
    constexpr int DPW = (NDMA + 3) / 4;          // DMA instructions per wave per chunk
    constexpr int XEL = CI_C * XS;
    constexpr int XLOADS = (XEL + 255) / 256;
    constexpr int IMG = WPAD + XEL;
    constexpr int REDF = (STATS || GAP) ? 4 * (CO_T / WCO) * 2 : 0;
    static_assert(REDF <= IMG, "stat scratch aliases image 0");
    static_assert(NST >= 2 + DPW + XLOADS, "not enough steps to spread the staging over");

    // ONE shared array: a second __shared__ object beside an LDS-DMA target makes hipcc wait
    // vmcnt(0) in front of every LDS read.
    __shared__ __attribute__((aligned(1024))) float lds[2 * IMG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    ECG_STAMP_AT(0);
    // logical tile order: the C_out tiles of one (n, t tile) are adjacent — they read the same x panel
    const int CT = Cout / CO_T;
    const int tile = xcd_chunked(blockIdx.x, gridDim.x);
    const int tile_co = tile % CT, tile_nt = tile / CT;
    const int tile_t = tile_nt % tiles_t, n = tile_nt / tiles_t;
    const int t0 = tile_t * T_T, co0 = tile_co * CO_T;
    const int wco = (wave / WT) * (CO_T / WCO), wt = (wave % WT) * (T_T / WT);
    const float *xn = x + (size_t)n * Cin * ldx;     // ldx >= L: row stride of the input tensor

    // TWO-LEVEL ACCUMULATION (ECG_FWD_FL): v_mfma_f32_32x32x2_f32 is one k-ordered fp32 fma chain, C_in * 15 terms long
    // (up to 3 840 in the block-3 input gradient): its rounding error grew with the square root of that — 2.8x (forward)
    // and 3.9x (input gradient) what oneDNN's blocked sums leave on the block-3 shapes (tools/wgrad_error.py), and with
    // it the number of ReLU / pooling decisions that differ from an exact forward pass.  The first MFMA of every chunk
    // therefore starts from the inline constant 0 and the chunk's 60-term sum joins a second register set at the end of
    // the chunk: chains of 60 + C_in / 4 terms, same fixed order for every launch.
    constexpr bool FL = (ECG_FWD_FL != 0);
    f32x16 acc[MC][MT], acc2[MC][MT];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) { acc[a][b] = zero16; acc2[a][b] = zero16; }

    // ---- per-channel epilogue parameters, lane-indexed: lane (r + 32*half), r < 16, holds those of accumulator
    // row (r, half) of each 32-channel group.  They are loaded HERE, before the main loop: vmcnt is in-order, so
    // a global load issued between the epilogue's stores has to wait for the round trip of every store before it
    // (16 rows x ~1 us per workgroup when the bias was fetched row by row).
    constexpr bool EVALM = (EPI == EPI_EVAL || EPI == EPI_EVAL_GAP);
    float p_b[MC], p_mu[MC], p_sc[MC], p_be[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        const int ch = co0 + wco + 32 * i + acc_row(l31 & 15, half);
        p_b[i] = bias ? bias[ch] : 0.f;
        p_mu[i] = p_sc[i] = p_be[i] = 0.f;
        if (EVALM) {
            const float is = (float)(1.0 / sqrt((double)ev.var[ch] + (double)ev.eps));
            p_sc[i] = is * ev.gamma[ch]; p_mu[i] = ev.mean[ch]; p_be[i] = ev.beta[ch];
        }
    }

    // ---- loop-invariant per-thread staging offsets -------------------------------------------
    int woff[DPW];          // weight piece j of this wave: element offset inside wp (chunk 0)
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int e = min(((j * 4 + wave) * 64 + lane) * 4, WSZ - 4);   // float index in the image
        const int row = e / CO_T, col = e - row * CO_T;                  // row = k*CI_C + ci
        const int k = row / CI_C, ci = row - k * CI_C;
        woff[j] = (k * Cin + ci) * Cout + col;
    }
    int xoff[XLOADS];
    unsigned xmask = 0;
#pragma unroll
    for (int j = 0; j < XLOADS; ++j) {
        const int e = min(tid + 256 * j, XEL - 1);
        const int ci = e / XS, pos = e - ci * XS;
        const int s = t0 - pad + pos;
        xoff[j] = ci * ldx + min(max(s, 0), L - 1);
        xmask |= ((s >= 0) && (s < L)) ? (1u << j) : 0u;
    }
    float xreg[XLOADS];

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_w = [&](int j, int ci0, float *img) {       // one 1 KB piece of the weight slice
        if ((j * 4 + wave_u) < NDMA)
            glds16(wp + (size_t)ci0 * Cout + co0, (unsigned)woff[j] * 4u,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)((img - lds) + (j * 4 + wave_u) * 256) * 4u)));
    };
    auto load_x = [&](int j, int ci0) { xreg[j] = xn[(size_t)ci0 * ldx + xoff[j]]; };
    auto commit_x = [&](int j, float *img) {
        const int e = tid + 256 * j;
        const unsigned keep = 0u - ((xmask >> j) & 1u);
        if (256 * (j + 1) <= XEL || e < XEL)
            img[WPAD + e] = __uint_as_float(__float_as_uint(xreg[j]) & keep);
    };

    const int nchunks = Cin / CI_C;
    // prologue: chunk 0 -> image 0; x tile of chunk 1 -> registers
#pragma unroll
    for (int j = 0; j < DPW; ++j) dma_w(j, 0, lds);
#pragma unroll
    for (int j = 0; j < XLOADS; ++j) load_x(j, 0);
#pragma unroll
    for (int j = 0; j < XLOADS; ++j) commit_x(j, lds);
    if (nchunks > 1) {
#pragma unroll
        for (int j = 0; j < XLOADS; ++j) load_x(j, CI_C);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the asm DMA pieces are not in hipcc's books)
    __syncthreads();
    ECG_STAMP_AT(1);

    for (int c = 0; c < nchunks; ++c) {
        const float *ws = lds + (c & 1) * IMG, *xs = ws + WPAD;
        float *nxt = lds + ((c + 1) & 1) * IMG;
        const bool do_next = c + 1 < nchunks, do_next2 = c + 2 < nchunks;
        const int ci_next = (c + 1) * CI_C, ci_next2 = (c + 2) * CI_C;

        // One reduction step = one (tap, ci-pair): MC + MT LDS reads feed MC*MT MFMAs; the
        // fragments of step s+1 are read BEFORE the MFMAs of step s are issued.
        auto ld = [&](int st, float *a, float *b) {
            const int k = st / (CI_C / 2), cp = st % (CI_C / 2);
            const float *wrow = ws + ((k * CI_C + 2 * cp + half) * CO_T + wco + l31);
            const float *xrow = xs + (2 * cp + half) * XS + wt + l31 + k;
#pragma unroll
            for (int i = 0; i < MC; ++i) a[i] = wrow[32 * i];
#pragma unroll
            for (int i = 0; i < MT; ++i) b[i] = xrow[32 * i];
        };
        float a_c[MC], b_c[MT], a_n[MC], b_n[MT];
        ld(0, a_c, b_c);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            ld(st + 1 < NST ? st + 1 : 0, a_n, b_n);
            wait_lgkm_f((MC + 1) / 2 + (MT + 1) / 2);      // the reads just issued (pairs merge into ds_read2_b32) may fly
            // staging, one operation per step: x commits, then weight DMA pieces, then x loads
            if (st < XLOADS) {
                if (do_next) commit_x(st, nxt);
            } else if (st < XLOADS + DPW) {
                if (do_next) dma_w(st - XLOADS, ci_next, nxt);
            } else if (st < 2 * XLOADS + DPW) {
                if (do_next2) load_x(st - XLOADS - DPW, ci_next2);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep reads + staging ABOVE this step's MFMAs
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = mfma32(a_c[i], b_c[j], (FL && st == 0) ? zero16 : acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MC; ++i) a_c[i] = a_n[i];
#pragma unroll
            for (int i = 0; i < MT; ++i) b_c[i] = b_n[i];
        }
        if (FL) {
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc2[i][j] += acc[i][j];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // image c&1 free again; image (c+1)&1 complete (vmcnt(0) + barrier)
        if (c == 0) ECG_STAMP_AT(2);
    }
    if (FL) {
#pragma unroll
        for (int i = 0; i < MC; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = acc2[i][j];
    }
    ECG_STAMP_AT(3);
    float *red = lds;         // all images are dead: reuse image 0 for the statistics scratch

    // ---- epilogue: bias, store, per-channel (sum, sum^2) partials --------------------------
    // Nothing is in flight here (the last chunk's barrier drained vmcnt), but the compiler cannot prove it for the
    // staging registers of a loop it thinks may run zero times: without this explicit (free) wait it protects their
    // reuse below with `s_waitcnt vmcnt(2..0)` in the MIDDLE of the stores, i.e. a wait for the stores themselves.
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0), expcnt/lgkmcnt untouched
    // No global load and no LDS-crossbar permute between the stores.  Row r of the two halves: parameters come out
    // of the lane-indexed registers by v_readlane; per-row sums are reduced over each 16-lane DPP row only and
    // accumulated in lanes r / r+16 of ONE register per 32 channels, the two 16-lane sums meet once at the end.
    auto pick = [&](float v, int r) {
        const int vi = __float_as_int(v);
        const float lo = __int_as_float(__builtin_amdgcn_readlane(vi, r));
        const float hi = __int_as_float(__builtin_amdgcn_readlane(vi, r + 32));
        return half ? hi : lo;
    };
    float st_s[MC], st_q[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
    const int Lp = Lo >> 1;
    // element offset of (row 0 of this lane's half, first column of this lane) inside the output of sample n
    float *yw = EVALM ? y + ((size_t)n * Cout + co0 + wco + 4 * half) * Lp + ((t0 + wt + l31) >> 1)
                      : y + ((size_t)n * Cout + co0 + wco + 4 * half) * Lo + t0 + wt + l31;
#pragma unroll
    for (int i = 0; i < MC; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowk = 32 * i + (r & 3) + 8 * (r >> 2);          // channel row inside the wave tile, minus 4*half
            const float bv = pick(p_b[i], r);
            float s = 0.f, q = 0.f;
            if (EVALM) {
                // p[j] = max(0, max(a[2j], a[2j+1])), a = (v - mean) * (invstd * gamma) + beta.  The two
                // samples of a pooling pair sit on adjacent lanes: one DPP quad_perm fetches the partner.
                const float mu = pick(p_mu[i], r), sc = pick(p_sc[i], r), be = pick(p_be[i], r);
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int t = t0 + wt + 32 * j + l31;
                    const float a = bn_apply1(acc[i][j][r] + bv, mu, sc, be);
                    const float o = dpp_move<0xB1>(a);                 // lane ^ 1
                    const float m = fmaxf(fmaxf(a, o), 0.f);
                    const bool owner = !(l31 & 1) && (t >> 1) < Lp;
                    if (GAP) s += owner ? m : 0.f;
                    else if (owner) yw[rowk * Lp + 16 * j] = m;
                }
            } else {
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const float v = acc[i][j][r] + bv;
                    if (t0 + wt + 32 * j + l31 < Lo) {
                        yw[rowk * Lo + 32 * j] = v;
                        if (STATS) { s += v; q = __fmaf_rn(v, v, q); }
                    }
                }
            }
            if (STATS || GAP) {
                const bool mine = (l31 & 15) == r;       // lanes r and r+16 of each half keep row (r, half)
                s = row16_sum(s);                        // 4 DPP adds: every lane holds the sum of its 16-lane row
                st_s[i] += mine ? s : 0.f;
                if (STATS) { q = row16_sum(q); st_q[i] += mine ? q : 0.f; }
            }
        }
    }
    if (STATS || GAP) {
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const float s = st_s[i] + __shfl_xor(st_s[i], 16, 64);
            const float q = st_q[i] + __shfl_xor(st_q[i], 16, 64);
            if (l31 < 16) {
                const int lc = 32 * i + acc_row(l31, half);   // channel inside the wave tile
                red[(wave * (CO_T / WCO) + lc) * 2] = s;
                red[(wave * (CO_T / WCO) + lc) * 2 + 1] = q;
            }
        }
    }
    if (GAP) {
        __syncthreads();
        // global average pool: combine the WT waves of each channel row, divide by the pooled length
        for (int col = tid; col < CO_T; col += 256) {
            const int wrow = col / (CO_T / WCO), lc = col - wrow * (CO_T / WCO);
            float g = 0.f;
#pragma unroll
            for (int j = 0; j < WT; ++j) g += red[((wrow * WT + j) * (CO_T / WCO) + lc) * 2];
            y[(size_t)n * Cout + co0 + col] = g / (float)(Lo >> 1);
        }
    }
    if (STATS) {
        __syncthreads();
        // combine the WT waves that share each channel row; one (sum, sum^2) pair per channel
        for (int e = tid; e < CO_T * 2; e += 256) {
            const int col = e >> 1, w = e & 1;
            const int wrow = col / (CO_T / WCO), lc = col - wrow * (CO_T / WCO);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < WT; ++j) s += red[((wrow * WT + j) * (CO_T / WCO) + lc) * 2 + w];
            const int pidx = n * tiles_t + tile_t;
            partials[((size_t)(co0 + col) * P + pidx) * 2 + w] = s;
        }
    }
#ifdef ECG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stores of the epilogue have left the wave
#endif
    ECG_STAMP_AT(4);
}

// ---------------------------------------------------------------------------------------
// Forward / input gradient by a two-phase fast-FIR split (all four epilogues of the kernel above).
// The kernels above are bound by the fp32 matrix pipe (0.76-0.84 of its nominal peak at the clocks the chip holds), so the
// only way down is fewer MFMAs.  With xt[p] = x[t0 - pad + p] and output column m of the tile <-> outputs t0 + 2m, t0 + 2m + 1:
//     A[m] = sum_{j=0..7} w[2j]             * xt[2m + 2j]
//     B[m] = sum_{j=0..6} w[2j+1]           * xt[2m + 2j + 1]
//     D[m] = sum_{j=0..7} (w[2j] - w[2j-1]) * (xt[2m + 2j] - xt[2m + 2j + 1])          (w[-1] = 0)
//     y[t0 + 2m] = A[m] + B[m],     y[t0 + 2m + 1] = A[m] + B[m + 1] - D[m]
// i.e. 23 multiplies per output pair instead of 30: three 32 x 32 accumulators (A, B, D) over M_T = T/2 columns take 46 MFMAs
// per chunk of four input channels where the direct form takes 60.  The third product on DIFFERENCES, not on the textbook's sums
// ((w[2j] + w[2j-1]) (x.. + x..), y_odd = C - A - B'; ECG_FFA_MINUS = 0 builds it for A/B): neighbouring samples of like sign and
// magnitude — pooled ReLU outputs — subtract exactly and D is small, where C is twice the size of A and every odd output a
// difference of large sums (trajectory drift against float64 1.62x the CPU fp32 path's with C, 0.41x with D: EXPERIMENTS I5).
// Nothing is pre-processed: the LDS images are the ones of
// the kernel above ({weights [K][4][CO_T] | x tile [4][2 M_T + 16]}); one ds_read_b64 per lane yields (xt[2m+2j], xt[2m+2j+1])
// for the A, B and D step of tap pair j (conflict-free: 32 lanes x 8 bytes), the two differences are one VALU operation each.
// B[m + 1] of the tile's last column belongs to the next tile, so a tile of M_T columns yields 2 M_T - 2 outputs: tiles are
// TS = 2 M_T - 2 apart (column M_T - 1 only supplies B); for the model's row lengths that is the same number of tiles as
// 2 M_T-wide ones.  In the epilogue B goes through LDS once (the images are dead) to come back shifted by a column.
// Rounding: the two extra subtractions per product and the final combination are fp32; the result differs from the direct form
// by a few ulp of the accumulated magnitude (tests state the bounds: test_conv_fast_fir_error_by_signal_class).  Inference epilogues: the pooling pair (2m, 2m + 1) sits in ONE
// lane, so BatchNorm + ReLU + MaxPool(2) is three VALU operations per pair and the pooled row leaves as contiguous dwords.
#ifndef ECG_FFA_MINB
#define ECG_FFA_MINB 2       // workgroups per CU the register allocation aims at (4: 128 registers, spills; measured slower)
#endif
#ifndef ECG_FFA_MINUS
#define ECG_FFA_MINUS 1      // third product on differences (0: the textbook sum form, A/B only — fails the trajectory bars)
#endif
#ifndef ECG_FFA_FLP2
#define ECG_FFA_FLP2 64      // one second-level add per TWO chunks where C_in % 8 == 0 and C_in >= this (0: per chunk everywhere)
#endif
template <int CO_T, int M_T, int WCO, int WT, int EPI, int CI_C = 4, int FLP = 1>
__global__ __launch_bounds__(256, ECG_FFA_MINB) void conv1d_mfma_ffa_kernel(
    const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias,
    float *__restrict__ y, float *__restrict__ partials, int Cin, int Cout, int L, int ldx, int Lo,
    int pad, int P, int tiles_t, EvalEpi ev) {
    constexpr bool STATS = (EPI == EPI_STATS), GAP = (EPI == EPI_EVAL_GAP);
    constexpr bool EVALM = (EPI == EPI_EVAL || EPI == EPI_EVAL_GAP);
    static_assert(WCO * WT == 4, "4 waves per workgroup");
    static_assert(kKM == 15, "tap split 8 + 7");
    static_assert(CO_T / WCO == 32 && M_T / WT == 32, "one 32 x 32 accumulator per product and wave");
    constexpr int KK = kKM, NJ = (KK + 1) / 2, NST = NJ * (CI_C / 2);
    constexpr int T_T = 2 * M_T, TS = T_T - 2;
    constexpr int XS = T_T + 16;                 // x-tile row stride (span 2 (M_T - 1) + 16), even: the b64 reads stay aligned
    constexpr int WSZ = KK * CI_C * CO_T;
    constexpr int NDMA = (WSZ + 255) / 256;
    constexpr int WPAD = NDMA * 256;
    constexpr int DPW = (NDMA + 3) / 4;
    constexpr int XEL = CI_C * XS;
    constexpr int XLOADS = (XEL + 255) / 256;
    constexpr int IMG = WPAD + XEL;
    constexpr int BXS = M_T + 1;                 // row stride of the B exchange (odd: conflict-free column walks)
    constexpr int REDF = (STATS || GAP) ? 4 * 32 * 2 : 0;
    static_assert(CO_T * BXS + REDF <= 2 * IMG, "B exchange + stat scratch alias the dead images");
    static_assert(NST >= 2 * XLOADS + DPW, "not enough steps to spread the staging over");

    __shared__ __attribute__((aligned(1024))) float lds[2 * IMG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    ECG_STAMP_AT(0);
    const int CT = Cout / CO_T;
    const int tile = xcd_chunked(blockIdx.x, gridDim.x);
    const int tile_co = tile % CT, tile_nt = tile / CT;
    const int tile_t = tile_nt % tiles_t, n = tile_nt / tiles_t;
    const int t0 = tile_t * TS, co0 = tile_co * CO_T;
    const int wco = (wave / WT) * 32, wm = (wave % WT) * 32;
    const float *xn = x + (size_t)n * Cin * ldx;

    constexpr bool FL = (ECG_FWD_FL != 0);
    f32x16 acc[3], acc2[3];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { acc[a] = zero16; acc2[a] = zero16; }

    const int pch = co0 + wco + acc_row(l31 & 15, half);      // lane-indexed epilogue parameters (see the kernel above)
    const float p_b = bias ? bias[pch] : 0.f;
    float p_mu = 0.f, p_sc = 0.f, p_be = 0.f;
    if (EVALM) {
        const float is = (float)(1.0 / sqrt((double)ev.var[pch] + (double)ev.eps));
        p_sc = is * ev.gamma[pch]; p_mu = ev.mean[pch]; p_be = ev.beta[pch];
    }

    int woff[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int e = min(((j * 4 + wave) * 64 + lane) * 4, WSZ - 4);
        const int row = e / CO_T, col = e - row * CO_T;
        const int k = row / CI_C, ci = row - k * CI_C;
        woff[j] = (k * Cin + ci) * Cout + col;
    }
    int xoff[XLOADS];
    unsigned xmask = 0;
#pragma unroll
    for (int j = 0; j < XLOADS; ++j) {
        const int e = min(tid + 256 * j, XEL - 1);
        const int ci = e / XS, pos = e - ci * XS;
        const int s = t0 - pad + pos;
        xoff[j] = ci * ldx + min(max(s, 0), L - 1);
        xmask |= ((s >= 0) && (s < L)) ? (1u << j) : 0u;
    }
    float xreg[XLOADS];

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_w = [&](int j, int ci0, float *img) {
        if ((j * 4 + wave_u) < NDMA)
            glds16(wp + (size_t)ci0 * Cout + co0, (unsigned)woff[j] * 4u,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)((img - lds) + (j * 4 + wave_u) * 256) * 4u)));
    };
    auto load_x = [&](int j, int ci0) { xreg[j] = xn[(size_t)ci0 * ldx + xoff[j]]; };
    auto commit_x = [&](int j, float *img) {
        const int e = tid + 256 * j;
        const unsigned keep = 0u - ((xmask >> j) & 1u);
        if (256 * (j + 1) <= XEL || e < XEL)
            img[WPAD + e] = __uint_as_float(__float_as_uint(xreg[j]) & keep);
    };

    const int nchunks = Cin / CI_C;
#pragma unroll
    for (int j = 0; j < DPW; ++j) dma_w(j, 0, lds);
#pragma unroll
    for (int j = 0; j < XLOADS; ++j) load_x(j, 0);
#pragma unroll
    for (int j = 0; j < XLOADS; ++j) commit_x(j, lds);
    if (nchunks > 1) {
#pragma unroll
        for (int j = 0; j < XLOADS; ++j) load_x(j, CI_C);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ECG_STAMP_AT(1);

    typedef float f32x2t __attribute__((ext_vector_type(2)));
    // FLP = 2: a first-level sum runs over TWO chunks (OPEN starts it from zero, CLOSE adds it to the second level)
    auto chunk = [&](int c, auto open_c, auto close_c) {
        constexpr bool OPEN = decltype(open_c)::value, CLOSE = decltype(close_c)::value;
        const float *ws = lds + (c & 1) * IMG, *xs = ws + WPAD;
        float *nxt = lds + ((c + 1) & 1) * IMG;
        const bool do_next = c + 1 < nchunks, do_next2 = c + 2 < nchunks;
        const int ci_next = (c + 1) * CI_C, ci_next2 = (c + 2) * CI_C;

        // one step = tap pair j of channel pair cp: w[2j], w[2j+1] (one dword each) and (xt[2m+2j], xt[2m+2j+1]) (one b64)
        // feed the A, B and C MFMAs; the reads of step s + 1 are issued before the MFMAs of step s
        auto ld = [&](int st, float &wa, float &wb, f32x2t &xq) {
            const int j = st / (CI_C / 2), cp = st % (CI_C / 2);
            const float *wrow = ws + ((2 * j * CI_C + 2 * cp + half) * CO_T + wco + l31);
            wa = wrow[0];
            wb = (2 * j + 1 < KK) ? wrow[CI_C * CO_T] : 0.f;
            xq = *reinterpret_cast<const f32x2t *>(xs + (2 * cp + half) * XS + 2 * (wm + l31) + 2 * j);
        };
        float wa_c, wb_c, wa_n, wb_n, wprev[CI_C / 2];
        f32x2t xq_c, xq_n;
#pragma unroll
        for (int i = 0; i < CI_C / 2; ++i) wprev[i] = 0.f;
        ld(0, wa_c, wb_c, xq_c);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int j = st / (CI_C / 2), cp = st % (CI_C / 2);
            ld(st + 1 < NST ? st + 1 : 0, wa_n, wb_n, xq_n);
            wait_lgkm_f(3);
            if (st < XLOADS) {
                if (do_next) commit_x(st, nxt);
            } else if (st < XLOADS + DPW) {
                if (do_next) dma_w(st - XLOADS, ci_next, nxt);
            } else if (st < 2 * XLOADS + DPW) {
                if (do_next2) load_x(st - XLOADS - DPW, ci_next2);
            }
            const float wc = ECG_FFA_MINUS ? wa_c - wprev[cp] : wa_c + wprev[cp];
            const float xc = ECG_FFA_MINUS ? xq_c[0] - xq_c[1] : xq_c[0] + xq_c[1];
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = mfma32(wa_c, xq_c[0], (FL && OPEN && j == 0 && cp == 0) ? zero16 : acc[0]);
            if (2 * j + 1 < KK) acc[1] = mfma32(wb_c, xq_c[1], (FL && OPEN && j == 0 && cp == 0) ? zero16 : acc[1]);
            acc[2] = mfma32(wc, xc, (FL && OPEN && j == 0 && cp == 0) ? zero16 : acc[2]);
            __builtin_amdgcn_sched_barrier(0);
            wprev[cp] = wb_c;
            wa_c = wa_n; wb_c = wb_n; xq_c = xq_n;
        }
        if (FL && CLOSE) {
#pragma unroll
            for (int a = 0; a < 3; ++a) acc2[a] += acc[a];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c == 0) ECG_STAMP_AT(2);
    };
    {
        const std::true_type yes{};
        const std::false_type no{};
        int c = 0;
        if (FLP == 2) {
            for (; c + 1 < nchunks; c += 2) { chunk(c, yes, no); chunk(c + 1, no, yes); }
        }
        for (; c < nchunks; ++c) chunk(c, yes, yes);
    }
    if (FL) {
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] = acc2[a];
    }
    ECG_STAMP_AT(3);

    // ---- epilogue -------------------------------------------------------------------------------------
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) (see the kernel above)
    // B one column to the left: through LDS (the last column of a wave's block comes from the wave beside it)
    float *bx = lds, *red = lds + CO_T * BXS;
#pragma unroll
    for (int r = 0; r < 16; ++r) bx[(wco + acc_row(r, half)) * BXS + wm + l31] = acc[1][r];
    __syncthreads();
    f32x16 bn;
#pragma unroll
    for (int r = 0; r < 16; ++r) bn[r] = bx[(wco + acc_row(r, half)) * BXS + min(wm + l31 + 1, M_T - 1)];
    auto pick = [&](float v, int r) {
        const int vi = __float_as_int(v);
        const float lo = __int_as_float(__builtin_amdgcn_readlane(vi, r));
        const float hi = __int_as_float(__builtin_amdgcn_readlane(vi, r + 32));
        return half ? hi : lo;
    };
    float st_s = 0.f, st_q = 0.f;
    const int m = wm + l31, t = t0 + 2 * m;
    const bool ok0 = m < M_T - 1 && t < Lo, ok1 = m < M_T - 1 && t + 1 < Lo;
    const int Lp = Lo >> 1, pm = (t0 >> 1) + m;               // (t0 is even: the pair (t, t + 1) is pooled output pm)
    const bool okp = m < M_T - 1 && pm < Lp;
    float *yw = EVALM ? y + ((size_t)n * Cout + co0 + wco + 4 * half) * Lp + pm
                      : y + ((size_t)n * Cout + co0 + wco + 4 * half) * Lo + t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rowk = (r & 3) + 8 * (r >> 2);
        const float bv = pick(p_b, r);
        const float v0 = (acc[0][r] + acc[1][r]) + bv;
        const float v1 = (ECG_FFA_MINUS ? (acc[0][r] + bn[r]) - acc[2][r] : (acc[2][r] - acc[0][r]) - bn[r]) + bv;
        float s = 0.f, q = 0.f;
        if (EVALM) {
            const float mu = pick(p_mu, r), sc = pick(p_sc, r), be = pick(p_be, r);
            const float mx = fmaxf(fmaxf(bn_apply1(v0, mu, sc, be), bn_apply1(v1, mu, sc, be)), 0.f);
            if (GAP) {
                s = row16_sum(okp ? mx : 0.f);
                st_s += ((l31 & 15) == r) ? s : 0.f;
            } else if (okp) yw[rowk * Lp] = mx;
            continue;
        }
        // (the pair is 4-byte aligned only: odd row lengths; gfx950 global stores do not need more)
        typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
        if (ok1) { f32x2u v; v[0] = v0; v[1] = v1; *reinterpret_cast<f32x2u *>(yw + rowk * Lo) = v; }
        else if (ok0) yw[rowk * Lo] = v0;
        if (STATS) {
            if (ok0) { s += v0; q = __fmaf_rn(v0, v0, q); }
            if (ok1) { s += v1; q = __fmaf_rn(v1, v1, q); }
        }
        if (STATS) {
            const bool mine = (l31 & 15) == r;
            s = row16_sum(s);
            st_s += mine ? s : 0.f;
            q = row16_sum(q);
            st_q += mine ? q : 0.f;
        }
    }
    if (STATS || GAP) {
        const float s = st_s + __shfl_xor(st_s, 16, 64);
        const float q = st_q + __shfl_xor(st_q, 16, 64);
        if (l31 < 16) {
            const int lc = acc_row(l31, half);
            red[(wave * 32 + lc) * 2] = s;
            red[(wave * 32 + lc) * 2 + 1] = q;
        }
        __syncthreads();
    }
    if (GAP) {       // global average pool: combine the WT waves of each channel row, divide by the pooled length
        for (int col = tid; col < CO_T; col += 256) {
            const int wrow = col / 32, lc = col - wrow * 32;
            float g = 0.f;
#pragma unroll
            for (int j = 0; j < WT; ++j) g += red[((wrow * WT + j) * 32 + lc) * 2];
            y[(size_t)n * Cout + co0 + col] = g / (float)Lp;
        }
    }
    if (STATS) {
        for (int e = tid; e < CO_T * 2; e += 256) {
            const int col = e >> 1, w = e & 1;
            const int wrow = col / 32, lc = col - wrow * 32;
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < WT; ++j) s2 += red[((wrow * WT + j) * 32 + lc) * 2 + w];
            const int pidx = n * tiles_t + tile_t;
            partials[((size_t)(co0 + col) * P + pidx) * 2 + w] = s2;
        }
    }
#ifdef ECG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stores of the epilogue have left the wave
#endif
    ECG_STAMP_AT(4);
}

struct FwdCfg { int co_t, t_t, stride; };     // stride: distance of the t tiles (fast-FIR kernel: t_t - 2)
#ifndef ECG_FWD_FFA
#define ECG_FWD_FFA 7       // bit 0: forward (statistics epilogue), bit 1: plain epilogue (input gradient, unfused forward),
#endif                      // bit 2: the inference epilogues

// Tile choice of the one-tile-per-workgroup kernel (inference epilogues).  Measured on MI355X (B=256): 64x128
// tiles at 4 resident workgroups per CU beat 128x128 at 2 per CU by 6-8 % (more independent waves per SIMD to
// cover each other's prologue, epilogue and staging waits), so the 64-channel tile is used whenever C_out allows it.
static FwdCfg fwd_cfg(int N, int Cout, int Lo, bool train) {
    (void)N; (void)Lo;
    const bool ffa = train ? (ECG_FWD_FFA & 1) != 0 : (ECG_FWD_FFA & 4) != 0;   // (callers: statistics partials | one-tile GAP)
    if (Cout % 64 == 0) return {64, 128, ffa ? 126 : 128};
    return {32, 256, ffa ? 254 : 256};
}

bool mfma_fwd_supported(int Cin, int Cout, int K, int pad) {
    (void)pad;
    return K == kKM && Cin % 4 == 0 && Cout % 32 == 0;   // other shapes take the direct path
}

int mfma_fwd_stat_partials(int N, int Cin, int Cout, int Lo) {
    (void)Cin;
    return N * cdiv(Lo, fwd_cfg(N, Cout, Lo, true).stride);
}

template <int CO_T, int T_T, int WCO, int WT>
static void launch_fwd(const float *x, const float *wp, const float *bias, float *y,
                       float *partials, const EvalEpi *ev, int N, int Cin, int Cout, int L, int ldx,
                       int Lo, int pad, hipStream_t st) {
    if (ev ? (ECG_FWD_FFA & 4) != 0 : (((ECG_FWD_FFA & 1) && partials) || ((ECG_FWD_FFA & 2) && !partials))) {
        const int tiles_t = cdiv(Lo, T_T - 2);
        dim3 grid((unsigned)((size_t)tiles_t * (Cout / CO_T) * N)), block(256);
        const int P = N * tiles_t;
        const EvalEpi none{nullptr, nullptr, nullptr, nullptr, 0.f, 0};
        // One second-level add per TWO four-channel chunks where the reduction is long (C_in % 8 == 0, C_in >= 64): first-level chains
        // of 64 instead of 32 terms, half as many second-level adds — the add (three accumulator sets behind the last MFMAs of a
        // chunk) is what a chunk boundary costs, not its barrier.  Same box, forward + input gradient of the four blocks against
        // eight-channel chunks with one add each (70 KB of LDS: two workgroups per CU; this form keeps three): 845.6 -> 832.8 us at
        // 12x1000, 3 737.6 -> 3 643.3 at 12x5000.  NOT on the 32-channel reduction of block 1's forward (another -9 / -31 us): its y
        // error against float64 goes 0.56 -> 0.69 of the CPU fp32 path's and the trajectory test fails its bars (drift 1.28x /
        // 2.76x the CPU's at B = 32 / 256; with the threshold at 64: 0.41x / 0.39x) — EXPERIMENTS I5.
#define ECG_FFA(MODE, EV) do { \
        if (ECG_FFA_FLP2 && Cin % 8 == 0 && Cin >= ECG_FFA_FLP2) \
            hipLaunchKernelGGL((conv1d_mfma_ffa_kernel<CO_T, T_T / 2, WCO, WT, MODE, 4, 2>), grid, block, 0, st, x, wp, \
                               bias, y, partials, Cin, Cout, L, ldx, Lo, pad, P, tiles_t, EV); \
        else \
            hipLaunchKernelGGL((conv1d_mfma_ffa_kernel<CO_T, T_T / 2, WCO, WT, MODE>), grid, block, 0, st, x, wp, bias, y, \
                               partials, Cin, Cout, L, ldx, Lo, pad, P, tiles_t, EV); } while (0)
        if (ev && ev->gap) ECG_FFA(EPI_EVAL_GAP, *ev);
        else if (ev) ECG_FFA(EPI_EVAL, *ev);
        else if (partials) ECG_FFA(EPI_STATS, none);
        else ECG_FFA(EPI_PLAIN, none);
#undef ECG_FFA
        return;
    }
    const int tiles_t = cdiv(Lo, T_T);
    dim3 grid((unsigned)((size_t)tiles_t * (Cout / CO_T) * N)), block(256);
    const int P = N * tiles_t;
    const EvalEpi none{nullptr, nullptr, nullptr, nullptr, 0.f, 0};
#define ECG_FWD(MODE, EV) \
    hipLaunchKernelGGL((conv1d_mfma_fwd_kernel<CO_T, T_T, WCO, WT, MODE>), grid, block, 0, st, x, wp, \
                       bias, y, partials, Cin, Cout, L, ldx, Lo, pad, P, tiles_t, EV)
    if (ev && ev->gap) ECG_FWD(EPI_EVAL_GAP, *ev);
    else if (ev) ECG_FWD(EPI_EVAL, *ev);
    else if (partials) ECG_FWD(EPI_STATS, none);
    else ECG_FWD(EPI_PLAIN, none);
#undef ECG_FWD
}

static int mfma_fwd_any(const float *x, const float *wp, const float *bias, float *y,
                        float *partials, const EvalEpi *ev, int N, int Cin, int Cout, int L, int ldx,
                        int K, int pad, hipStream_t st) {
    const int Lo = L + 2 * pad - K + 1;
    const FwdCfg c = fwd_cfg(N, Cout, Lo, ev == nullptr);
    if (c.co_t == 64)
        launch_fwd<64, 128, 2, 2>(x, wp, bias, y, partials, ev, N, Cin, Cout, L, ldx, Lo, pad, st);
    else
        launch_fwd<32, 256, 1, 4>(x, wp, bias, y, partials, ev, N, Cin, Cout, L, ldx, Lo, pad, st);
    return check_launch("conv1d_mfma_fwd_kernel");
}

// ldx >= L is the row stride of x (dgrad reads a row-padded dY through it)
int mfma_fwd(const float *x, int ldx, const float *wp, const float *bias, float *y, float *partials,
             int N, int Cin, int Cout, int L, int K, int pad, hipStream_t st) {
    return mfma_fwd_any(x, wp, bias, y, partials, nullptr, N, Cin, Cout, L, ldx, K, pad, st);
}

// eval-mode ConvBlock in one launch: p = MaxPool2(ReLU(BN_running(conv(x))))
// gap != 0: the global average pool is folded in too (p is then g [N][C_out]); needs the whole row in one
// t tile — mfma_fwd_eval_gap_supported
int mfma_fwd_eval_pool(const float *x, const float *wp, const float *bias, const float *gamma,
                       const float *beta, const float *mean, const float *var, float eps, float *p,
                       int N, int Cin, int Cout, int L, int K, int pad, hipStream_t st, int gap) {
    const EvalEpi ev{gamma, beta, mean, var, eps, gap};
    return mfma_fwd_any(x, wp, bias, p, nullptr, &ev, N, Cin, Cout, L, L, K, pad, st);
}

bool mfma_fwd_eval_gap_supported(int Cin, int Cout, int L, int K, int pad) {
    const int Lo = L + 2 * pad - K + 1;
    return mfma_fwd_supported(Cin, Cout, K, pad) && Lo >= 2 && Lo <= fwd_cfg(1, Cout, Lo, false).stride;
}

// =======================================================================================
// weight gradient
// =======================================================================================
// grid = (ceil(R/R_T), Cout/M_T, S), R = Cin*K.  4 waves laid out WM x WR x WK: WK > 1 splits the
// staged t range between waves (their accumulators are summed through LDS at the end).
// slab[s][co][r] (+ bias slab [s][co] behind the S weight slabs), summed by wgrad_reduce_kernel.
//
// Pipeline: a stage is one (n, t-tile).  LDS holds TWO stage images {dY tile | x tile}.  While the
// MFMAs of stage i run out of image i&1, the registers holding stage i+1 are written into image
// (i+1)&1 (first half of the steps) and the loads of stage i+2 are issued (second half), one
// staging operation per MFMA group; ONE barrier closes the stage.  Per-thread global offsets are
// loop-invariant; a stage only moves uniform base pointers.
// FL: two-level accumulation (see conv1d_mfma_wgrad_dma_kernel below).
template <int M_T, int R_T, int WM, int WR, int WK, int T_T, int KK, int FL = 1>
__global__ __launch_bounds__(256, 2) void conv1d_mfma_wgrad_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, float *__restrict__ slab, int N,
    int Cin, int Cout, int L, int Lo, int ldy, int pad, int S) {
    static_assert(WM * WR * WK == 4, "4 waves per workgroup");
    constexpr int MC = M_T / WM / 32, MR = R_T / WR / 32;
    constexpr int TW = T_T / WK;                        // t range of one wave per staged tile
    constexpr int NST = TW / 2;                         // reduction steps per stage
    constexpr int DS = T_T + 1;                         // dY tile row stride: odd -> conflict-free A reads
    constexpr int XSPAN = T_T + KK - 1;
    constexpr int XS = ((XSPAN - KK + 31) / 32) * 32 + KK;   // >= XSPAN and == KK (mod 32)
    constexpr int NCI = (R_T + KK - 2) / KK + 1;        // input channels a column tile can touch
    constexpr int DEL = M_T * T_T, DLOADS = DEL / 256;
    constexpr int XEL = NCI * XS, XLOADS = (XEL + 255) / 256;
    constexpr int NOPS = DLOADS + XLOADS;               // staging operations per stage (per thread)
    constexpr int IMG = M_T * DS + XEL;
    constexpr int ACCF = MC * MR * 16 * 64;             // floats of one wave's accumulators
    constexpr int LDSF = (WK > 1 && WM * WR * ACCF > 2 * IMG) ? WM * WR * ACCF : 2 * IMG;
    static_assert(XS >= XSPAN, "x row stride too small");
    static_assert(DEL % 256 == 0, "dY tile must be a whole number of 256-thread passes");
    static_assert(256 % T_T == 0 || T_T % 256 == 0, "dY tile rows per pass");
    static_assert(XLOADS <= 32 && DLOADS <= 64, "mask bits");

    __shared__ float lds[LDSF + (WK > 1 ? 4 * 32 : 0)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int R = Cin * KK;
    // logical order: the R tiles of one (C_out tile, split) are adjacent — they read the same dY slice
    const int RT = (R + R_T - 1) / R_T, CT = Cout / M_T;
    const int tile = xcd_chunked(blockIdx.x, gridDim.x);
    const int tile_r = tile % RT, tile_cs = tile / RT;
    const int r0 = tile_r * R_T, co0 = (tile_cs % CT) * M_T, s = tile_cs / CT;
    const int wk = wave % WK, wr = (wave / WK) % WR, wm = wave / (WK * WR);
    const int wm0 = wm * (M_T / WM), wr0 = wr * (R_T / WR), wt0 = wk * TW;
    const int ci_base = r0 / KK;
    const int n_begin = (int)((long long)N * s / S), n_end = (int)((long long)N * (s + 1) / S);

    // per-lane LDS offset of column r = r0 + wr0 + 32*j + l31 inside the x tile: ci_local*XS + k
    int xcol[MR];
#pragma unroll
    for (int j = 0; j < MR; ++j) {
        int r = r0 + wr0 + 32 * j + l31;
        if (r >= R) r = R - 1;                 // clamped columns compute garbage that is never stored
        const int ci = r / KK;
        xcol[j] = (ci - ci_base) * XS + (r - ci * KK);
    }

    f32x16 acc[MC][MR], acc2[MC][MR];      // acc2: the second level (FL), dead otherwise
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b < MR; ++b) { acc[a][b] = zero16; acc2[a][b] = zero16; }
    float bsum[MC], bsum2[MC];
#pragma unroll
    for (int a = 0; a < MC; ++a) bsum[a] = bsum2[a] = 0.f;
    const bool want_bias = (tile_r == 0) && (wr == 0);

    // ---- staging: loop-invariant per-thread pieces ------------------------------------------
    constexpr int RPP = (T_T >= 256) ? 1 : 256 / T_T;       // dY rows fetched per pass
    constexpr int PPR = (T_T >= 256) ? T_T / 256 : 1;       // passes per dY row
    const int drow0 = (T_T >= 256) ? 0 : tid / T_T, dtt = (T_T >= 256) ? tid : tid % T_T;
    int xci[XLOADS], xpos[XLOADS];
#pragma unroll
    for (int j = 0; j < XLOADS; ++j) {
        const int e = min(tid + 256 * j, XEL - 1);
        xci[j] = min(ci_base + e / XS, Cin - 1) * L;          // row offset (clamped into the tensor)
        xpos[j] = e % XS - pad;
    }
    float dreg[DLOADS], xreg[XLOADS];
    unsigned xmask = 0, dmask = 0;
    const int ntt = (Lo + T_T - 1) / T_T;
    const int total = (n_end - n_begin) * ntt;

    // stage geometry -> uniform bases + the few per-stage per-thread values
    const float *dyn = dy, *xn = x;
    int dvoff = 0, t0 = 0;
    auto stage_setup = [&](int it) {
        const int n = n_begin + it / ntt;
        t0 = (it % ntt) * T_T;
        dyn = dy + ((size_t)n * Cout + co0) * ldy + t0;
        xn = x + (size_t)n * Cin * L;
        dvoff = drow0 * ldy + min(dtt, Lo - 1 - t0);          // same VGPR offset for every dY load
        dmask = 0;
#pragma unroll
        for (int q = 0; q < PPR; ++q) dmask |= (t0 + dtt + 256 * q < Lo) ? (1u << q) : 0u;
    };
    auto load_op = [&](int o) {
        if (o < DLOADS) {
            // row = drow0 + (o/PPR)*RPP, tt = dtt + (o%PPR)*256: the (o-dependent) part is uniform
            const int extra = (o % PPR) * 256;
            const float *pj = dyn + (size_t)(o / PPR) * RPP * ldy;
            dreg[o] = pj[PPR == 1 ? dvoff : drow0 * ldy + min(dtt + extra, Lo - 1 - t0)];
        } else {
            const int j = o - DLOADS;
            const int sidx = t0 + xpos[j];
            xreg[j] = xn[xci[j] + min(max(sidx, 0), L - 1)];
            const unsigned bit = ((sidx >= 0) && (sidx < L)) ? (1u << j) : 0u;
            xmask = (j == 0) ? bit : (xmask | bit);
        }
    };
    // the masks of the stage being committed must survive the loads of the following stage
    unsigned cxmask = 0, cdmask = 0;
    auto commit_op = [&](int o, float *img) {
        if (o < DLOADS) {
            const int row = drow0 + (o / PPR) * RPP, tt = dtt + (o % PPR) * 256;
            const unsigned keep = 0u - ((cdmask >> (o % PPR)) & 1u);
            img[row * DS + tt] = __uint_as_float(__float_as_uint(dreg[o]) & keep);
        } else {
            const int j = o - DLOADS;
            const int e = tid + 256 * j;
            const unsigned keep = 0u - ((cxmask >> j) & 1u);
            if (256 * (j + 1) <= XEL || e < XEL)
                img[M_T * DS + e] = __uint_as_float(__float_as_uint(xreg[j]) & keep);
        }
    };

    // prologue: stage 0 -> image 0, stage 1 -> registers
    if (total > 0) {
        stage_setup(0);
#pragma unroll
        for (int o = 0; o < NOPS; ++o) load_op(o);
        cxmask = xmask; cdmask = dmask;
#pragma unroll
        for (int o = 0; o < NOPS; ++o) commit_op(o, lds);
        if (total > 1) {
            stage_setup(1);
#pragma unroll
            for (int o = 0; o < NOPS; ++o) load_op(o);
        }
    }
    __syncthreads();

    for (int it = 0; it < total; ++it) {
        const float *dys = lds + (it & 1) * IMG, *xs = dys + M_T * DS;
        float *nxt = lds + ((it + 1) & 1) * IMG;
        const bool do_commit = it + 1 < total, do_load = it + 2 < total;
        cxmask = xmask; cdmask = dmask;                    // masks of stage it+1 (now in registers)
        if (do_load) stage_setup(it + 2);                  // bases/masks for the loads issued below

        const float *arow = dys + (wm0 + l31) * DS + wt0 + half;
        const float *brow = xs + wt0 + half;
        auto ld = [&](int tp, float *a, float *b) {
#pragma unroll
            for (int i = 0; i < MC; ++i) a[i] = arow[32 * i * DS + tp];
#pragma unroll
            for (int j = 0; j < MR; ++j) b[j] = brow[xcol[j] + tp];
        };
        float a_c[MC], b_c[MR], a_n[MC], b_n[MR];
        ld(0, a_c, b_c);
        constexpr int H1 = NST / 2;
#pragma unroll
        for (int st = 0; st < NST; ++st) {      // next step's fragments are read before this step's MFMAs
            ld(st + 1 < NST ? 2 * (st + 1) : 0, a_n, b_n);
            if (st < H1) {
                if (do_commit) {
#pragma unroll
                    for (int o = st * NOPS / H1; o < (st + 1) * NOPS / H1; ++o) commit_op(o, nxt);
                }
            } else {
                if (do_load) {
#pragma unroll
                    for (int o = (st - H1) * NOPS / (NST - H1); o < (st - H1 + 1) * NOPS / (NST - H1); ++o)
                        load_op(o);
                }
            }
            __builtin_amdgcn_sched_barrier(0);         // keep reads + staging ABOVE these MFMAs
#pragma unroll
            for (int i = 0; i < MC; ++i) bsum[i] = (FL && st == 0) ? a_c[i] : bsum[i] + a_c[i];   // bias-grad rides on the A fragments
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < MR; ++j)
                    acc[i][j] = mfma32(a_c[i], b_c[j], (FL && st == 0) ? zero16 : acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MC; ++i) a_c[i] = a_n[i];
#pragma unroll
            for (int j = 0; j < MR; ++j) b_c[j] = b_n[j];
        }
        if (FL) {               // second level: the stage's sums join the running totals
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < MR; ++j) acc2[i][j] += acc[i][j];
#pragma unroll
            for (int i = 0; i < MC; ++i) bsum2[i] += bsum[i];
        }
        __syncthreads();      // image it&1 free again; image (it+1)&1 complete
    }

    if (FL) {
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            bsum[i] = bsum2[i];
#pragma unroll
            for (int j = 0; j < MR; ++j) acc[i][j] = acc2[i][j];
        }
    }
    // ---- combine the WK t-split waves through LDS (fixed order), then write the slab ---------
    if (WK > 1) {
        float *bred = lds + LDSF;      // [4][32] bias partials
        float *ex = lds + (wr + WR * wm) * ACCF;      // one exchange area per group of WK waves that share an output tile
        for (int w = 1; w < WK; ++w) {
            __syncthreads();
            if (wk == w) {
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int j = 0; j < MR; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            ex[((i * MR + j) * 16 + r) * 64 + lane] = acc[i][j][r];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int j = 0; j < MR; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[i][j][r] += ex[((i * MR + j) * 16 + r) * 64 + lane];
            }
        }
        static_assert(WK == 1 || (WM == 1 && MC == 1), "the bias exchange below assumes one 32-channel row, wk fastest in the wave index");
        {
            // WK > 1 is only instantiated with WM == 1, MC == 1: 32 channels; waves 0 .. WK-1 are (wr = 0, wk = 0 .. WK-1).
            // The barriers are unconditional: want_bias differs between the waves of a workgroup when WR > 1.
            const float b = bsum[0] + __shfl_xor(bsum[0], 32, 64);
            __syncthreads();
            if (want_bias && half == 0) bred[wave * 32 + l31] = b;
            __syncthreads();
            if (want_bias && wave == 0 && half == 0) {
                float t = bred[l31];
                for (int w = 1; w < WK; ++w) t += bred[w * 32 + l31];
                bsum[0] = t;
            }
        }
    } else if (want_bias) {
#pragma unroll
        for (int i = 0; i < MC; ++i) bsum[i] += __shfl_xor(bsum[i], 32, 64);
    }

    const size_t wslab = (size_t)Cout * R;
    if (wk == 0) {
        float *out = slab + (size_t)s * wslab;
#pragma unroll
        for (int i = 0; i < MC; ++i)
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                const int r = r0 + wr0 + 32 * j + l31;
                if (r < R) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int co = co0 + wm0 + 32 * i + acc_row(q, half);
                        out[(size_t)co * R + r] = acc[i][j][q];
                    }
                }
            }
        if (want_bias && half == 0) {
#pragma unroll
            for (int i = 0; i < MC; ++i)
                slab[(size_t)S * wslab + (size_t)s * Cout + co0 + wm0 + 32 * i + l31] = bsum[i];
        }
    }
}

// ---------------------------------------------------------------------------------------
// Weight gradient, dY streamed by LDS-DMA.  Needs a ROW-PADDED dY: row stride ldy a multiple of
// 64 floats with zeros in [Lo, ldy) (ecg_bn_relu_pool_bwd_ld writes it that way), 16-byte aligned
// base.  Then every (n, 64-wide t tile) of dY is M_T/4 whole 1 KB wave transfers straight into LDS
// (no VGPRs, no ds_write, no tail masks: the pad supplies the zeros that cancel the garbage x
// columns of a ragged last tile).
//   LDS dY image [M_T][64] floats, 16-byte chunk c of row m stored at chunk c ^ (m & 15): the
//   ds_read_b128 of the A fragments is conflict-free.  K-index assignment inside a group of four
//   reduction steps (8 consecutive t): lane half h owns t = 8g + 4h + j at step j, so ONE b128
//   read per lane feeds four MFMA steps.  The x tile (B operand) is register-staged as in the
//   kernel above and read at +4h+j.
// grid = (ceil(R/R_T), Cout/M_T, S); T_T = 64; slab layout as above.
//
// TWO-LEVEL ACCUMULATION (FL != 0).  v_mfma_f32_32x32x2_f32 is one k-ordered fp32 fma chain, so a workgroup that
// multiplies `total` stages into one accumulator builds a chain of 64 * total terms (512 ... 1 900 at B = 256): its
// rounding error grows with the square root of that length and at the headline batch exceeded what oneDNN's blocked
// sums leave (tests/test_gpu_model.py: float64 trajectory).  With FL the first MFMA of every stage starts from the
// inline constant 0 and the stage's 64-term sum is added to a second register set at the end of the stage: chains of
// 64 + total terms, fixed order (bitwise reproducible), no extra memory traffic; cost = one v_pk_add_f32 per two
// accumulator registers per stage (32 MFMAs of 64 cycles per register pair).
// T_T = 128 (round 5, the 64- and 32-channel tiles of blocks 0-1): a stage of 128 time steps — the fixed cost per stage (barrier,
// drained waits, DMA issue, the second-level adds) is paid half as often on the layers whose stages are shortest; needs the
// row stride to be a multiple of 128 floats, and two images of 38 KB still leave two workgroups per CU.
template <int M_T, int R_T, int WM, int WR, int WK, int KK, int FL = 1, int T_T = 64>
__global__ __launch_bounds__(256, 2) void conv1d_mfma_wgrad_dma_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, float *__restrict__ slab, int N,
    int Cin, int Cout, int L, int Lo, int ldy, int pad, int S) {
    static_assert(WM * WR * WK == 4, "4 waves per workgroup");
    static_assert(T_T == 64 || T_T == 128, "64 or 128 time steps per stage");
    constexpr int LPR = T_T / 4, RPP = 64 / LPR;        // 16-byte chunks (lanes) per dY row; rows per 1 KB DMA piece
    constexpr int MC = M_T / WM / 32, MR = R_T / WR / 32;
    constexpr int TW = T_T / WK;                        // WK > 1: the waves split the stage's t range (32-channel layer)
    constexpr int NST = TW / 2, NGRP = NST / 4;         // reduction steps per stage, in groups of 4
    static_assert(NST % 4 == 0, "whole groups of four steps");
    constexpr int XSPAN = T_T + KK - 1;
    constexpr int XS = ((XSPAN - KK + 31) / 32) * 32 + KK;   // >= XSPAN and == KK (mod 32)
    constexpr int NCI = (R_T + KK - 2) / KK + 1;
    constexpr int XEL = NCI * XS, XLOADS = (XEL + 255) / 256;
    constexpr int AEL = M_T * T_T;                      // floats of the dY image
    constexpr int NDMA = AEL / 256;                     // 1 KB pieces (4 rows each)
    constexpr int DPW = NDMA / 4;                       // pieces per wave per stage
    constexpr int IMG = ((AEL + XEL + 63) / 64) * 64;   // keeps image 1 16-byte aligned
    constexpr int ACCF = MC * MR * 16 * 64;             // floats of one wave's accumulators (WK > 1 exchange)
    static_assert(WK == 1 || WM * WR * ACCF + 4 * 32 <= 2 * IMG, "accumulator exchange must fit the dead images");
    static_assert(XS >= XSPAN, "x row stride too small");
    static_assert(NDMA % 4 == 0, "dY image must split evenly over the four waves");
    static_assert(XLOADS <= 32, "mask bits");
    static_assert(NST >= 2 * XLOADS + DPW, "not enough steps to spread the staging over");

    __shared__ __attribute__((aligned(1024))) float lds[2 * IMG];
    ECG_STAMP_AT(0);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int R = Cin * KK;
    // logical order: the R tiles of one (C_out tile, split) are adjacent — they read the same dY slice
    const int RT = (R + R_T - 1) / R_T, CT = Cout / M_T;
    const int tile = xcd_chunked(blockIdx.x, gridDim.x);
    const int tile_r = tile % RT, tile_cs = tile / RT;
    const int r0 = tile_r * R_T, co0 = (tile_cs % CT) * M_T, s = tile_cs / CT;
    const int wk = wave % WK, wr = (wave / WK) % WR, wm = wave / (WK * WR);
    const int wm0 = wm * (M_T / WM), wr0 = wr * (R_T / WR), wt0 = wk * TW;
    const int ci_base = r0 / KK;
    const int ntt = (Lo + T_T - 1) / T_T;
    // the split runs over stages (n, t tile), not whole samples: finer balance, and enough workgroups for
    // layers with a single output tile
    const int it_begin = (int)((long long)N * ntt * s / S), it_end = (int)((long long)N * ntt * (s + 1) / S);

    int xcol[MR];
#pragma unroll
    for (int j = 0; j < MR; ++j) {
        int r = r0 + wr0 + 32 * j + l31;
        if (r >= R) r = R - 1;                 // clamped columns compute garbage that is never stored
        const int ci = r / KK;
        xcol[j] = (ci - ci_base) * XS + (r - ci * KK);
    }

    f32x16 acc[MC][MR], acc2[MC][MR];      // acc2: the second level (FL), dead otherwise
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b < MR; ++b) { acc[a][b] = zero16; acc2[a][b] = zero16; }
    float bsum[MC], bsum2[MC];
#pragma unroll
    for (int a = 0; a < MC; ++a) bsum[a] = bsum2[a] = 0.f;
    const bool want_bias = (tile_r == 0) && (wr == 0);

    // ---- staging: loop-invariant per-thread pieces ------------------------------------------
    int doff[DPW];          // dY piece j of this wave: element offset from the stage base
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int row = (j * 4 + wave) * RPP + lane / LPR;          // row inside the M_T tile
        doff[j] = row * ldy + (((lane % LPR) ^ (row & 15)) << 2);   // LDS chunk c <- global chunk c^(row&15)
    }
    int xci[XLOADS], xpos[XLOADS];
#pragma unroll
    for (int j = 0; j < XLOADS; ++j) {
        const int e = min(tid + 256 * j, XEL - 1);
        xci[j] = min(ci_base + e / XS, Cin - 1) * L;
        xpos[j] = e % XS - pad;
    }
    float xreg[XLOADS];
    unsigned xmask = 0, cxmask = 0;
    const int total = it_end - it_begin;

    // stage coordinates advance incrementally (uniform, no division in the loop)
    int sn = it_begin / ntt, stt = it_begin - sn * ntt;     // stage whose x tile is loaded next
    int dn = sn, dtt = stt;                                 // stage whose dY tile is DMA'd next
    auto advance = [&]() { if (++stt == ntt) { stt = 0; ++sn; } };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_a = [&](int j, float *img) {               // one 1 KB piece of stage (dn, dtt)'s dY tile (asm LDS-DMA: glds16)
        const float *base = dy + ((size_t)min(dn, N - 1) * Cout + co0) * ldy + dtt * T_T;        // uniform
        glds16(base, (unsigned)doff[j] * 4u,
               (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)((img - lds) + (j * 4 + wave_u) * 256) * 4u)));
    };
    auto load_x = [&](int j) {                          // x tile element of stage (sn, stt)
        const float *xn = x + (size_t)min(sn, N - 1) * Cin * L;
        const int sidx = stt * T_T + xpos[j];
        xreg[j] = xn[xci[j] + min(max(sidx, 0), L - 1)];
        const unsigned bit = ((sidx >= 0) && (sidx < L)) ? (1u << j) : 0u;
        xmask = (j == 0) ? bit : (xmask | bit);
    };
    auto commit_x = [&](int j, float *img) {
        const int e = tid + 256 * j;
        const unsigned keep = 0u - ((cxmask >> j) & 1u);
        if (256 * (j + 1) <= XEL || e < XEL)
            img[AEL + e] = __uint_as_float(__float_as_uint(xreg[j]) & keep);
    };

    // prologue: stage 0 -> image 0, x tile of stage 1 -> registers
    if (total > 0) {
#pragma unroll
        for (int j = 0; j < DPW; ++j) dma_a(j, lds);
#pragma unroll
        for (int j = 0; j < XLOADS; ++j) load_x(j);
        cxmask = xmask;
#pragma unroll
        for (int j = 0; j < XLOADS; ++j) commit_x(j, lds);
        advance();
        dn = sn; dtt = stt;                             // stage 1
#pragma unroll
        for (int j = 0; j < XLOADS; ++j) load_x(j);
        advance();                                      // stage 2
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the asm DMA pieces are not in hipcc's books)
    __syncthreads();
    ECG_STAMP_AT(1);

    const int aoff = (wm0 + l31) * T_T, swz = (l31 & 15) << 2;
    for (int it = 0; it < total; ++it) {
        const float *dys = lds + (it & 1) * IMG, *xs = dys + AEL;
        float *nxt = lds + ((it + 1) & 1) * IMG;
        cxmask = xmask;                                    // mask of stage it+1 (now in registers)

        const float *arow = dys + aoff;
        const float *brow = xs + 4 * half + wt0;
        auto lda = [&](int g, f32x4 *a) {                  // group g: chunk 2g+half, swizzled
#pragma unroll
            for (int i = 0; i < MC; ++i)
                a[i] = *reinterpret_cast<const f32x4 *>(arow + 32 * i * T_T + ((((2 * g) << 2) + (half << 2) + wt0) ^ swz));
        };
        auto ldb = [&](int st, float *b) {
            const int tp = 8 * (st >> 2) + (st & 3);
#pragma unroll
            for (int j = 0; j < MR; ++j) b[j] = brow[xcol[j] + tp];
        };
        f32x4 aq_c[MC], aq_n[MC];
        float b_c[MR], b_n[MR];
        lda(0, aq_c);
        ldb(0, b_c);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int j4 = st & 3;
            ldb(st + 1 < NST ? st + 1 : 0, b_n);
            if (j4 == 2 && (st >> 2) + 1 < NGRP) lda((st >> 2) + 1, aq_n);
            // staging, one operation per step: x commits, x loads, then the dY DMA pieces.
            // UNCONDITIONAL (stage coordinates are clamped, the last stages restage a valid tile
            // nobody reads): under a branch hipcc puts s_waitcnt vmcnt(0) in front of every load.
            // (Issuing the DMA pieces FIRST — 14 instead of 4 steps of slack before the barrier on the
            // 32-channel tile — measured the same: the stage loop is not waiting for the DMA.)
            if (st < XLOADS) commit_x(st, nxt);
            else if (st < 2 * XLOADS) load_x(st - XLOADS);
            else if (st < 2 * XLOADS + DPW) dma_a(st - 2 * XLOADS, nxt);
            __builtin_amdgcn_sched_barrier(0);         // keep reads + staging ABOVE these MFMAs
#pragma unroll
            for (int i = 0; i < MC; ++i) bsum[i] = (FL && st == 0) ? aq_c[i][j4] : bsum[i] + aq_c[i][j4];
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < MR; ++j)
                    acc[i][j] = mfma32(aq_c[i][j4], b_c[j], (FL && st == 0) ? zero16 : acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < MR; ++j) b_c[j] = b_n[j];
            if (j4 == 3) {
#pragma unroll
                for (int i = 0; i < MC; ++i) aq_c[i] = aq_n[i];
            }
        }
        if (FL) {               // second level: the stage's sums join the running totals, in issue order of the MFMAs
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < MR; ++j) acc2[i][j] += acc[i][j];
#pragma unroll
            for (int i = 0; i < MC; ++i) bsum2[i] += bsum[i];
        }
        dn = sn; dtt = stt;
        advance();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // image it&1 free again; image (it+1)&1 complete (vmcnt(0) + barrier)
        if (it == 0) ECG_STAMP_AT(2);
    }
    ECG_STAMP_AT(3);
#ifdef ECG_STAMP
    if (g_stamps && threadIdx.x == 0)          // stages | HW_ID (cu / sh / se) << 16 | XCC_ID << 48: which workgroups share a CU
        g_stamps[(size_t)blockIdx.x * 8 + 5] = (unsigned long long)(total & 0xFFFF) |
            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 16) |
            ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 48);
#endif

    if (FL) {
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            bsum[i] = bsum2[i];
#pragma unroll
            for (int j = 0; j < MR; ++j) acc[i][j] = acc2[i][j];
        }
    }
    if (want_bias) {
#pragma unroll
        for (int i = 0; i < MC; ++i) bsum[i] += __shfl_xor(bsum[i], 32, 64);
    }
    if (WK > 1) {
        // sum the WK waves that share an output tile through LDS, in wave order (the images are dead)
        float *ex = lds + (wr + WR * wm) * ACCF, *bex = lds + WM * WR * ACCF;
        for (int w = 1; w < WK; ++w) {
            __syncthreads();
            if (wk == w) {
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int j = 0; j < MR; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ex[((i * MR + j) * 16 + r) * 64 + lane] = acc[i][j][r];
                if (want_bias && half == 0 && MC == 1) bex[(wr + WR * wm) * 32 + l31] = bsum[0];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int j = 0; j < MR; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] += ex[((i * MR + j) * 16 + r) * 64 + lane];
                if (want_bias && half == 0 && MC == 1) bsum[0] += bex[(wr + WR * wm) * 32 + l31];
            }
        }
        static_assert(WK == 1 || MC == 1, "the bias exchange above assumes one 32-channel row per wave");
        if (wk != 0) return;
    }
    const size_t wslab = (size_t)Cout * R;
    float *out = slab + (size_t)s * wslab;
#pragma unroll
    for (int i = 0; i < MC; ++i)
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            const int r = r0 + wr0 + 32 * j + l31;
            if (r < R) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = co0 + wm0 + 32 * i + acc_row(q, half);
                    out[(size_t)co * R + r] = acc[i][j][q];
                }
            }
        }
    if (want_bias && half == 0) {
#pragma unroll
        for (int i = 0; i < MC; ++i)
            slab[(size_t)S * wslab + (size_t)s * Cout + co0 + wm0 + 32 * i + l31] = bsum[i];
    }
#ifdef ECG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    ECG_STAMP_AT(4);
}

// ---------------------------------------------------------------------------------------
// Weight gradient by the transposed two-phase fast-FIR split (the forward's split, conv1d_mfma_ffa_kernel, read as a trilinear
// form in (dY, w, x) and differentiated by w).  With m <-> the output pair (2m, 2m + 1) of a stage, xt[p] = x[t0 - pad + p]:
//     U[co][ci][j] = sum_m (dY[2m] + dY[2m+1]) * xt[2m + 2j]                        j = 0..7
//     V[co][ci][j] = sum_m (dY[2m] + dY[2m-1]) * xt[2m + 2j + 1]                    j = 0..6, m = 0..T/2 (dY outside the stage = 0)
//     G[co][ci][j] = sum_m  dY[2m+1]           * (xt[2m + 2j] - xt[2m + 2j + 1])    j = 0..7
//     dW[2j] = U[j] - G[j],     dW[2j+1] = V[j] + G[j+1]
// 23 column families of half as many reduction steps instead of 15: 23/30 of the MFMAs.  The pairing (dY[2m], dY[2m-1]) of V
// is kept INSIDE a stage (its first and last m carry one term each: T/2 + 1 values, one extra MFMA step per stage), so
// stages stay independent.  A workgroup owns one column tile of ONE family: the A operand (the dY combination) is uniform
// per workgroup; B is one ds_read_b64 per lane and step ((xt[2m+2j], xt[2m+2j+1]); x rows 16 floats (mod 64) apart: four
// channels x eight tap pairs of a 32-lane column block cover the 64 banks once).  For smooth x the G products are small
// and xt[2m+2j] - xt[2m+2j+1] is exact, so U - G loses nothing against the direct sum (tests: float64 comparison).
// Slab layout: [s][co][U columns ci*8+j | V columns ci*7+j | G columns ci*8+j] (+ the bias slab); wgrad_ffa_reduce_kernel
// adds the slabs in double, in slab order, and forms the 15 taps.  Image, DMA, swizzle, two-level accumulation: as in
// conv1d_mfma_wgrad_dma_kernel above.
template <int M_T, int R_T, int WM, int WR, int FL, int T_T>
__global__ __launch_bounds__(256, 2) void conv1d_mfma_wgrad_ffa_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, float *__restrict__ slab, int N,
    int Cin, int Cout, int L, int Lo, int ldy, int pad, int S) {
    static_assert(WM * WR == 4, "4 waves per workgroup");
    static_assert(kKM == 15, "tap split 8 + 7");
    static_assert(T_T == 64 || T_T == 128, "64 or 128 time steps per stage");
    constexpr int LPR = T_T / 4, RPP = 64 / LPR;
    constexpr int MC = M_T / WM / 32, MR = R_T / WR / 32;
    constexpr int NST = T_T / 4, NGRP = NST / 4;        // MFMA steps per stage (two m each), in groups of 4 (= 16 t per lane half)
    constexpr int XS = T_T + 16;                        // >= T_T + 14, == 16 (mod 64), even
    static_assert(XS % 64 == 16, "x rows 16 banks apart");
    constexpr int NCI = (R_T + 6) / 7 + 1;              // channels a column tile of the 7-tap family can touch
    constexpr int XEL = NCI * XS, XLOADS = (XEL + 255) / 256;
    constexpr int AEL = M_T * T_T;
    constexpr int NDMA = AEL / 256;
    constexpr int DPW = NDMA / 4;
    constexpr int IMG = ((AEL + XEL + 63) / 64) * 64;
    constexpr int OPS = 2 * XLOADS + DPW;               // staging operations of a stage, spread over its steps
    static_assert(NDMA % 4 == 0, "dY image must split evenly over the four waves");
    static_assert(XLOADS <= 32, "mask bits");
    static_assert(2 * IMG * 4 <= 80 * 1024, "two workgroups per CU");

    __shared__ __attribute__((aligned(1024))) float lds[2 * IMG];
    ECG_STAMP_AT(0);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int RU = Cin * 8, RV = Cin * 7;
    const int TU = (RU + R_T - 1) / R_T, TV = (RV + R_T - 1) / R_T;
    const int RT = 2 * TU + TV, CT = Cout / M_T;
    const int tile = xcd_chunked(blockIdx.x, gridDim.x);
    const int tile_r = tile % RT, tile_cs = tile / RT;
    const int fam = tile_r < TU ? 0 : (tile_r < TU + TV ? 1 : 2);                 // uniform per workgroup
    const int tile_rf = tile_r - (fam == 0 ? 0 : (fam == 1 ? TU : TU + TV));
    const int NJ = fam == 1 ? 7 : 8, RF = Cin * NJ, offf = fam == 0 ? 0 : (fam == 1 ? RU : RU + RV);
    const int r0 = tile_rf * R_T, co0 = (tile_cs % CT) * M_T, s = tile_cs / CT;
    const int wr = wave % WR, wm = wave / WR;
    const int wm0 = wm * (M_T / WM), wr0 = wr * (R_T / WR);
    const int ci_base = r0 / NJ;
    const int ntt = (Lo + T_T - 1) / T_T;
    const int it_begin = (int)((long long)N * ntt * s / S), it_end = (int)((long long)N * ntt * (s + 1) / S);

    int xcol[MR];
#pragma unroll
    for (int j = 0; j < MR; ++j) {
        int r = r0 + wr0 + 32 * j + l31;
        if (r >= RF) r = RF - 1;               // clamped columns compute garbage that is never stored
        const int ci = r / NJ;
        xcol[j] = (ci - ci_base) * XS + 2 * (r - ci * NJ) + 8 * half;
    }

    f32x16 acc[MC][MR], acc2[MC][MR];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b < MR; ++b) { acc[a][b] = zero16; acc2[a][b] = zero16; }
    float bsum[MC], bsum2[MC];
#pragma unroll
    for (int a = 0; a < MC; ++a) bsum[a] = bsum2[a] = 0.f;
    const bool want_bias = (tile_r == 0) && (wr == 0);     // (family U: its A operand sums to the bias gradient)

    int doff[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int row = (j * 4 + wave) * RPP + lane / LPR;
        doff[j] = row * ldy + (((lane % LPR) ^ (row & 15)) << 2);
    }
    float xreg[XLOADS];                                 // (element -> (channel, position) is recomputed per load: registers)
    unsigned xmask = 0, cxmask = 0;
    const int total = it_end - it_begin;

    int sn = it_begin / ntt, stt = it_begin - sn * ntt;
    int dn = sn, dtt = stt;
    auto advance = [&]() { if (++stt == ntt) { stt = 0; ++sn; } };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_a = [&](int j, float *img) {
        const float *base = dy + ((size_t)min(dn, N - 1) * Cout + co0) * ldy + dtt * T_T;
        glds16(base, (unsigned)doff[j] * 4u,
               (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)((img - lds) + (j * 4 + wave_u) * 256) * 4u)));
    };
    auto load_x = [&](int j) {
        const float *xn = x + (size_t)min(sn, N - 1) * Cin * L;
        const int e = min(tid + 256 * j, XEL - 1), row = e / XS;
        const int sidx = stt * T_T + (e - row * XS) - pad;
        xreg[j] = xn[min(ci_base + row, Cin - 1) * L + min(max(sidx, 0), L - 1)];
        const unsigned bit = ((sidx >= 0) && (sidx < L)) ? (1u << j) : 0u;
        xmask = (j == 0) ? bit : (xmask | bit);
    };
    auto commit_x = [&](int j, float *img) {
        const int e = tid + 256 * j;
        const unsigned keep = 0u - ((cxmask >> j) & 1u);
        if (256 * (j + 1) <= XEL || e < XEL)
            img[AEL + e] = __uint_as_float(__float_as_uint(xreg[j]) & keep);
    };

    if (total > 0) {
#pragma unroll
        for (int j = 0; j < DPW; ++j) dma_a(j, lds);
#pragma unroll
        for (int j = 0; j < XLOADS; ++j) load_x(j);
        cxmask = xmask;
#pragma unroll
        for (int j = 0; j < XLOADS; ++j) commit_x(j, lds);
        advance();
        dn = sn; dtt = stt;
#pragma unroll
        for (int j = 0; j < XLOADS; ++j) load_x(j);
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ECG_STAMP_AT(1);

    typedef float f32x2t __attribute__((ext_vector_type(2)));
    const int aoff = (wm0 + l31) * T_T, swz = (l31 & 15) << 2;
    // one stage of family FAM (0 = U, 1 = V, 2 = G)
    auto stage = [&](auto fam_c, int it) {
        constexpr int FAM = decltype(fam_c)::value;
        const float *dys = lds + (it & 1) * IMG, *xs = dys + AEL;
        float *nxt = lds + ((it + 1) & 1) * IMG;
        cxmask = xmask;
        const float *arow = dys + aoff;
        // group g: this lane half's eight consecutive t (two 16-byte chunks, swizzled) [+ V: the element in front of them]
        // (the first chunk is read two steps before the group starts, the second during its first step: one spare set of
        // four registers per row block instead of two)
        auto lda_lo = [&](int g, f32x4 *lo, float *prev) {
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                const float *rb = arow + 32 * i * T_T;
                const int c0 = 4 * g + 2 * half;
                lo[i] = *reinterpret_cast<const f32x4 *>(rb + ((c0 << 2) ^ swz));
                if (FAM == 1) {
                    const float pv = rb[((max(c0 - 1, 0) << 2) ^ swz) + 3];
                    prev[i] = (g == 0 && half == 0) ? 0.f : pv;
                }
            }
        };
        auto lda_hi = [&](int g, f32x4 *hi) {
#pragma unroll
            for (int i = 0; i < MC; ++i)
                hi[i] = *reinterpret_cast<const f32x4 *>(arow + 32 * i * T_T + (((4 * g + 2 * half + 1) << 2) ^ swz));
        };
        auto ldb = [&](int st, f32x2t *b) {
            const int tp = 16 * (st >> 2) + 2 * (st & 3);
#pragma unroll
            for (int j = 0; j < MR; ++j) b[j] = *reinterpret_cast<const f32x2t *>(xs + xcol[j] + tp);
        };
        f32x4 lo_c[MC], hi_c[MC], lo_n[MC];
        float pv_c[MC], pv_n[MC];
        f32x2t b_c[MR], b_n[MR];
        lda_lo(0, lo_c, pv_c);
        lda_hi(0, hi_c);
        ldb(0, b_c);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int j4 = st & 3;
            ldb(st + 1 < NST ? st + 1 : 0, b_n);
            if (j4 == 2 && (st >> 2) + 1 < NGRP) lda_lo((st >> 2) + 1, lo_n, pv_n);
            if (j4 == 0 && st > 0) lda_hi(st >> 2, hi_c);
#pragma unroll
            for (int o = st * OPS / NST; o < (st + 1) * OPS / NST; ++o) {
                // dY pieces FIRST (a stage is 16 short steps: issued last they would be waited for at the barrier), then the
                // x commits, then the loads of the stage after next — the youngest vector-memory operations, left in flight
                if (o < DPW) dma_a(o, nxt);
                else if (o < DPW + XLOADS) commit_x(o - DPW, nxt);
                else load_x(o - DPW - XLOADS);
            }
            float av[MC], bv[MR];
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                const f32x4 d = j4 < 2 ? lo_c[i] : hi_c[i];
                const float d0 = d[2 * (j4 & 1)], d1 = d[2 * (j4 & 1) + 1];
                const float dp = j4 == 0 ? pv_c[i] : (j4 == 1 ? lo_c[i][1] : (j4 == 2 ? lo_c[i][3] : hi_c[i][1]));
                av[i] = FAM == 0 ? d0 + d1 : (FAM == 1 ? d0 + dp : d1);
            }
#pragma unroll
            for (int j = 0; j < MR; ++j) bv[j] = FAM == 0 ? b_c[j][0] : (FAM == 1 ? b_c[j][1] : b_c[j][0] - b_c[j][1]);
            __builtin_amdgcn_sched_barrier(0);
            if (FAM == 0) {
#pragma unroll
                for (int i = 0; i < MC; ++i) bsum[i] = (FL && st == 0) ? av[i] : bsum[i] + av[i];
            }
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < MR; ++j)
                    acc[i][j] = mfma32(av[i], bv[j], (FL && st == 0) ? zero16 : acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < MR; ++j) b_c[j] = b_n[j];
            if (j4 == 3) {
#pragma unroll
                for (int i = 0; i < MC; ++i) { lo_c[i] = lo_n[i]; pv_c[i] = pv_n[i]; }
            }
        }
        if (FAM == 1) {
            // m = T_T / 2: the last dY of the stage alone (lane half 0; half 1 contributes an exact zero)
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                const float dl = arow[32 * i * T_T + (((LPR - 1) << 2) ^ swz) + 3];
                const float a = half ? 0.f : dl;
#pragma unroll
                for (int j = 0; j < MR; ++j)
                    acc[i][j] = mfma32(a, xs[xcol[j] - 8 * half + T_T + 1], acc[i][j]);
            }
        }
        if (FL) {
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < MR; ++j) acc2[i][j] += acc[i][j];
            if (FAM == 0) {
#pragma unroll
                for (int i = 0; i < MC; ++i) bsum2[i] += bsum[i];
            }
        }
        dn = sn; dtt = stt;
        advance();
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(XLOADS) : "memory");    // the DMA pieces are older than the x loads
        __syncthreads();
    };
    if (fam == 0) { for (int it = 0; it < total; ++it) { stage(std::integral_constant<int, 0>{}, it); if (it == 0) ECG_STAMP_AT(2); } }
    else if (fam == 1) { for (int it = 0; it < total; ++it) { stage(std::integral_constant<int, 1>{}, it); if (it == 0) ECG_STAMP_AT(2); } }
    else { for (int it = 0; it < total; ++it) { stage(std::integral_constant<int, 2>{}, it); if (it == 0) ECG_STAMP_AT(2); } }
    ECG_STAMP_AT(3);
#ifdef ECG_STAMP
    if (g_stamps && threadIdx.x == 0)          // stages | HW_ID << 16 | XCC_ID << 48 (as the direct kernel)
        g_stamps[(size_t)blockIdx.x * 8 + 5] = (unsigned long long)(total & 0xFFFF) |
            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 16) |
            ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 48);
#endif

    if (FL) {
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            bsum[i] = bsum2[i];
#pragma unroll
            for (int j = 0; j < MR; ++j) acc[i][j] = acc2[i][j];
        }
    }
    if (want_bias) {
#pragma unroll
        for (int i = 0; i < MC; ++i) bsum[i] += __shfl_xor(bsum[i], 32, 64);
    }
    const size_t RVT = (size_t)Cin * 23, wslab = (size_t)Cout * RVT;
    float *out = slab + (size_t)s * wslab + offf;
#pragma unroll
    for (int i = 0; i < MC; ++i)
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            const int r = r0 + wr0 + 32 * j + l31;
            if (r < RF) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = co0 + wm0 + 32 * i + acc_row(q, half);
                    out[(size_t)co * RVT + r] = acc[i][j][q];
                }
            }
        }
    if (want_bias && half == 0) {
#pragma unroll
        for (int i = 0; i < MC; ++i)
            slab[(size_t)S * wslab + (size_t)s * Cout + co0 + wm0 + 32 * i + l31] = bsum[i];
    }
#ifdef ECG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    ECG_STAMP_AT(4);
}

// dw[co][ci][k] from the S slabs of the kernel above: one lane per (co, ci, tap pair j) forms dW[2j] = U[j] - G[j] and
// dW[2j+1] = V[j] + G[j+1] (double sums in slab order; four slabs = sixteen loads in flight per lane); the bias row behind them
// as in wgrad_reduce_kernel.  One wave per workgroup: the layers that take this form have >= 2 048 of them.
__global__ __launch_bounds__(64) void wgrad_ffa_reduce_kernel(const float *__restrict__ slab, float *__restrict__ dw,
                                                              float *__restrict__ db, int Cin, int Cout, int S) {
    constexpr int G = 1, w = 0;
    const int lane = threadIdx.x;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    const size_t npair = (size_t)Cout * Cin * 8, RVT = (size_t)Cin * 23, wslab = (size_t)Cout * RVT;
    const bool live = i < npair + Cout && (i < npair || db);
    double a[4] = {0.0, 0.0, 0.0, 0.0};            // U[j], G[j], V[j], G[j+1]  |  bias
    int j = 0;
    size_t o = 0;
    if (live) {
        if (i < npair) {
            const int co = (int)(i / ((size_t)Cin * 8)), rem = (int)(i - (size_t)co * Cin * 8);
            const int ci = rem >> 3;
            j = rem & 7;
            o = ((size_t)co * Cin + ci) * 15 + 2 * j;
            const int jv = j < 7 ? j : 6;           // (tap 15 does not exist: the lane of j = 7 re-reads valid columns and drops them)
            const float *pu = slab + (size_t)co * RVT + ci * 8 + j;
            const float *pg = slab + (size_t)co * RVT + Cin * 15 + ci * 8 + j;
            const float *pv = slab + (size_t)co * RVT + Cin * 8 + ci * 7 + jv;
            const int g1 = j < 7 ? 1 : 0;
            int s = w;
            for (; s + 3 * G < S; s += 4 * G) {
                float v[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const size_t off = (size_t)(s + u * G) * wslab;
                    v[u][0] = pu[off]; v[u][1] = pg[off]; v[u][2] = pv[off]; v[u][3] = pg[off + g1];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] += (double)v[u][e];
            }
            for (; s < S; s += G) {
                const size_t off = (size_t)s * wslab;
                a[0] += (double)pu[off]; a[1] += (double)pg[off]; a[2] += (double)pv[off]; a[3] += (double)pg[off + g1];
            }
        } else {
            const float *src = slab + (size_t)S * wslab + (i - npair);
            for (int s = w; s < S; s += G) a[0] += (double)src[(size_t)s * Cout];
        }
    }
    if (live) {
        if (i < npair) {
            dw[o] = (float)(a[0] - a[1]);
            if (j < 7) dw[o + 1] = (float)(a[2] + a[3]);
        } else db[i - npair] = (float)a[0];
    }
}

// conv1d_direct.hip: dw[i] = sum_s slab[s][i] in fixed order
int wgrad_reduce(const float *ws, float *dw, float *db, size_t wslab, int Cout, int S,
                 hipStream_t st);

#ifndef ECG_WG_FL
#define ECG_WG_FL 1          // two-level accumulation; tools build -DECG_WG_FL=0 to A/B its cost and its effect on the error
#endif
struct WgCfg { int m_t, r_t, splits; };

// Development knobs are COMPILE-TIME (A/B libraries: make VARIANT=x EXTRA="-DECG_WG_SLOTS=768"): the product library reads no
// environment variable.
#ifndef ECG_WG_SLOTS
#define ECG_WG_SLOTS 512
#endif
#ifndef ECG_WG_RT
#define ECG_WG_RT 0
#endif
#ifndef ECG_WG_TT128
#define ECG_WG_TT128 1       // 128-step stages on the 64- / 32-channel tiles (A/B: -DECG_WG_TT128=0)
#endif

static WgCfg wgrad_cfg(int N, int Cin, int Cout, int Lo, bool dma, int tt = 64) {
    const int R = Cin * kKM;
    constexpr int slots = ECG_WG_SLOTS, rt128 = ECG_WG_RT;
    WgCfg c;
    // 128 x 192 tiles where 128-wide column tiles would leave a ragged last tile and 192 divide the columns (block 2:
    // R = 960 = 5 x 192 instead of 7.5 x 128: 154.7 -> 150.3 us; block 3, R = 1920 = 15 x 128 = 10 x 192: 264.0 vs 267.5 us,
    // stays at 128).  More, smaller workgroups (ECG_WG_SLOTS 768 / 1024: a second round in the slots the early finishers
    // free) measured 2-8 % SLOWER on every layer — the second prologue / slab costs more than the tail it evens out.
    // (With the two-level accumulation the 128 x 192 tile needs 96 + 96 accumulator registers and spills: it is only
    // instantiated in the single-level A/B build; block 2 then takes 128 x 128 tiles, +3 us.)
    const bool wide = dma && !ECG_WG_FL && R % 192 == 0 && (rt128 == 192 || (rt128 == 0 && R % 128 != 0));
    if (Cout % 128 == 0) c = {128, wide ? 192 : 128, 0};
    else if (Cout % 64 == 0) c = {64, 128, 0};
    else c = {32, 192, 0};
    const int tiles = cdiv(R, c.r_t) * (Cout / c.m_t);
    int s = slots / tiles;             // fill, but never exceed, the 2 x 256 resident-workgroup slots:
                                       // one workgroup over and the launch takes two rounds.
                                       // (64-channel tiles at 4 workgroups per CU measured no better here; nor did
                                       // 128 x 256 column tiles for block 3: 273 vs 271 us, 236 VGPRs.)
    // the DMA kernel splits over stages (n, 64- or 128-wide t tile), the register-staged one over samples
    const long long cap = dma ? (long long)N * cdiv(Lo, tt) : N;
    if (s > cap) s = (int)cap;
    if (s < 1) s = 1;
    c.splits = s;
    return c;
}

#ifndef ECG_WG_FFA
#define ECG_WG_FFA 1         // fast-FIR weight gradient on the 64- / 128-channel tiles (A/B: -DECG_WG_FFA=0)
#endif
// fast-FIR form: column tiles of 128 over the three families (U: 8 C_in, V: 7 C_in, G: 8 C_in columns), 64-step stages
// Measured (B = 256, 12x1000, same box): 128 -> 256 channels 290.3 -> 277.0 us; 64 -> 128 164.5 -> 164.8; 32 -> 64 88.9 -> 94.4 —
// a stage is half as many MFMA steps between the same barrier, x commits and second-level adds (~2 000 cycles per stage
// in both forms), and 128-step stages do not fit two workgroups per CU here: used where the column count makes it pay.
static bool wgrad_ffa_ok(int Cin, int Cout, bool dma) { return ECG_WG_FFA && dma && Cout % 128 == 0 && Cin >= 128; }
static int wgrad_ffa_splits(int N, int Cin, int Cout, int Lo) {
    const int m_t = Cout % 128 == 0 ? 128 : 64;
    const int tiles = (2 * cdiv(Cin * 8, 128) + cdiv(Cin * 7, 128)) * (Cout / m_t);
    long long s = ECG_WG_SLOTS / tiles;
    const long long cap = (long long)N * cdiv(Lo, 64);
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return (int)s;
}

bool mfma_wgrad_dma_supported(int Cin, int Cout, int K);
// what ecg_conv1d_multiplies_per_output_pair reports: mirrors the dispatch of launch_fwd / mfma_wgrad
int mfma_multiplies_per_pair(int op, int Cin, int Cout, int K, int pad) {
    if (op == 2) return (K == kKM && wgrad_ffa_ok(Cin, Cout, mfma_wgrad_dma_supported(Cin, Cout, K))) ? 23 : 2 * K;
    const bool mfma = op == 1 ? mfma_fwd_supported(Cout, Cin, K, K - 1 - pad) : mfma_fwd_supported(Cin, Cout, K, pad);
    return (mfma && (ECG_FWD_FFA & (op == 0 ? 1 : (op == 1 ? 2 : 4)))) ? 23 : 2 * K;
}

bool mfma_wgrad_supported(int Cin, int Cout, int K, int pad) {
    (void)pad; (void)Cin;
    return K == kKM && Cout % 32 == 0;
}

size_t mfma_wgrad_ws_floats(int N, int Cin, int Cout, int L, int K, int pad) {
    const int Lo = L + 2 * pad - K + 1;
    const WgCfg a = wgrad_cfg(N, Cin, Cout, Lo, false), b = wgrad_cfg(N, Cin, Cout, Lo, true);      // (128-step stages: never more splits)
    size_t need = (size_t)(a.splits > b.splits ? a.splits : b.splits) * ((size_t)Cout * Cin * K + Cout);
    if (wgrad_ffa_ok(Cin, Cout, true)) {
        const size_t ffa = (size_t)wgrad_ffa_splits(N, Cin, Cout, Lo) * ((size_t)Cout * Cin * 23 + Cout);
        if (ffa > need) need = ffa;
    }
    return need;
}

// dY rows that the DMA kernel can stream: 64-float multiples with a zero pad (see the kernel)
bool mfma_wgrad_dma_supported(int Cin, int Cout, int K) {
    (void)Cin;
    return K == kKM && Cout % 32 == 0;
}

int mfma_wgrad(const float *dy, int ldy, const float *x, float *dw, float *db, float *ws, int N,
               int Cin, int Cout, int L, int K, int pad, hipStream_t st) {
    const int Lo = L + 2 * pad - K + 1;
    const int R = Cin * K;
    const bool dma = mfma_wgrad_dma_supported(Cin, Cout, K) && ldy % 64 == 0 && ldy >= cdiv(Lo, 64) * 64 &&
                     (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
    // 128-step stages for the small tiles (blocks 0-1) when the rows allow it (stride a multiple of 128 floats, zero pad to it)
    const bool tt128 = ECG_WG_TT128 && dma && Cout % 128 != 0 && ldy % 128 == 0 && ldy >= cdiv(Lo, 128) * 128;
    if (wgrad_ffa_ok(Cin, Cout, dma)) {
        const int S = wgrad_ffa_splits(N, Cin, Cout, Lo), m_t = Cout % 128 == 0 ? 128 : 64;
        const int tiles = (2 * cdiv(Cin * 8, 128) + cdiv(Cin * 7, 128)) * (Cout / m_t);
        dim3 fgrid((unsigned)(tiles * S)), fblock(256);
        if (m_t == 128)
            hipLaunchKernelGGL((conv1d_mfma_wgrad_ffa_kernel<128, 128, 2, 2, ECG_WG_FL, 64>), fgrid, fblock, 0, st, dy, x, ws, N, Cin,
                               Cout, L, Lo, ldy, pad, S);
        else
            hipLaunchKernelGGL((conv1d_mfma_wgrad_ffa_kernel<64, 128, 2, 2, ECG_WG_FL, 64>), fgrid, fblock, 0, st, dy, x, ws, N, Cin,
                               Cout, L, Lo, ldy, pad, S);
        int frc = check_launch("conv1d_mfma_wgrad_ffa_kernel");
        if (frc) return frc;
        const size_t nout = (size_t)Cout * Cin * 8 + Cout;          // one lane per (co, ci, tap pair) + the bias row
        const dim3 rgrid((unsigned)cdiv(nout, (size_t)64));
        hipLaunchKernelGGL(wgrad_ffa_reduce_kernel, rgrid, dim3(64), 0, st, ws, dw, db, Cin, Cout, S);
        return check_launch("wgrad_ffa_reduce_kernel");
    }
    const WgCfg c = wgrad_cfg(N, Cin, Cout, Lo, dma, tt128 ? 128 : 64);
    dim3 grid((unsigned)(cdiv(R, c.r_t) * (Cout / c.m_t) * c.splits)), block(256);
#define ECG_WG(KERNEL) \
    hipLaunchKernelGGL(KERNEL, grid, block, 0, st, dy, x, ws, N, Cin, Cout, L, Lo, ldy, pad, c.splits)
#if !ECG_WG_FL
    if (dma && c.m_t == 128 && c.r_t == 192) ECG_WG((conv1d_mfma_wgrad_dma_kernel<128, 192, 2, 2, 1, kKM, 0>));
    else
#endif
    if (dma && c.m_t == 128) ECG_WG((conv1d_mfma_wgrad_dma_kernel<128, 128, 2, 2, 1, kKM, ECG_WG_FL>));
    else if (tt128 && c.m_t == 64) ECG_WG((conv1d_mfma_wgrad_dma_kernel<64, 128, 2, 2, 1, kKM, ECG_WG_FL, 128>));
    else if (tt128) ECG_WG((conv1d_mfma_wgrad_dma_kernel<32, 192, 1, 2, 2, kKM, ECG_WG_FL, 128>));
    else if (dma && c.m_t == 64) ECG_WG((conv1d_mfma_wgrad_dma_kernel<64, 128, 2, 2, 1, kKM, ECG_WG_FL>));
    else if (dma) ECG_WG((conv1d_mfma_wgrad_dma_kernel<32, 192, 1, 2, 2, kKM, ECG_WG_FL>));
    else if (c.m_t == 128) ECG_WG((conv1d_mfma_wgrad_kernel<128, 128, 2, 2, 1, 64, kKM, ECG_WG_FL>));
    else if (c.m_t == 64) ECG_WG((conv1d_mfma_wgrad_kernel<64, 128, 2, 2, 1, 64, kKM, ECG_WG_FL>));
    else ECG_WG((conv1d_mfma_wgrad_kernel<32, 192, 1, 2, 2, 128, kKM, ECG_WG_FL>));
#undef ECG_WG
    int rc = check_launch("conv1d_mfma_wgrad_kernel");
    if (rc) return rc;
    return wgrad_reduce(ws, dw, db, (size_t)Cout * R, Cout, c.splits, st);
}

}  // namespace ecg

#ifdef ECG_STAMP
extern "C" __attribute__((visibility("default"))) int ecg_debug_set_stamp_buffer(unsigned long long *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(ecg::g_stamps), &buf, sizeof(buf));
}
#endif
