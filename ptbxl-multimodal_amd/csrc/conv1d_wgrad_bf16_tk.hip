// conv1d_wgrad_bf16_tk.hip — mixed-precision weight gradient with TIME on the MFMA's K axis (round 4).
//
//   dW[co][ci][k] = sum_n sum_t dY[n][co][t] * X[n][ci][t + k - pad]
//
// conv1d_wgrad_bf16.hip puts a group of 16 SAMPLES on the K axis of v_mfma_f32_32x32x16_bf16 because a tap shift along t
// moves the x operand by 2 bytes — an unaligned 16-byte fragment.  The price was a second copy of every activation and
// of every dY in the "n16" layout, written by the BatchNorm passes: 0.49 GB of the 2.3 GB those passes move per step of
// BASELINE config 5, plus a packing pass over the network input.  This kernel reads the tensors the OTHER two convs read —
// bf16 [N][C][ld] rows — and takes the 16 reduction indices of an MFMA along t:
//   A[co][j] = dY[n][co][t0 + j]            one aligned ds_read_b128 per lane (8 consecutive t)
//   B[j][(ci,k)] = X[n][ci][t0 + j + k - 7] the x tile is kept FOUR times in LDS, shifted by 0 / 1 / 2 / 3 elements (built
//                                           with v_alignbit when the tile is committed): a fragment that starts at element e
//                                           reads copy e % 4 at an 8-byte aligned address — two ds_read_b64, the same LDS
//                                           cycles as one aligned ds_read_b128.  (Measured on the way: gfx950 serves a 2-byte
//                                           aligned ds_read_b128, but at 1/11 of the aligned rate — tools/lds_unaligned_bench.hip;
//                                           two copies read with ds_read2_b32 pairs run at a quarter of the b64 rate on 32
//                                           banks: 3x the LDS cycles of the n16 kernel, 190 vs 147 us on block 3.)
// A stage is (sample n, 128 time steps) = 8 MFMA k-steps; workgroup tile M_T (co) x 512 columns (32 input channels: two per
// 32-lane column block, 15 taps each), eight waves,
// one workgroup per CU, two dY images and two x images in LDS:
//   dY tile [M_T][128] bf16 (32 KB at M_T = 128) streams global -> LDS by DMA (asm, see conv1d_bf16_ring.hip), rows of 16
//     16-byte slots with slot s of row co stored at s ^ (co & 15) (conflict-free A reads) — needs dY rows zero-filled to a
//     multiple of 128 (ecg_conv1d_bf16_tk_dy_stride), which is what makes a ragged last tile harmless;
//   x tile [NCI][152] bf16 is register-staged one stage ahead (zero padding by mask, the four copies written with four
//     ds_write_b128 per 16-byte chunk); x is the previous block's bf16 activation [N][C_in][ldx] (rows zero-filled past L),
//     or the fp32 network input, rounded to bf16 here (block 0: no packing pass at all).
// Split over stages into <= 256 slabs (<= 512 on the 32-channel plan) summed in fixed order by wgrad_reduce_kernel, bias
// gradient on the dY fragments.
// Exact on bf16-rounded operands up to fp32 accumulation order (tests/test_gpu_ops.py::test_bf16_tk_*).
// Replaces autograd's conv weight-gradient (reference src/models/ecg_cnn.py:13 via loss.backward()).
#include "common.h"

namespace ecg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4t __attribute__((ext_vector_type(4)));

namespace tk {

constexpr int KK = 15, PAD = 7;
constexpr int TT = 128;          // time steps per stage
constexpr int KS = TT / 16;      // MFMA k-steps per stage
constexpr int XC = 19;           // 16-byte chunks of an x row image: elements t0 - 8 ... t0 + 143
constexpr int XRS = 320;         // x row stride in bytes: 80 dwords = 16 banks (mod 64) between input channels
constexpr int XCOPY_PAD[4] = {0, 32, 128, 160};   // the four shifted copies start 0 / 8 / 32 / 40 banks into a bank row: the
                                                  // (channel, copy) windows of 8 banks that the 32 lanes of a half-wave
                                                  // read then overlap only between channel c and c + 2

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const u16 a = __builtin_bit_cast(u16, (__bf16)lo), b = __builtin_bit_cast(u16, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

// WK > 1: the WK waves of a wave-tile group take every WK-th k-step of a stage and are summed through LDS at the end (the
// 32-channel first layer: one 32 x 192 tile, MFMA work for two waves — the kernel is a stream of dY rows there).
template <int M_T, int R_T, int WM, int WR, int WK, bool XF32>
__global__ __launch_bounds__(512) void conv1d_wgrad_bf16_tk_kernel(
    const u16 *__restrict__ dy, const void *__restrict__ xin, float *__restrict__ slab, int N, int Cin, int Cout,
    int L, int ldy, int ldx, int ntt, int S) {
    static_assert(WM * WR * WK == 8, "8 waves per workgroup");
    constexpr bool AHEAD2 = (WK > 1);                       // x tiles prefetched two stages ahead (short stages)
    constexpr int MC = M_T / WM / 32, MR = R_T / WR / 32;
    static_assert((MC == 2 && (MR == 4 || MR == 2)) || (MC == 1 && MR == 3), "wave tile 64 x 128, 64 x 64 or 32 x 96");
    constexpr int KPW = KS / WK;                            // k-steps per wave and stage
    static_assert(KS % WK == 0 && KPW % WR == 0, "k-steps split evenly; bias rows rotate over the WR waves: every k-step exactly once");
    // Columns of a tile <-> (input channel, tap): every 32-lane MFMA column block holds exactly TWO channels (lanes 0-14 and
    // 15-29; lanes 30 / 31 idle) — a tile of R_T columns is CPT = R_T / 16 channels.  With r = ci * 15 + tap on the lanes (rounds
    // 4) a block straddled three channels and the third one's bank window (16 banks per channel and copy, see XCOPY_PAD)
    // coincided with the first one's: one 2-way conflict in every fragment read, i.e. twice the LDS cycles for the x operand
    // (SQ_LDS_BANK_CONFLICT = 36 % of the LDS cycles of blocks 1-3).  The padded width is the same for the model's layers
    // (128 channels: 64 blocks x 32 = 2048 columns, as 4 x 512 before).
    constexpr int CPT = R_T / 16;
    constexpr int NCI = CPT;
    constexpr int AIMG = M_T * 16 * 16;                    // dY image: M_T rows of 16 slots of 16 bytes
    constexpr int ADMA = AIMG / 1024, APW = (ADMA + 7) / 8; // 1 KB DMA pieces, per wave
    static_assert(ADMA % 8 == 0, "dY image must split evenly over the eight waves");
    constexpr int XCSZ = NCI * XRS;                        // bytes of one copy
    constexpr int XIMG = ((4 * XCSZ + XCOPY_PAD[3] + 255) / 256) * 256;
    constexpr int XITEMS = NCI * XC, XI = (XITEMS + 511) / 512;
    constexpr int ACCF = MC * MR * 16 * 64;                // floats of one wave's accumulators (WK > 1 exchange)
    static_assert(2 * (AIMG + XIMG) <= 160 * 1024, "LDS");
    static_assert(WK == 1 || WM * WR * ACCF * 4 + 8 * MC * 32 * 4 <= 2 * (AIMG + XIMG), "accumulator exchange must fit the dead images");

    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (AIMG + XIMG)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int R = Cin * KK;
    const int RT = (Cin + CPT - 1) / CPT, CT = Cout / M_T;
    int tile;
    {       // XCD-aware order (conv1d_mfma.hip): the R tiles of one (C_out tile, split) share a dY slice
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_r = tile % RT, tile_cs = tile / RT;
    const int co0 = (tile_cs % CT) * M_T, s = tile_cs / CT;
    const int wk = wave % WK, wr = (wave / WK) % WR, wm = wave / (WK * WR);
    const int wm0 = wm * (M_T / WM), wr0 = wr * (R_T / WR);
    const int ci_base = tile_r * CPT;
    // this lane's column: channel ci_base + 2 * (block index) + (lane >= 15), tap lane % 15; lanes 30 / 31 read what lane 29 reads
    const int lch = l31 >= 15 ? 1 : 0, ltap = min(l31, 29) - 15 * lch;
    const int total = N * ntt;
    const int st_begin = (int)((long long)total * s / S), st_end = (int)((long long)total * (s + 1) / S);

    // ---- fragment addressing: byte offsets inside an image ---------------------------------------------
    const int aoffl = (wm0 + l31) * 256;                    // + ((2 ks + half) ^ (co & 15)) * 16, + i * 32 rows
    const int asw = l31 & 15;                               // (wm0 + 32 i) is a multiple of 16: co & 15 = l31 & 15
    int boffl[MR];
#pragma unroll
    for (int j = 0; j < MR; ++j) {
        const int row = 2 * ((wr0 >> 5) + j) + lch;         // row of the x image (columns past C_in compute garbage that is never stored)
        // first element of the fragment at k-step 0, half 0: e = tap + 1 -> copy e % 4 at byte 2 (e - e % 4): a multiple of 8
        const int e = ltap + 1, sft = e & 3;
        boffl[j] = row * XRS + sft * XCSZ + (sft == 0 ? XCOPY_PAD[0] : sft == 1 ? XCOPY_PAD[1] : sft == 2 ? XCOPY_PAD[2] : XCOPY_PAD[3])
                   + 2 * (e - sft) + 16 * half + 32 * wk;   // (this wave's first k-step of a stage is wk)
    }

    f32x16 acc[MC][MR];
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b < MR; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[MC];
#pragma unroll
    for (int a = 0; a < MC; ++a) bsum[a] = 0.f;
    const bool want_bias = (tile_r == 0);

    // ---- dY DMA geometry (loop-invariant per lane) ---------------------------------------------------------
    int aoff[APW];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int sl = min((j * 8 + wave) * 64 + lane, M_T * 16 - 1);
        const int co = sl >> 4, ls = sl & 15;
        const int gs = ls ^ (co & 15);                      // LDS slot ls of row co holds the row's 16-byte chunk gs
        aoff[j] = (co * ldy + 8 * gs) * 2;                  // bytes
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    auto glds = [&](const u16 *base, unsigned voff, unsigned dst_off) __attribute__((always_inline)) {
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + dst_off));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
    };

    // ---- x staging geometry ------------------------------------------------------------------------------
    int xrow[XI], xe[XI], xdst[XI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int idx = min(tid + 512 * j, XITEMS - 1);
        const int row = idx / XC, c = idx - row * XC;
        xrow[j] = min(ci_base + row, Cin - 1) * ldx;       // clamped rows feed columns that are never stored
        xe[j] = 8 * c - 8;
        xdst[j] = row * XRS + 16 * c;
    }
    u32x4 xE[XI];
    unsigned xN[XI][2];                                     // the first two dwords of the next chunk (shifted copies)

    // Two cursors over the stages of this workgroup: `d*` = the stage whose dY tile is DMA'd next (one stage ahead of the
    // MFMAs), `x*` = the stage whose x tile is LOADED next: one ahead, or (AHEAD2) two ahead — a tile is then loaded during
    // stage s, committed to LDS during stage s + 1 and multiplied in stage s + 2, so the loads have a whole stage to land
    // however short a stage is
    int dn = st_begin / ntt, dt0 = (st_begin - dn * ntt) * TT;
    int xn = dn, xt0 = dt0;
    auto advance = [&](int &n_, int &t_) __attribute__((always_inline)) {
        t_ += TT;
        if (t_ >= ntt * TT) { t_ = 0; ++n_; }
        if (n_ >= N) { n_ = N - 1; t_ = (ntt - 1) * TT; }   // past the end: restage a valid tile nobody reads
    };
    auto dma_a = [&](int j, int aimg) __attribute__((always_inline)) {
        if (ADMA % 8 == 0 || j * 8 + wave < ADMA) {
            const u16 *abase = dy + ((size_t)dn * Cout + co0) * ldy + dt0;                    // uniform
            glds(abase, (unsigned)aoff[j], (unsigned)(aimg * AIMG + (j * 8 + wave) * 1024));
        }
    };
    auto load_x = [&](int j) __attribute__((always_inline)) {
        const int g0 = xt0 + xe[j];
        if (XF32) {
            const float *xr = static_cast<const float *>(xin) + (size_t)xn * Cin * ldx + xrow[j];
            // (host: L % 8 == 0, so a chunk is inside the row or outside it as a whole; the clamp only keeps the load in bounds)
            const int gc = min(max(g0, 0), L - 8);
            const f32x4t a = *reinterpret_cast<const f32x4t *>(xr + gc), b = *reinterpret_cast<const f32x4t *>(xr + gc + 4);
            const int gn = min(max(g0 + 8, 0), L - 4);
            const f32x4t nx = *reinterpret_cast<const f32x4t *>(xr + gn);
            const bool ok = g0 >= 0 && g0 < L;
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = ok ? a[i] : 0.f; v[4 + i] = ok ? b[i] : 0.f; }
#pragma unroll
            for (int i = 0; i < 4; ++i) xE[j][i] = pack2(v[2 * i], v[2 * i + 1]);
            const bool okn = g0 + 8 >= 0 && g0 + 8 < L;
            xN[j][0] = okn ? pack2(nx[0], nx[1]) : 0u;
            xN[j][1] = okn ? pack2(nx[2], nx[3]) : 0u;
        } else {
            const u16 *xr = static_cast<const u16 *>(xin) + (size_t)xn * Cin * ldx + xrow[j];
            const int gc = min(max(g0, 0), ldx - 8);
            xE[j] = *reinterpret_cast<const u32x4 *>(xr + gc);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 nx = *reinterpret_cast<const u32x2 *>(xr + min(max(g0 + 8, 0), ldx - 4));
            const bool ok = g0 >= 0 && g0 < ldx;           // rows are zero-filled from L to ldx by their producer
#pragma unroll
            for (int i = 0; i < 4; ++i) xE[j][i] = ok ? xE[j][i] : 0u;
            const bool okn = g0 + 8 >= 0 && g0 + 8 < ldx;
            xN[j][0] = okn ? nx[0] : 0u;
            xN[j][1] = okn ? nx[1] : 0u;
        }
    };
    // copy s holds the row shifted by s elements: copy_s[e] = E[e + s].  One ds_write_b128 per (item, copy): `commit_piece`
    // writes copy c of item j, so that a stage can spread the writes over its k-steps (a burst of XI * 4 wide stores at the top
    // of a stage holds up the fragment reads queued behind it: ~13 LDS-path cycles per store and wave)
    auto commit_piece = [&](int j, int c, int ximg) __attribute__((always_inline)) {
        if (512 * (j + 1) <= XITEMS || tid + 512 * j < XITEMS) {
            const unsigned e0 = xE[j][0], e1 = xE[j][1], e2 = xE[j][2], e3 = xE[j][3], n0 = xN[j][0], n1 = xN[j][1];
            u32x4 v;
            if (c == 0) v = xE[j];
            else if (c == 1) {
                v[0] = __builtin_amdgcn_alignbit(e1, e0, 16); v[1] = __builtin_amdgcn_alignbit(e2, e1, 16);
                v[2] = __builtin_amdgcn_alignbit(e3, e2, 16); v[3] = __builtin_amdgcn_alignbit(n0, e3, 16);
            } else if (c == 2) { v[0] = e1; v[1] = e2; v[2] = e3; v[3] = n0; }
            else {
                v[0] = __builtin_amdgcn_alignbit(e2, e1, 16); v[1] = __builtin_amdgcn_alignbit(e3, e2, 16);
                v[2] = __builtin_amdgcn_alignbit(n0, e3, 16); v[3] = __builtin_amdgcn_alignbit(n1, n0, 16);
            }
            unsigned char *dst = lds + 2 * AIMG + ximg * XIMG + xdst[j] + c * XCSZ + (c == 0 ? XCOPY_PAD[0] : c == 1 ? XCOPY_PAD[1] : c == 2 ? XCOPY_PAD[2] : XCOPY_PAD[3]);
            *reinterpret_cast<u32x4 *>(dst) = v;
        }
    };
    auto commit_x = [&](int j, int ximg) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) commit_piece(j, c, ximg);
    };

    if (st_begin < st_end) {
#pragma unroll
        for (int j = 0; j < APW; ++j) dma_a(j, 0);
#pragma unroll
        for (int j = 0; j < XI; ++j) load_x(j);
#pragma unroll
        for (int j = 0; j < XI; ++j) commit_x(j, 0);
        advance(dn, dt0);
        advance(xn, xt0);
        if (AHEAD2) {
#pragma unroll
            for (int j = 0; j < XI; ++j) load_x(j);         // the second stage's tile: committed during the first
            advance(xn, xt0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    for (int st = st_begin; st < st_end; ++st) {
        const int img = (st - st_begin) & 1;
        const int abase_l = img * AIMG, xbase_l = 2 * AIMG + img * XIMG;
        // (per-lane terms are made opaque where they are used: left alone, hipcc precomputes one address register per
        // (k-step, fragment) and spills — conv1d_wgrad_bf16.hip)
        auto ld_a = [&](int i_, int i) __attribute__((always_inline)) {      // k-step wk + WK * i_
            int sw = asw;
            asm volatile("" : "+v"(sw));
            return *reinterpret_cast<const bf16x8 *>(&lds[abase_l + aoffl + (((2 * WK * i_ + 2 * wk + half) ^ sw) << 4) + i * (32 * 256)]);
        };
        // two base registers per fragment, 8 bytes apart and opaque to the compiler: with one base it fuses the two 8-byte
        // reads into ds_read2_b64, which runs at a quarter of the ds_read_b64 rate (MI355X_MICROARCH.md, LDS table)
        int bl[MR], bh[MR];
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            bl[j] = xbase_l + boffl[j];
            bh[j] = bl[j] + 8;
            asm volatile("" : "+v"(bl[j]), "+v"(bh[j]));
        }
        auto ld_b = [&](int i_, int j) __attribute__((always_inline)) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(&lds[bl[j] + 32 * WK * i_]);          // 8-byte aligned
            const u32x2 hi = *reinterpret_cast<const u32x2 *>(&lds[bh[j] + 32 * WK * i_]);
            u32x4 v;
            v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
            return __builtin_bit_cast(bf16x8, v);
        };
        bf16x8 a_c[MC], a_n[MC], b_c[MR];
#pragma unroll
        for (int i = 0; i < MC; ++i) a_c[i] = ld_a(0, i);
#pragma unroll
        for (int j = 0; j < MR; ++j) b_c[j] = ld_b(0, j);
        // staging.  Top of the stage: the dY DMA pieces of the next stage (image img ^ 1 is free since the last barrier).
        // AHEAD2 (short stages: the k-split 32-channel plan): the x tile loaded during the LAST stage goes into the other x image
        // in the first half of the k-steps (its XI * 4 wide stores spread over them); then, once the registers are free, the
        // loads of the tile after it — half a stage before their first use.  vmcnt retires in order (pieces first, x loads
        // last), so the wait at the end of the stage lets exactly the x loads stay in flight.
        // Otherwise (8 k-steps per wave): x loads of the NEXT stage at the first k-step, committed two k-steps before the stage
        // ends — measured 4-5 % faster on blocks 1-3 than the two-stages-ahead schedule (157 against 164-168 us on block 3).
        if (AHEAD2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the x loads issued half a stage ago)
#pragma unroll
        for (int j = 0; j < APW; ++j) dma_a(j, img ^ 1);
        constexpr int NWR = XI * 4, HK = KPW >= 2 ? KPW / 2 : 1, WPK = (NWR + HK - 1) / HK;     // stores per k-step of the first half
#pragma unroll
        for (int i_ = 0; i_ < KPW; ++i_) {
            if (AHEAD2) {
                if (i_ < HK) {
#pragma unroll
                    for (int w = i_ * WPK; w < (i_ + 1) * WPK && w < NWR; ++w) commit_piece(w >> 2, w & 3, img ^ 1);
                }
                if (i_ == (KPW >= 2 ? HK : 0)) {
#pragma unroll
                    for (int j = 0; j < XI; ++j) load_x(j);
                }
            } else {
                if (i_ == 0) {
#pragma unroll
                    for (int j = 0; j < XI; ++j) load_x(j);
                }
                // 128-channel tiles: the two waves of a SIMD (w and w + 4) commit at different k-steps, three apart — one's stores
                // and shifts run under the other's MFMAs instead of beside its stores (same-box A/B, B=256 12x5000: block 3
                // 184.0 -> 177.4 us, block 2 99.8 -> 98.3; two apart -4.7 / -0.6, four apart +5.6 / +3.2; the 64-channel plan of
                // block 1 loses 0.9 us and keeps one commit point)
                constexpr int STG = (M_T == 128) ? 3 : 0;
                if (i_ == KPW - 2 || (STG && i_ == KPW - 2 - STG)) {
                    if (STG == 0 || ((i_ == KPW - 2) == (wave < 4))) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (x loads and DMA pieces issued at the top of the stage)
#pragma unroll
                        for (int j = 0; j < XI; ++j) commit_x(j, img ^ 1);
                    }
                }
            }
            const int kn = i_ + 1 < KPW ? i_ + 1 : KPW - 1;           // (the last step re-reads its own fragments: unused)
#pragma unroll
            for (int i = 0; i < MC; ++i) a_n[i] = ld_a(kn, i);
            if (want_bias && (i_ % WR) == wr) {                        // (bf16-rounded dY, fp32 sum; every k-step exactly once)
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[i] += (float)a_c[i][e];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < MR; ++j) {
#pragma unroll
                for (int i = 0; i < MC; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c[i], b_c[j], acc[i][j], 0, 0, 0);
                b_c[j] = ld_b(kn, j);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MC; ++i) a_c[i] = a_n[i];
        }
        advance(dn, dt0);
        advance(xn, xt0);
        // this wave's x commits (LDS) and DMA pieces must have landed before the barrier publishes image img ^ 1; the x loads
        // issued BEHIND the pieces (at least one vector-memory instruction per item) may stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(AHEAD2 ? XI : 0) : "memory");
        __syncthreads();                                                 // image img is free again; image img ^ 1 is complete
    }

    // ---- combine: bias partials of the waves that hold the same channel rows; accumulators of the WK k-split waves ----
    if (want_bias) {                                        // uniform per workgroup; the images are dead (last barrier of the loop)
        float *bred = reinterpret_cast<float *>(lds) + (WK > 1 ? WM * WR * ACCF : 0);
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            bsum[i] += __shfl_xor(bsum[i], 32, 64);
            if (half == 0) bred[wave * (MC * 32) + 32 * i + l31] = bsum[i];
        }
        __syncthreads();
        if (wr == 0 && wk == 0) {
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                float t = 0.f;
                for (int w = 0; w < WR * WK; ++w) t += bred[(wm * WR * WK + w) * (MC * 32) + 32 * i + l31];     // fixed order
                bsum[i] = t;
            }
        }
    }
    if (WK > 1) {
        float *ex = reinterpret_cast<float *>(lds) + (wr + WR * wm) * ACCF;
        for (int w = 1; w < WK; ++w) {
            __syncthreads();
            if (wk == w) {
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int j = 0; j < MR; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ex[((i * MR + j) * 16 + r) * 64 + lane] = acc[i][j][r];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int j = 0; j < MR; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] += ex[((i * MR + j) * 16 + r) * 64 + lane];
            }
        }
        if (wk != 0) return;
    }
    const size_t wslab = (size_t)Cout * R;
    float *out = slab + (size_t)s * wslab;
#pragma unroll
    for (int i = 0; i < MC; ++i)
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            const int ci = ci_base + 2 * ((wr0 >> 5) + j) + lch;
            if (l31 < 30 && ci < Cin) {
                const int r = ci * KK + ltap;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = co0 + wm0 + 32 * i + acc_row(q, half);
                    out[(size_t)co * R + r] = acc[i][j][q];
                }
            }
        }
    if (want_bias && wr == 0 && half == 0) {
#pragma unroll
        for (int i = 0; i < MC; ++i)
            slab[(size_t)S * wslab + (size_t)s * Cout + co0 + wm0 + 32 * i + l31] = bsum[i];
    }
}

}  // namespace tk

// conv1d_direct.hip: dw[i] = sum_s slab[s][i] in fixed order
int wgrad_reduce(const float *ws, float *dw, float *db, size_t wslab, int Cout, int S, hipStream_t st);

#ifndef ECG_TK_RT128
#define ECG_TK_RT128 512         // column-tile width of the 128-channel plan (A/B knob: 256 halves the slab bytes per layer)
#endif
#ifndef ECG_TK_SLOTS32
#define ECG_TK_SLOTS32 512       // workgroups of the 32-channel tile plan (compile-time A/B knob: make VARIANT=.. EXTRA=-DECG_TK_SLOTS32=256)
#endif
struct TkPlan { int m_t, r_t, ntt, splits; size_t slab_floats; };

static TkPlan tk_plan(int N, int Cin, int Cout, int Lo) {
    TkPlan p{0, 512, cdiv(Lo, tk::TT), 1, 0};
    if (Cout % 32 != 0) return p;
    p.m_t = Cout % 128 == 0 ? 128 : Cout % 64 == 0 ? 64 : 32;
    if (p.m_t == 32) p.r_t = 192;                           // the 32-channel first layer: 32 x 192 tiles, k-steps split over waves
    if (p.m_t == 128) p.r_t = ECG_TK_RT128;
    const int tiles = cdiv(Cin, p.r_t / 16) * (Cout / p.m_t);      // a tile of r_t columns holds r_t / 16 input channels
    int s = (p.m_t == 32 ? ECG_TK_SLOTS32 : 256) / tiles;   // one eight-wave workgroup per CU (two of the small ones)
    const int total = N * p.ntt;
    if (s > total / 4) s = total / 4;                       // a slab is written and re-read per split: >= 4 stages each
    if (s < 1) s = 1;
    p.splits = s;
    p.slab_floats = (size_t)s * ((size_t)Cout * Cin * tk::KK + Cout);
    return p;
}

// K = 15, pad = 7, C_out a multiple of 32
bool wgrad_bf16_tk_supported(int Cin, int Cout, int K, int pad) {
    (void)Cin;
    return K == tk::KK && pad == tk::PAD && Cout % 32 == 0;
}

int wgrad_bf16_tk_dy_stride(int Lo) { return cdiv(Lo, tk::TT) * tk::TT; }

size_t wgrad_bf16_tk_ws_floats(int N, int Cin, int Cout, int L, int K, int pad) {
    return tk_plan(N, Cin, Cout, L + 2 * pad - K + 1).slab_floats + 16;
}

int wgrad_bf16_tk(const void *dy_bf16, int ldy, const void *x, int x_is_bf16, int ldx, float *dw, float *db,
                  float *ws, int N, int Cin, int Cout, int L, int K, int pad, hipStream_t st) {
    const int Lo = L + 2 * pad - K + 1;
    const TkPlan p = tk_plan(N, Cin, Cout, Lo);
    if (!p.m_t) return fail(ECG_EINVAL, "conv1d_bwd_weight_bias_bf16_ncl: C_out=%d is not a multiple of 32", Cout);
    const int R = Cin * K;
    const u16 *dy = static_cast<const u16 *>(dy_bf16);
    dim3 grid((unsigned)(cdiv(Cin, p.r_t / 16) * (Cout / p.m_t) * p.splits)), block(512);
#define ECG_TK(MT, RT, WM, WR, WK, XF) \
    hipLaunchKernelGGL((tk::conv1d_wgrad_bf16_tk_kernel<MT, RT, WM, WR, WK, XF>), grid, block, 0, st, dy, x, ws, N, Cin, Cout, L, \
                       ldy, ldx, p.ntt, p.splits)
    if (p.m_t == 128) { if (x_is_bf16) ECG_TK(128, ECG_TK_RT128, 2, 4, 1, false); else ECG_TK(128, ECG_TK_RT128, 2, 4, 1, true); }
    else if (p.m_t == 64) { if (x_is_bf16) ECG_TK(64, 512, 1, 8, 1, false); else ECG_TK(64, 512, 1, 8, 1, true); }
    else { if (x_is_bf16) ECG_TK(32, 192, 1, 2, 4, false); else ECG_TK(32, 192, 1, 2, 4, true); }
#undef ECG_TK
    int rc = check_launch("conv1d_wgrad_bf16_tk_kernel");
    if (rc) return rc;
    return wgrad_reduce(ws, dw, db, (size_t)Cout * R, Cout, p.splits, st);
}

}  // namespace ecg

using namespace ecg;

ECG_API int ecg_conv1d_bf16_tk_supported(int C_in, int C_out, int K, int pad) {
    return wgrad_bf16_tk_supported(C_in, C_out, K, pad) ? 1 : 0;
}

ECG_API int ecg_conv1d_bf16_tk_dy_stride(int Lo) { return Lo > 0 ? wgrad_bf16_tk_dy_stride(Lo) : 0; }

ECG_API size_t ecg_conv1d_bwd_weight_bf16_ncl_ws_floats(int N, int C_in, int C_out, int L, int K, int pad) {
    if (!wgrad_bf16_tk_supported(C_in, C_out, K, pad) || N <= 0 || L + 2 * pad - K + 1 <= 0) return 0;
    return wgrad_bf16_tk_ws_floats(N, C_in, C_out, L, K, pad);
}

ECG_API int ecg_conv1d_bwd_weight_bias_bf16_ncl(const void *dy_bf16, int ldy, const void *x, int x_is_bf16, int ldx,
                                                float *dw, float *db, float *ws, int N, int C_in, int C_out, int L,
                                                int K, int pad, ecg_stream_t stream) {
    ECG_REQUIRE(N > 0 && C_in > 0 && C_out > 0 && L >= 16, "conv1d_bwd_weight_bias_bf16_ncl: N=%d C_in=%d C_out=%d L=%d", N,
                C_in, C_out, L);
    ECG_REQUIRE(wgrad_bf16_tk_supported(C_in, C_out, K, pad),
                "conv1d_bwd_weight_bias_bf16_ncl: needs K == 15, pad == 7, C_out %% 32 == 0");
    ECG_REQUIRE(dy_bf16 && x && dw && ws, "conv1d_bwd_weight_bias_bf16_ncl: null pointer");
    const int Lo = L + 2 * pad - K + 1;
    ECG_REQUIRE(ldy % 128 == 0 && ldy >= wgrad_bf16_tk_dy_stride(Lo),
                "conv1d_bwd_weight_bias_bf16_ncl: dY rows must be zero-filled to a stride of %d (got %d)",
                wgrad_bf16_tk_dy_stride(Lo), ldy);
    if (x_is_bf16)
        ECG_REQUIRE(ldx % 8 == 0 && ldx >= L, "conv1d_bwd_weight_bias_bf16_ncl: bf16 x needs a row stride >= L that is a "
                    "multiple of 8 (rows zero-filled past L); got %d", ldx);
    else
        ECG_REQUIRE(L % 8 == 0 && ldx % 4 == 0 && ldx >= L, "conv1d_bwd_weight_bias_bf16_ncl: fp32 x needs L %% 8 == 0 and "
                    "a row stride that is a multiple of 4 (L=%d, ldx=%d)", L, ldx);
    ECG_REQUIRE(((reinterpret_cast<uintptr_t>(dy_bf16) | reinterpret_cast<uintptr_t>(x)) & 15) == 0,
                "conv1d_bwd_weight_bias_bf16_ncl: operands must be 16-byte aligned");
    ECG_REQUIRE((long long)C_out * ldy * 2 < (1LL << 31) && (long long)N * C_in * ldx < (1LL << 31) &&
                (long long)N * C_out * ldy < (1LL << 31), "conv1d_bwd_weight_bias_bf16_ncl: tensor too large for 32-bit offsets");
    return wgrad_bf16_tk(dy_bf16, ldy, x, x_is_bf16, ldx, dw, db, ws, N, C_in, C_out, L, K, pad, as_stream(stream));
}
