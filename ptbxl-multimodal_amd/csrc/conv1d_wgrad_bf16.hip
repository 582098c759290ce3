// conv1d_wgrad_bf16.hip — mixed-precision weight gradient (the opt-in path of BASELINE.json config 5):
// bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 dW / db out.
//
//   dW[co][ci][k] = sum_n sum_t dY[n][co][t] * X[n][ci][t + k - pad]
//
// The reduction runs over (n, t).  A bf16 MFMA consumes 16 reduction indices at once, 8 per lane as ONE
// 16-byte operand — so the 16 indices must be contiguous in memory for BOTH operands.  Along t that fails
// for X (tap k shifts the window by one 2-byte element: unaligned).  Along n it works: the MFMA's K
// dimension is a group of 16 SAMPLES at a fixed time step,
//   A[co][j] = dY[16g + j][co][t],      B[j][(ci,k)] = X[16g + j][ci][t + k - pad],
// and with both tensors re-laid as [sample group][channel][time][16 samples] (bf16, "n16") a tap shift
// moves by whole 32-byte granules.  Two launches:
//   pack_n16_kernel    fp32 [N][C][ld] -> bf16 [G][C][P][16], round to nearest even, zero-filled past N
//                      and outside the row (for X that bakes the conv's zero padding into the layout)
//   conv1d_wgrad_bf16_kernel   workgroup tile M_T (co) x R_T (r = ci*15 + k); a stage = (sample group,
//                      16 time steps); both operand tiles go global -> LDS by DMA (no registers, no
//                      masks — the layouts already contain every zero; measured, NOT adopted: a 128 x 256
//                      eight-wave tile with a 2- or 3-image DMA ring and the pieces issued between the MFMAs ran
//                      at the same 190-200 us on block 3 of 12x5000 as this kernel); per time step one conflict-free
//                      ds_read_b128 per fragment (the two 16-byte halves of a granule are XOR-swizzled by
//                      bit 3 of the granule index; the x rows are 31 granules apart so that the column
//                      index r maps to consecutive granules mod 16).  Split over stages into slabs that
//                      wgrad_reduce_kernel sums in fixed order, like the fp32 kernel.
// Replaces autograd's conv weight-gradient (reference src/models/ecg_cnn.py:13 via loss.backward()).
#include "common.h"
#include <cstdlib>

namespace ecg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kKW = 15;
constexpr int kTW = 16;        // time steps per stage
constexpr int kXP = 31;        // x-tile granules per channel row: >= kTW + 14, == 15 (mod 16)

__device__ __forceinline__ int acc_row_w(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// s_waitcnt lgkmcnt(n), vmcnt / expcnt left at "no wait" (n compile-time after unrolling)
__device__ __forceinline__ void wait_lgkm_w(int n) {
    switch (n) {
        case 2: __builtin_amdgcn_s_waitcnt(0xC27F); break;
        case 3: __builtin_amdgcn_s_waitcnt(0xC37F); break;
        case 4: __builtin_amdgcn_s_waitcnt(0xC47F); break;
        case 5: __builtin_amdgcn_s_waitcnt(0xC57F); break;
        case 6: __builtin_amdgcn_s_waitcnt(0xC67F); break;
        default: break;
    }
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const u16 a = __builtin_bit_cast(u16, (__bf16)lo), b = __builtin_bit_cast(u16, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

// dst[g][c][p][j] = src[16g + j][c][p - shift]  (0 outside the row or past N).  grid = (ceil(P/256), C, G)
__global__ __launch_bounds__(256) void pack_n16_kernel(const float *__restrict__ src, u16 *__restrict__ dst,
                                                       int N, int C, int ld, int Lsrc, int P, int shift) {
    const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, g = blockIdx.z;
    if (p >= P) return;
    const int t = p - shift;
    const bool in_row = (t >= 0) && (t < Lsrc);
    const int tc = min(max(t, 0), Lsrc - 1);
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)                     // 16 unconditional, clamped loads in flight ...
        v[j] = src[((size_t)min(16 * g + j, N - 1) * C + c) * ld + tc];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 16; ++j)                     // ... zeroing applied afterwards
        v[j] = (in_row && 16 * g + j < N) ? v[j] : 0.f;
    u32x4 lo, hi;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        lo[q] = pack2(v[2 * q], v[2 * q + 1]);
        hi[q] = pack2(v[8 + 2 * q], v[8 + 2 * q + 1]);
    }
    u32x4 *o = reinterpret_cast<u32x4 *>(dst + (((size_t)g * C + c) * P + p) * 16);
    o[0] = lo;
    o[1] = hi;
}

// grid = (R tiles * C_out tiles * S) with the XCD-chunked order of conv1d_mfma.hip's kernels.
template <int M_T, int R_T, int WM, int WR>
__global__ __launch_bounds__(256, 3) void conv1d_wgrad_bf16_kernel(
    const u16 *__restrict__ dyb, const u16 *__restrict__ xb, float *__restrict__ slab, int G, int Cin,
    int Cout, int PA, int PX, int ntt, int S) {
    static_assert(WM * WR == 4, "4 waves per workgroup");
    constexpr int KK = kKW;
    constexpr int MC = M_T / WM / 32, MR = R_T / WR / 32;
    constexpr int NCI = (R_T + KK - 2) / KK + 1;
    constexpr int ASLOTS = kTW * M_T * 2;                  // 16-byte slots of the dY image
    constexpr int BSLOTS = NCI * kXP * 2;
    constexpr int ADMA = ASLOTS / 64, BDMA = (BSLOTS + 63) / 64;
    constexpr int APW = ADMA / 4, BPW = (BDMA + 3) / 4;    // DMA instructions per wave per stage
    static_assert(ADMA % 4 == 0, "dY image must split evenly over the four waves");
    constexpr int IMGB = (ASLOTS + BDMA * 64) * 16;

    __shared__ __attribute__((aligned(1024))) unsigned char lds[IMGB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int R = Cin * KK;
    const int RT = (R + R_T - 1) / R_T, CT = Cout / M_T;
    // same XCD-aware order as the fp32 kernels: the R tiles of one (C_out tile, split) share a dY slice
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_r = tile % RT, tile_cs = tile / RT;
    const int r0 = tile_r * R_T, co0 = (tile_cs % CT) * M_T, s = tile_cs / CT;
    const int wr = wave % WR, wm = wave / WR;
    const int wm0 = wm * (M_T / WM), wr0 = wr * (R_T / WR);
    const int ci_base = r0 / KK;
    const int total = G * ntt;
    const int st_begin = (int)((long long)total * s / S), st_end = (int)((long long)total * (s + 1) / S);

    // B fragments: granule index (before the time offset) of column r = r0 + wr0 + 32j + l31
    int qb0[MR];
#pragma unroll
    for (int j = 0; j < MR; ++j) {
        int r = r0 + wr0 + 32 * j + l31;
        if (r >= R) r = R - 1;                 // clamped columns compute garbage that is never stored
        const int ci = r / KK;
        qb0[j] = (ci - ci_base) * kXP + (r - ci * KK);
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
    const bool want_bias = (tile_r == 0) && (wr == 0);

    // ---- DMA geometry: LDS slot -> element offset from the stage base (loop-invariant) -------------
    // A image: [co][2*kTW 16-byte slots], slot (2t + half) of row co stored at slot ^ (co & 15): a DMA piece is two
    // whole channel rows, i.e. two CONTIGUOUS 512-byte runs of the n16 tensor (a [t][co] image would gather 32 bytes
    // from each of 32 rows 20 KB apart: a quarter of every line it touches), and the fragment read below — 32 lanes,
    // 32 rows, one time step — still covers all banks once per 16-lane group.
    // B image: slot = 2*granule + (half ^ bit3(granule)), granule = ci_local*kXP + pos (contiguous per channel row).
    int aoff[APW], boff[BPW];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int sl = (j * 4 + wave) * 64 + lane;
        const int co = sl / (2 * kTW), ls = sl - co * (2 * kTW);     // LDS slot ls of row co ...
        const int gs = ls ^ (co & 15);                               // ... holds global slot gs = 2t + half
        aoff[j] = (co * PA + (gs >> 1)) * 16 + 8 * (gs & 1);
    }
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        const int sl = min((j * 4 + wave) * 64 + lane, BSLOTS - 1);       // the tail of the last piece re-reads a valid slot
        const int qb = sl >> 1, h = (sl & 1) ^ ((qb >> 3) & 1);
        const int cl = qb / kXP, pos = qb - cl * kXP;
        boff[j] = (min(ci_base + cl, Cin - 1) * PX + pos) * 16 + 8 * h;
    }
    unsigned char *aimg = lds, *bimg = lds + ASLOTS * 16;

    for (int st = st_begin; st < st_end; ++st) {
        const int g = st / ntt, t0 = (st - g * ntt) * kTW;
        const u16 *abase = dyb + (((size_t)g * Cout + co0) * PA + t0) * 16;
        const u16 *bbase = xb + ((size_t)g * Cin * PX + t0) * 16;
#pragma unroll
        for (int j = 0; j < APW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(abase + aoff[j]),
                                             (__attribute__((address_space(3))) void *)(aimg + (j * 4 + wave) * 1024),
                                             16, 0, 0);
#pragma unroll
        for (int j = 0; j < BPW; ++j)
            if ((j * 4 + wave) < BDMA)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bbase + boff[j]),
                                                 (__attribute__((address_space(3))) void *)(bimg + (j * 4 + wave) * 1024),
                                                 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this wave's DMA pieces have landed in LDS ...
        __syncthreads();                        // ... and so have everybody else's

        auto ld = [&](int t, bf16x8 *a, bf16x8 *b) {
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                const int co = wm0 + 32 * i + l31;
                a[i] = *reinterpret_cast<const bf16x8 *>(aimg + (co * (2 * kTW) + ((2 * t + half) ^ (co & 15))) * 16);
            }
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                const int qb = qb0[j] + t;
                b[j] = *reinterpret_cast<const bf16x8 *>(bimg + (2 * qb + (half ^ ((qb >> 3) & 1))) * 16);
            }
        };
        bf16x8 a_c[MC], b_c[MR], a_n[MC], b_n[MR];
        ld(0, a_c, b_c);
#pragma unroll
        for (int t = 0; t < kTW; ++t) {
            ld(t + 1 < kTW ? t + 1 : 0, a_n, b_n);
            wait_lgkm_w(MC + MR);        // the MFMAs need the fragments read one step ago, not the ones just issued
            __builtin_amdgcn_sched_barrier(0);
            if (want_bias) {                      // db rides on the A fragments (bf16-rounded dY, fp32 sum)
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[i] += (float)a_c[i][e];
            }
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < MR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c[i], b_c[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MC; ++i) a_c[i] = a_n[i];
#pragma unroll
            for (int j = 0; j < MR; ++j) b_c[j] = b_n[j];
        }
        __syncthreads();            // everybody is done reading before the next stage's DMA lands
    }

    if (want_bias) {
#pragma unroll
        for (int i = 0; i < MC; ++i) bsum[i] += __shfl_xor(bsum[i], 32, 64);
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
                    const int co = co0 + wm0 + 32 * i + acc_row_w(q, half);
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

// ---------------------------------------------------------------------------------------------------------
// Round 3: the same reduction on LARGE tiles with a double-buffered stage pipeline (one workgroup of eight waves per CU).
// The kernel above (64 x 128 tiles, one LDS image, three workgroups per CU covering for each other) sits at 0.24-0.34 of
// the bf16 peak: every (co, t) granule of dY is pulled L2 -> LDS once per 128 columns, and a workgroup's DMA round trip is
// exposed at every stage.  Here a tile is M_T x 512 columns (the dY tile is re-used 4x more), a stage is (sample group,
// 8 time steps), the next stage's two operand tiles stream into the other LDS image while this one is multiplied (asm
// LDS-DMA: see conv1d_bf16_ring.hip for why not the builtin), one vmcnt(0) + barrier per stage — the pieces were
// issued in the first half of the stage — and the step loop carries explicit counted lgkmcnt waits.
// Images: dY [M_T][16 slots] (row co: slot 2t + half at slot ^ (co & 15)); x [NCI][kXP] granules as above.
template <int M_T, int WM, int WR>
__global__ __launch_bounds__(512) void conv1d_wgrad_bf16_ring_kernel(
    const u16 *__restrict__ dyb, const u16 *__restrict__ xb, float *__restrict__ slab, int G, int Cin,
    int Cout, int PA, int PX, int S) {
    static_assert(WM * WR == 8, "8 waves per workgroup");
    constexpr int KK = kKW, R_T = 512, TS = 8;
    constexpr int MC = M_T / WM / 32, MR = R_T / WR / 32;
    static_assert(MC == 2 && (MR == 4 || MR == 2), "wave tile 64 x 128 or 64 x 64");
    constexpr int NCI = (R_T + KK - 2) / KK + 1;
    constexpr int ASLOTS = TS * M_T * 2;                   // 16-byte slots of the dY image
    constexpr int BSLOTS = NCI * kXP * 2;
    constexpr int ADMA = ASLOTS / 64, BDMA = (BSLOTS + 63) / 64;
    constexpr int APW = ADMA / 8, BPW = (BDMA + 7) / 8;    // DMA pieces per wave per stage
    static_assert(ADMA % 8 == 0, "dY image must split evenly over the eight waves");
    // The x image spans 16 time steps (+ 14 of halo = kXP - 1): it serves TWO stages and is re-streamed only when the next
    // stage starts a new 16-step block — per stage the dY tile is 32 KB (M_T = 128) and the x tile 35 KB of which 9 KB are
    // new, so streaming it every stage would double the LDS-DMA work.  Two dY images and two x images, flipped separately.
    constexpr int AIMG = ASLOTS * 16, BIMG = BDMA * 64 * 16;
    static_assert(2 * (AIMG + BIMG) <= 160 * 1024, "LDS");
    static_assert(APW + BPW <= 2 * TS, "at most two DMA pieces per time step");
    static_assert(kTW == 2 * TS && kXP >= kTW + KK - 1, "an x image covers two stages");

    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (AIMG + BIMG)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int R = Cin * KK;
    const int RT = (R + R_T - 1) / R_T, CT = Cout / M_T;
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_r = tile % RT, tile_cs = tile / RT;
    const int r0 = tile_r * R_T, co0 = (tile_cs % CT) * M_T, s = tile_cs / CT;
    const int wr = wave % WR, wm = wave / WR;
    const int wm0 = wm * (M_T / WM), wr0 = wr * (R_T / WR);
    const int ci_base = r0 / KK;
    const int ntt = PA / TS;
    const int total = G * ntt;
    const int st_begin = (int)((long long)total * s / S), st_end = (int)((long long)total * (s + 1) / S);

    // fragment addressing: byte offsets inside an image
    int aoffl[MC], boffl[MR];
#pragma unroll
    for (int i = 0; i < MC; ++i) aoffl[i] = (wm0 + 32 * i + l31) * (2 * TS) * 16;        // + ((2t + half) ^ (co & 15)) * 16
    const int asw = l31 & 15;                               // (wm0 + 32 i) is a multiple of 16: co & 15 = l31 & 15
#pragma unroll
    for (int j = 0; j < MR; ++j) {
        int r = r0 + wr0 + 32 * j + l31;
        if (r >= R) r = R - 1;                              // clamped columns compute garbage that is never stored
        const int ci = r / KK;
        boffl[j] = (ci - ci_base) * kXP + (r - ci * KK);    // granule index before the time offset
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
    // db rides on the dY fragments of the workgroups of column tile 0; the WR waves that hold the same channel rows take
    // every WR-th time step each (on two of eight waves the unpack + add work delayed the whole workgroup at every barrier)
    const bool want_bias = (tile_r == 0);

    // ---- DMA geometry (loop-invariant per lane) ---------------------------------------------------------
    int aoff[APW], brow[BPW], bpos[BPW], bhalf[BPW];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int sl = (j * 8 + wave) * 64 + lane;
        const int co = sl / (2 * TS), ls = sl - co * (2 * TS);
        const int gs = ls ^ (co & 15);                      // LDS slot ls of row co holds global slot 2t + half
        aoff[j] = ((co * PA + (gs >> 1)) * 16 + 8 * (gs & 1)) * 2;       // bytes
    }
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        const int sl = min((j * 8 + wave) * 64 + lane, BSLOTS - 1);
        // x image WITHOUT the half swap of the kernel above: slot = 2 * granule + half.  Two granules 8 apart then share
        // their banks (2-way conflict on the x fragment reads, the LDS is ~20 % busy in this kernel), but a fragment's
        // address becomes base + 32 t: an instruction offset instead of four VALU instructions per read — with two waves per
        // SIMD the loop was bound by vector ISSUE (5 VALU per MFMA), not by the matrix pipe.
        const int qb = sl >> 1, h = sl & 1;
        const int cl = qb / kXP;
        brow[j] = min(ci_base + cl, Cin - 1) * PX;
        bpos[j] = qb - cl * kXP;
        bhalf[j] = 16 * h;                                  // bytes
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    // LDS-DMA piece: wave-uniform 64-bit base in SGPRs + a per-lane 32-bit BYTE offset (no 64-bit VALU address math)
    auto glds = [&](const u16 *base, unsigned voff, unsigned dst_off) __attribute__((always_inline)) {
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + dst_off));   // wave-uniform by construction
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
    };
    // stage coordinates of the stage whose tiles are DMA'd next
    int dg = st_begin / ntt, dt0 = (st_begin - dg * ntt) * TS;
    // piece j (0 .. APW + BPW - 1) of stage (dg, dt0): dY pieces into dY image `aimg`; x pieces — only when the stage opens a
    // new 16-step block (uniform) — into x image `bimg`
    auto dma_piece = [&](int j, int aimg, int bimg) __attribute__((always_inline)) {
        if (j < APW) {
            const u16 *abase = dyb + (((size_t)dg * Cout + co0) * PA + dt0) * 16;          // uniform
            glds(abase, (unsigned)aoff[j], (unsigned)(aimg * AIMG + (j * 8 + wave) * 1024));
        } else {
            const int jb = j - APW;
            if ((dt0 & (kTW - 1)) == 0 && jb * 8 + wave < BDMA) {
                // granules past the end of the x row (only reached by the slots nobody reads) are clamped into it
                const u16 *bbase = xb + (size_t)dg * Cin * PX * 16;                        // uniform
                const unsigned voff = (unsigned)(brow[jb] + min(dt0 + bpos[jb], PX - 1)) * 32u + (unsigned)bhalf[jb];
                glds(bbase, voff, (unsigned)(2 * AIMG + bimg * BIMG + (jb * 8 + wave) * 1024));
            }
        }
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {
        dt0 += TS;
        if (dt0 >= PA) { dt0 = 0; ++dg; }
        if (dg >= G) { dg = G - 1; dt0 = PA - TS; }          // past the end: restage a valid tile nobody reads
    };

    int bcur = 0;                                           // x image that holds the block of the stage being multiplied
    int ct0 = dt0;                                          // first time step of the stage being multiplied
    if (st_begin < st_end) {
        // the first stage may start in the middle of a 16-step block: stream the x block from its start
        const int keep = dt0;
        dt0 &= ~(kTW - 1);
#pragma unroll
        for (int j = APW; j < APW + BPW; ++j) dma_piece(j, 0, 0);
        dt0 = keep;
#pragma unroll
        for (int j = 0; j < APW; ++j) dma_piece(j, 0, 0);
        dma_advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int st = st_begin; st < st_end; ++st) {
        const int img = (st - st_begin) & 1;
        const bool new_block = (dt0 & (kTW - 1)) == 0;      // the NEXT stage opens a new x block (uniform)
        const int bnext = new_block ? bcur ^ 1 : bcur;
        const int abase_l = img * AIMG, bbase_l = 2 * AIMG + bcur * BIMG;
        const int tb = ct0 & TS;                            // second stage of a block: its granules start 8 further in
        // (the per-lane terms are made opaque where they are used: left alone, hipcc precomputes one address register per
        // (time step, fragment) — 48 of them — and spills)
        auto ld_a = [&](int t, int i) __attribute__((always_inline)) {
            int sw = asw;
            asm volatile("" : "+v"(sw));
            return *reinterpret_cast<const bf16x8 *>(&lds[abase_l + aoffl[0] + (((2 * t + half) ^ sw) << 4) + i * (32 * 2 * TS * 16)]);
        };
        int bl[MR];                                         // this stage's x fragment bases: + 32 t is an instruction offset
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            bl[j] = bbase_l + (boffl[j] + tb) * 32 + half * 16;
            asm volatile("" : "+v"(bl[j]));
        }
        auto ld_b = [&](int t, int j) __attribute__((always_inline)) {
            return *reinterpret_cast<const bf16x8 *>(&lds[bl[j] + 32 * t]);
        };
        bf16x8 a_c[MC], a_n[MC], b_c[MR];
#pragma unroll
        for (int i = 0; i < MC; ++i) a_c[i] = ld_a(0, i);
#pragma unroll
        for (int j = 0; j < MR; ++j) b_c[j] = ld_b(0, j);
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            if (2 * t < APW + BPW) dma_piece(2 * t, img ^ 1, bnext);            // the next stage's tiles, two pieces per time step
            if (2 * t + 1 < APW + BPW) dma_piece(2 * t + 1, img ^ 1, bnext);    // (all issued in the first half of the stage)
            const int tn = t + 1 < TS ? t + 1 : TS - 1;      // (the last step re-reads its own fragments: unused)
#pragma unroll
            for (int i = 0; i < MC; ++i) a_n[i] = ld_a(tn, i);
            if (want_bias && (t % WR) == wr) {               // (bf16-rounded dY, fp32 sum)
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[i] += (float)a_c[i][e];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                // issued since fragment j of this step was read: the other MR - 1 x fragments and the MC dY fragments
                if (MR == 4) __builtin_amdgcn_s_waitcnt(0xC07F | (5 << 8)); else __builtin_amdgcn_s_waitcnt(0xC07F | (3 << 8));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MC; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c[i], b_c[j], acc[i][j], 0, 0, 0);
                b_c[j] = ld_b(tn, j);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MC; ++i) a_c[i] = a_n[i];
        }
        ct0 = dt0;
        bcur = bnext;
        dma_advance();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this wave's pieces of the next stage have landed
        __syncthreads();                                                 // ... everybody's; and this image is free again
    }

    if (want_bias) {                                        // uniform per workgroup; the images are dead (last barrier of the loop)
        float *bred = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            bsum[i] += __shfl_xor(bsum[i], 32, 64);
            if (half == 0) bred[wave * (MC * 32) + 32 * i + l31] = bsum[i];
        }
        __syncthreads();
        if (wr == 0) {
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                float t = 0.f;
                for (int w = 0; w < WR; ++w) t += bred[(wm * WR + w) * (MC * 32) + 32 * i + l31];     // fixed order
                bsum[i] = t;
            }
        }
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
                    const int co = co0 + wm0 + 32 * i + acc_row_w(q, half);
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

// conv1d_direct.hip: dw[i] = sum_s slab[s][i] in fixed order
int wgrad_reduce(const float *ws, float *dw, float *db, size_t wslab, int Cout, int S, hipStream_t st);

struct WgBf16Plan { int G, PA, PX, ntt, splits, ring_mt; size_t dyb_elems, xb_elems, slab_floats; };

#ifndef ECG_WGB_RING
#define ECG_WGB_RING 1       // compile-time A/B knob (make VARIANT=x EXTRA="-DECG_WGB_RING=0"): no environment is read
#endif
static constexpr int wgb_ring_enabled() { return ECG_WGB_RING; }

static WgBf16Plan wgrad_bf16_plan(int N, int Cin, int Cout, int Lo) {
    WgBf16Plan p;
    p.G = cdiv(N, 16);
    p.ntt = cdiv(Lo, kTW);
    p.PA = p.ntt * kTW;                  // dY rows padded to whole stages (zeros)
    p.PX = p.PA + 16;                    // x rows: conv padding baked in, +kXP-1 granules readable past the last stage
    // the ring kernel (512-column tiles, one eight-wave workgroup per CU) for layers with at least one full column tile's
    // worth of work per output tile; stages of 8 time steps
    p.ring_mt = 0;
    if (wgb_ring_enabled() && Cout % 64 == 0 && Cin * kKW >= 448 && (long long)p.G * (p.PA / 8) >= 64) {
        p.ring_mt = Cout % 128 == 0 ? 128 : 64;
        const int tiles = cdiv(Cin * kKW, 512) * (Cout / p.ring_mt);
        int s = 256 / tiles;
        const int total = p.G * (p.PA / 8);
        if (s > total / 8) s = total / 8;
        if (s < 1) s = 1;
        p.splits = s;
        p.dyb_elems = (size_t)p.G * Cout * p.PA * 16;
        p.xb_elems = (size_t)p.G * Cin * p.PX * 16;
        p.slab_floats = (size_t)s * ((size_t)Cout * Cin * kKW + Cout);
        return p;
    }
    const int m_t = Cout % 64 == 0 ? 64 : 32, r_t = Cout % 64 == 0 ? 128 : 256;
    const int tiles = cdiv(Cin * kKW, r_t) * (Cout / m_t);
    int s = 768 / tiles;                 // 3 resident workgroups per CU ...
    const int total = p.G * p.ntt;
    if (s > total / 8) s = total / 8;    // ... but at least 8 stages per workgroup: a slab is written and re-read per split
    if (s < 1) s = 1;
    p.splits = s;
    p.dyb_elems = (size_t)p.G * Cout * p.PA * 16;
    p.xb_elems = (size_t)p.G * Cin * p.PX * 16;
    p.slab_floats = (size_t)s * ((size_t)Cout * Cin * kKW + Cout);
    return p;
}

bool wgrad_bf16_supported(int Cin, int Cout, int K, int pad) {
    (void)Cin;
    return K == kKW && pad == kKW / 2 && Cout % 32 == 0;
}

// workspace: slabs (fp32) | dY in n16 layout (bf16) | x in n16 layout (bf16)
size_t wgrad_bf16_ws_floats(int N, int Cin, int Cout, int L, int K, int pad) {
    const WgBf16Plan p = wgrad_bf16_plan(N, Cin, Cout, L + 2 * pad - K + 1);
    return p.slab_floats + (p.dyb_elems + p.xb_elems + 1) / 2 + 16;
}

// the MFMA kernel + the slab reduce on operands that are already in the n16 layout
static int wgrad_bf16_run(const u16 *dyb, const u16 *xb, float *dw, float *db, float *slab, const WgBf16Plan &p,
                          int Cin, int Cout, int K, hipStream_t st) {
    const int R = Cin * K;
    if (p.ring_mt) {
        dim3 grid((unsigned)(cdiv(R, 512) * (Cout / p.ring_mt) * p.splits)), block(512);
        if (p.ring_mt == 128)
            hipLaunchKernelGGL((conv1d_wgrad_bf16_ring_kernel<128, 2, 4>), grid, block, 0, st, dyb, xb, slab, p.G, Cin, Cout,
                               p.PA, p.PX, p.splits);
        else
            hipLaunchKernelGGL((conv1d_wgrad_bf16_ring_kernel<64, 1, 8>), grid, block, 0, st, dyb, xb, slab, p.G, Cin, Cout,
                               p.PA, p.PX, p.splits);
        int rc = check_launch("conv1d_wgrad_bf16_ring_kernel");
        if (rc) return rc;
        return wgrad_reduce(slab, dw, db, (size_t)Cout * R, Cout, p.splits, st);
    }
    dim3 block(256);
    if (Cout % 64 == 0) {
        dim3 grid((unsigned)(cdiv(R, 128) * (Cout / 64) * p.splits));
        hipLaunchKernelGGL((conv1d_wgrad_bf16_kernel<64, 128, 2, 2>), grid, block, 0, st, dyb, xb, slab, p.G, Cin,
                           Cout, p.PA, p.PX, p.ntt, p.splits);
    } else {                               // C_out = 32 (block 0): one 32-channel row of wide column tiles
        dim3 grid((unsigned)(cdiv(R, 256) * (Cout / 32) * p.splits));
        hipLaunchKernelGGL((conv1d_wgrad_bf16_kernel<32, 256, 1, 4>), grid, block, 0, st, dyb, xb, slab, p.G, Cin,
                           Cout, p.PA, p.PX, p.ntt, p.splits);
    }
    int rc = check_launch("conv1d_wgrad_bf16_kernel");
    if (rc) return rc;
    return wgrad_reduce(slab, dw, db, (size_t)Cout * R, Cout, p.splits, st);
}

int wgrad_bf16(const float *dy, int ldy, const float *x, float *dw, float *db, float *ws, int N, int Cin,
               int Cout, int L, int K, int pad, hipStream_t st) {
    const int Lo = L + 2 * pad - K + 1;
    const WgBf16Plan p = wgrad_bf16_plan(N, Cin, Cout, Lo);
    float *slab = ws;
    u16 *dyb = reinterpret_cast<u16 *>(ws + ((p.slab_floats + 3) / 4) * 4);      // 16-byte aligned
    u16 *xb = dyb + p.dyb_elems;
    hipLaunchKernelGGL(pack_n16_kernel, dim3(cdiv(p.PA, 256), Cout, p.G), dim3(256), 0, st, dy, dyb, N, Cout, ldy,
                       Lo, p.PA, 0);
    hipLaunchKernelGGL(pack_n16_kernel, dim3(cdiv(p.PX, 256), Cin, p.G), dim3(256), 0, st, x, xb, N, Cin, L, L,
                       p.PX, pad);
    int rc = check_launch("pack_n16_kernel");
    if (rc) return rc;
    return wgrad_bf16_run(dyb, xb, dw, db, slab, p, Cin, Cout, K, st);
}

// n16 geometry of a layer, for producers that write the operands in that layout themselves
void wgrad_bf16_positions(int L, int K, int pad, int *PA, int *PX) {
    const WgBf16Plan p = wgrad_bf16_plan(16, 4, 32, L + 2 * pad - K + 1);
    *PA = p.PA; *PX = p.PX;
}

size_t wgrad_bf16_packed_ws_floats(int N, int Cin, int Cout, int L, int K, int pad) {
    return wgrad_bf16_plan(N, Cin, Cout, L + 2 * pad - K + 1).slab_floats + 16;
}

int wgrad_bf16_packed(const void *dyb, const void *xb, float *dw, float *db, float *ws, int N, int Cin, int Cout,
                      int L, int K, int pad, hipStream_t st) {
    const WgBf16Plan p = wgrad_bf16_plan(N, Cin, Cout, L + 2 * pad - K + 1);
    return wgrad_bf16_run(static_cast<const u16 *>(dyb), static_cast<const u16 *>(xb), dw, db, ws, p, Cin, Cout, K, st);
}

int pack_n16(const float *src, void *dst, int N, int C, int ld, int Lsrc, int P, int shift, hipStream_t st) {
    hipLaunchKernelGGL(pack_n16_kernel, dim3(cdiv(P, 256), C, cdiv(N, 16)), dim3(256), 0, st, src,
                       static_cast<u16 *>(dst), N, C, ld, Lsrc, P, shift);
    return check_launch("pack_n16_kernel");
}

}  // namespace ecg
