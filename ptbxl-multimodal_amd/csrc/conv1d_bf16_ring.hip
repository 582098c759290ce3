// conv1d_bf16_ring.hip — mixed-precision Conv1d forward / input-grad for LONG rows with bf16 activations on both
// sides (BASELINE.json config 5: 12x5000, blocks 1-3 forward and every input gradient of the bf16-storage train step).
// Round-3 replacement of conv1d_mfma_bf16.hip's kernel for those shapes; that kernel keeps every other shape (fp32
// input or output, short rows).  Same arithmetic: bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate,
//   D[t][co] += sum_{ci in chunk of 16} X[t + tap][ci] * W[tap][ci][co]          (one MFMA per tap and 32x32 sub-tile)
// What changed, and why (the round-2 kernel sat at 0.10-0.39 of the bf16 MFMA peak):
//   * ACCUMULATORS TRANSPOSED: the x fragment is the MFMA's A operand (rows = time), the weight fragment its B operand
//     (columns = output channel).  A lane then owns ONE output channel and 16 time steps of it: the bias is a lane
//     constant (the accumulators start from it), BatchNorm statistics are plain per-lane adds with no cross-lane step
//     until the workgroup ends, and four consecutive time steps of a row sit in four consecutive registers -> ONE
//     8-byte store per lane instead of four 2-byte ones.  The round-2 epilogue (16-lane DPP row sums + 2-byte stores per
//     accumulator row) cost ~5 us per 128 x 256 tile, a third of the matrix time of block 3 and several times that of
//     the small layers.
//   * TILES 160 time steps per wave (5 MFMA rows): 640 / 1280 per workgroup.  Rows of 625 * 2^k (what 12x5000 pools
//     to) pad to 640 * 2^k: 2.4 % idle lanes instead of 23 % (625 -> 3 x 256).  A wave holds 2 x 5 (or 1 x 5)
//     accumulators: 7 fragment reads per 10 MFMAs instead of 4 per 4, and the weight bytes per MFMA fall 2.5x.
//   * WEIGHTS THROUGH A RING, not a double buffer: a slot is 8 KB = GT taps of [CO_T][16] bf16, one 1 KB LDS-DMA piece
//     per wave; five slots, four in flight.  A group start costs each wave one counted s_waitcnt vmcnt(N) (N = what it
//     issued after the piece it needs: never 0 in the loop), one s_barrier and one DMA issue — the round-2 kernel
//     drained to its x loads and restaged 61 KB behind one barrier per 16-channel chunk.  Layers whose whole weight
//     slice fits (<= 60 KB: blocks 1 forward / input-grad) keep it RESIDENT: no ring at all.
//   * x tile: register-staged as before (bf16 NCL rows -> [position][16 channels] LDS rows, halves swapped by bit 3 of
//     the row: conflict-free ds_read_b128 for any tap shift), loaded one chunk ahead at the first taps of a chunk and
//     committed at its last taps — load and use of every register stay inside one loop body.
// Parity: tests/test_gpu_ops.py::test_bf16_ring_* (exact on bf16-rounded operands vs the oracle; statistics; ragged
// rows; both weight modes), test_bf16_conv_full_size_config5_*.
// Replaces the ATen work behind ConvBlock.net[0] (reference src/models/ecg_cnn.py:13) and its input gradient.
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace ecg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// in-kernel stamps of the diagnostic build (make STAMP=1; tools/stamp_ring.py): s_memtime at start / prologue done / first
// tile's taps done / first tile's epilogue done / end of the workgroup
#ifdef ECG_STAMP
__device__ unsigned long long *g_stamps_r = nullptr;
#define ECG_STAMPR_AT(slot) do { if (g_stamps_r && threadIdx.x == 0) { \
    g_stamps_r[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    if ((slot) == 0) g_stamps_r[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime(); \
    if ((slot) == 4) g_stamps_r[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define ECG_STAMPR_AT(slot) do { } while (0)
#endif

namespace ring {

constexpr int KK = 15;           // taps
constexpr int CB = 16;           // input channels per MFMA (its K dimension)
constexpr int NW = 8;            // waves per workgroup
constexpr int NT = 64 * NW;
constexpr int NSLOT = 5;         // ring slots of 8 KB
constexpr int LA = 4;            // groups in flight ahead of the one being computed
constexpr int SLOTB = 8192;

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(((N >> 4) << 14) | 0x0F70 | (N & 15));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }
template <int N>
__device__ __forceinline__ void wait_lgkm() {           // lgkmcnt(N), vmcnt / expcnt left at "no wait"
    static_assert(N >= 0 && N < 16, "lgkmcnt is a 4-bit counter");
    __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));
}

// (The timing-only 16x16x32 what-if variant of this step — profiles/r03_mfma_shape_whatif.txt — lived here in round 3;
// it was removed from the product source once measured: 3-7 % at best against a certain spill.)
template <int PAR>
__device__ __forceinline__ f32x16 mfma_step(bf16x8 a, bf16x8 b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// x-load dwords issued in the 3*GT taps before tap u (taps with chunk-local index < XL issue one item = 4 dwords)
constexpr int x_dwords_before(int u, int GT, int XL) {
    int n = 0;
    for (int v = u - 3 * GT; v < u; ++v) {
        const int k = ((v % KK) + KK) % KK;
        if (k < XL) n += 4;
    }
    return n;
}

// ring mode: the tap (chunk-local) of chunk cb of a body at which its x commits are issued = the last group start <= 12
constexpr int commit_tap(int cb, int GT) {
    int best = -1;
    for (int k = 0; k <= 12; ++k)
        if ((cb * KK + k) % GT == 0) best = k;
    return best;
}

// CO_T output channels x (WT * MT * 32) time steps per workgroup; WCO x WT = 8 waves; wave tile (MT*32) x (MC*32).
// RES_CH = 0: weights through the ring; RES_CH > 0: the whole slice of <= RES_CH chunks resident in LDS.
// XF32: x is the fp32 NETWORK INPUT ([N][C_in][ldx] floats; block 0 of the model, C_in = 12 -> ONE chunk per tile) and is
// rounded to bf16 on its way into the LDS image.  RES_CH == 1 (one chunk per tile): the loop body is two TILES, each
// chunk followed by its epilogue.
// MT == 2 (512-step tiles, the fp32-input variant): 128 registers and ~70 KB of LDS, TWO workgroups per CU — a one-chunk
// tile is 75 MFMAs per wave against an epilogue of comparable length, and two independent workgroups fill each other's
// epilogues (in-kernel stamps with one 1280-step workgroup per CU: taps 5.4 us, epilogue 2.9 us per tile, matrix pipe
// idle in the latter).
template <int CO_T, int WCO, int WT, int MT, bool STATS, int RES_CH, bool XF32 = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(MT == 2 ? 4 : 2))) void conv1d_bf16_ring_kernel(
    const void *__restrict__ xv, const u16 *__restrict__ wb, const float *__restrict__ bias, u16 *__restrict__ y,
    float *__restrict__ partials, int Cin, int Cout, int L, int Lo, int pad, int tiles_t, int N, int G, int P_stride,
    int ldx, int ldyo) {
    static_assert(WCO * WT == NW, "eight waves");
    constexpr int MC = CO_T / WCO / 32;
    static_assert(MC == 1 || MC == 2, "one or two 32-channel groups per wave");
    constexpr int T_T = WT * MT * 32;
    constexpr int SPAN = T_T + KK - 1;
    constexpr int PPT = CO_T / 32;                       // 1 KB pieces per tap slice [CO_T][16] bf16
    constexpr bool RES = RES_CH > 0;
    constexpr bool ONECH = RES_CH == 1;
    static_assert(!XF32 || ONECH, "the fp32 input is the network input: one chunk");
    using XT = std::conditional_t<XF32, float, u16>;
    const XT *const x = static_cast<const XT *>(xv);
    constexpr int XR = XF32 ? 8 : 4;                     // registers per staged item
    constexpr int GT = RES ? 1 : NW / PPT;               // taps per ring group (8 pieces = one per wave)
    constexpr int BODY_CH = RES ? 2 : GT;                // chunks per unrolled loop body (GT * 15 taps = 15 groups)
    static_assert(RES || (GT * PPT == NW && (GT == 2 || GT == 4)), "ring groups of 2 or 4 taps");
    constexpr int WAREA = RES ? RES_CH * KK * CO_T * 32 : NSLOT * SLOTB;
    constexpr int NPAIR = SPAN / 2 + 1;                  // position pairs (2i-1, 2i)
    constexpr int XITEMS = NPAIR * 4;                    // (pair, quarter of 4 channels)
    constexpr int XL = (XITEMS + NT - 1) / NT;
    static_assert(XL <= 6, "x loads at taps 0..XL-1, commits at taps 13-XL..12");
    constexpr int KC0 = 13 - XL;                         // first commit tap
    constexpr int XB = ((SPAN * 32 + 16 + 15) / 16) * 16;   // x image + one dummy slot for items outside the tile
    constexpr int REDB = STATS ? NW * (CO_T / WCO) * 2 * 4 : 0;
    static_assert(REDB <= XB, "stat scratch aliases x image 0");
    constexpr int EPROW = 80;                            // bytes per epilogue patch row (64 of data)
    constexpr int EPB = (NW * 32 * EPROW <= XB) ? 0 : NW * 32 * EPROW;   // patches in the dead x image, or (short tiles) behind the images
    static_assert(WAREA + 2 * XB + EPB <= 160 * 1024, "LDS");

    __shared__ __attribute__((aligned(1024))) unsigned char lds[WAREA + 2 * XB + EPB];
    unsigned char *const xlds = lds + WAREA;

    ECG_STAMPR_AT(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int wg;
    {   // XCD-aware order: the C_out tiles of one tile range sit on one XCD and read the same x panels from its L2
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int CT = Cout / CO_T;
    const int tile_co = wg % CT, g = wg / CT;
    const int ntiles = N * tiles_t;
    const int q0 = (int)((long long)ntiles * g / G), q1 = (int)((long long)ntiles * (g + 1) / G);
    const int co0 = tile_co * CO_T;
    const int wco = (wave / WT) * (CO_T / WCO), wt = (wave % WT) * (MT * 32);
    const int nchunks = (Cin + CB - 1) / CB;
    const int KT = nchunks * KK;                         // taps of one tile's reduction
    const int total = (q1 - q0) * nchunks;               // flat chunks of this workgroup
    if (total <= 0) return;                              // uniform

    // ---- lane = output channel: the bias is a lane constant, added in the epilogue --------------------------
    // (accumulators that START from the bias would save that add, but a register splat is not rematerialisable: hipcc
    // then keeps the re-initialised accumulators of the next tile in scratch across the epilogue — 46 dependent scratch
    // reloads per tile, 30-60 us; zeros are immediates)
    float bia[MC], st_s[MC], st_q[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        bia[i] = bias ? bias[co0 + wco + 32 * i + l31] : 0.f;
        st_s[i] = 0.f; st_q[i] = 0.f;
    }
    f32x16 acc[MC][MT];
#pragma unroll
    for (int i = 0; i < MC; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- weight stream -------------------------------------------------------------------------------------
    // global slice of tap f (flat inside the tile's reduction): wb + f * Cout * 32 bytes, rows co0 .. co0 + CO_T
    // LDS row r of a slice holds global half h at half position h ^ bit3(r)  (swizzle on the SOURCE address)
    const size_t tap_bytes = (size_t)Cout * 32;
    // The LDS-DMA is issued from INLINE ASM: hipcc models __builtin_amdgcn_global_load_lds like a FLAT access, and then waits
    // vmcnt(0) / lgkmcnt(0) for every load and LDS read that was pending when one issued (39 + 4 full drains per 30 taps in
    // this loop).  An asm statement with no register result is invisible to that bookkeeping and safe (nothing to
    // protect but LDS, which the counted vmcnt + barrier at the group starts orders).  M0 is saved and restored inside
    // the statement (the compiler owns it).
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    // source = wave-uniform 64-bit base (SGPR pair: tap slice + row block) + this lane's 32-bit byte offset inside the piece
    const unsigned wlane = (unsigned)((lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4));
    const unsigned char *const wtile = reinterpret_cast<const unsigned char *>(wb) + (size_t)co0 * 32;
    auto dma_piece = [&](int f, int rb, unsigned dst_off) __attribute__((always_inline)) {   // rows rb*32 .. +31 of tap f -> 1 KB at lds + dst_off
        const unsigned long long s0 = (unsigned long long)(wtile + (size_t)f * tap_bytes + rb * 1024);   // uniform; made provably so
        const unsigned char *src = reinterpret_cast<const unsigned char *>(
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(s0 >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)s0));
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + dst_off));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(wlane), "s"(src), "s"(dst) : "memory");
    };
    int wtap = 0;                                        // ring: first tap (mod KT) of the NEXT group to issue
    auto dma_group = [&](int slot) __attribute__((always_inline)) {
        int f = wtap + wave / PPT;
        if (f >= KT) f -= KT;
        dma_piece(f, wave % PPT, slot * SLOTB + wave * 1024);
        wtap += GT;
        if (wtap >= KT) wtap -= KT;
    };

    // ---- x staging geometry (bf16 rows, position pairs) ---------------------------------------------------
    // item = (pair i -> positions 2i-1 and 2i, quarter q of 4 channels); kept PACKED (pair | q << 16 | live << 31): the
    // LDS destinations are recomputed at the XL commits of a chunk instead of living in 4 * XL registers
    unsigned xitem[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        const int it = tid + NT * j;
        const int itc = min(it, XITEMS - 1);
        const int q = itc / NPAIR, pi = itc - q * NPAIR;
        xitem[j] = (unsigned)pi | ((unsigned)q << 16) | (it < XITEMS ? 0x80000000u : 0u);
    }
    unsigned xreg[XL][XR];
    unsigned xok = 0, xok1 = 0;

    int cn = q0 / tiles_t, ctt = q0 - cn * tiles_t;               // compute stage: (sample, t tile)
    const size_t xstep_n = (size_t)Cin * ldx;
    const XT *xld = x + (size_t)cn * xstep_n;
    int ltt = ctt, lc = 0, lleft = total;
    auto ld_advance = [&]() __attribute__((always_inline)) {      // stays on the last chunk once everything is loaded
        if (--lleft > 0) {
            if (++lc == nchunks) {
                lc = 0;
                if (++ltt == tiles_t) { ltt = 0; xld += xstep_n; }
            }
        } else lleft = 1;
    };
    auto load_x = [&](int j) __attribute__((always_inline)) {
        unsigned item = xitem[j];
        asm volatile("" : "+v"(item));                   // opaque: decode here (a few VALU per chunk), do not hoist 8 registers per item
        const int pi = (int)(item & 0xFFFFu), q = (int)((item >> 16) & 3u);
        const int ci = lc * CB + 4 * q;
        const int sidx = ltt * T_T + 2 * pi - 1 - pad;   // even: pad is odd
        const int sc = min(max(sidx, 0), ldx - 2);
        const size_t off = (size_t)min(ci, Cin - 4) * ldx + sc;
        if constexpr (XF32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {                // two positions of channel ci + u: one aligned 8-byte load
                const uint2 v = *reinterpret_cast<const uint2 *>(xld + off + (size_t)u * ldx);
                xreg[j][2 * u] = v.x; xreg[j][2 * u + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) xreg[j][u] = *reinterpret_cast<const unsigned *>(xld + off + (size_t)u * ldx);
        }
        const unsigned cok = ci < Cin ? 1u : 0u;
        const unsigned ok0 = ((sidx >= 0) && (sidx < ldx)) ? cok : 0u, ok1 = ((sidx + 1 >= 0) && (sidx + 1 < ldx)) ? cok : 0u;
        xok = (xok & ~(1u << j)) | (ok0 << j);
        xok1 = (xok1 & ~(1u << j)) | (ok1 << j);
    };
    auto commit_x = [&](int j, int img_off) __attribute__((always_inline)) {
        unsigned item = xitem[j];
        asm volatile("" : "+v"(item));
        const int pi = (int)(item & 0xFFFFu), q = (int)((item >> 16) & 3u);
        const bool live = (item >> 31) != 0;
        const int pos = 2 * pi - 1, pos1 = 2 * pi;
        // quarter q = channels 4q..4q+3 of a position: half (q >> 1), stored swapped when bit 3 of the position is set;
        // positions outside the tile (and the items past the last one) land in the dummy slot behind the image
        // (computed unconditionally, then selected: a conditional expression here becomes an exec-masked block per store)
        const int a0 = pos * 32 + ((((q >> 1) ^ (pos >> 3)) & 1) << 4) + (q & 1) * 8;
        const int a1 = pos1 * 32 + ((((q >> 1) ^ (pos1 >> 3)) & 1) << 4) + (q & 1) * 8;
        const bool in0 = live & (pos >= 0) & (pos < SPAN), in1 = live & (pos1 < SPAN);
        const int d0 = in0 ? a0 : SPAN * 32, d1 = in1 ? a1 : SPAN * 32;
        const unsigned keep = 0u - ((xok >> j) & 1u), keep1 = 0u - ((xok1 >> j) & 1u);
        uint2 a, b;
        if constexpr (XF32) {
            auto pk = [](unsigned lo, unsigned hi) __attribute__((always_inline)) {
                return (unsigned)__builtin_bit_cast(u16, (__bf16)__uint_as_float(lo)) |
                       ((unsigned)__builtin_bit_cast(u16, (__bf16)__uint_as_float(hi)) << 16);
            };
            a = make_uint2(pk(xreg[j][0], xreg[j][2]) & keep, pk(xreg[j][4], xreg[j][6]) & keep);
            b = make_uint2(pk(xreg[j][1], xreg[j][3]) & keep1, pk(xreg[j][5], xreg[j][7]) & keep1);
        } else {
            const unsigned w0 = xreg[j][0], w1 = xreg[j][1], w2 = xreg[j][2], w3 = xreg[j][3];
            a = make_uint2(((w0 & 0xFFFFu) | (w1 << 16)) & keep, ((w2 & 0xFFFFu) | (w3 << 16)) & keep);
            b = make_uint2(((w0 >> 16) | (w1 & 0xFFFF0000u)) & keep1, ((w2 >> 16) | (w3 & 0xFFFF0000u)) & keep1);
        }
        *reinterpret_cast<uint2 *>(&lds[img_off + d0]) = a;
        *reinterpret_cast<uint2 *>(&lds[img_off + d1]) = b;
    };

    // ---- epilogue of a finished tile -------------------------------------------------------------------
    // acc[i][j][r]: output channel co0 + wco + 32 i + l31, time pt0 + 32 j + (r & 3) + 8 (r >> 2) + 4 half.
    // A lane owns a channel, so statistics and packing are lane-local — but stored straight from the registers every
    // lane would write 8 bytes into a different row (64 partial lines per instruction: measured, the epilogue then costs
    // as much as the matrix work of a 4-chunk tile).  Each 32 x 32 accumulator therefore takes one trip through a
    // wave-private 2.5 KB LDS patch ([32 channels][32 times] bf16, 80-byte rows) and leaves as two 16-byte-per-lane
    // stores: 16 rows x 64 contiguous bytes per instruction.  The patch lives in the x image of the tile's LAST chunk,
    // which nobody reads any more once the workgroup has passed the barrier in front of the epilogue.
    auto epilogue = [&](int n, int tt, int img_off) __attribute__((always_inline)) {
        const int pt0 = tt * T_T + wt;
        const bool full = pt0 + MT * 32 <= Lo;           // wave-uniform: no masks, no store predicates
        const int patch = (EPB ? WAREA + 2 * XB : img_off) + wave * (32 * EPROW);
        const int wr_off = patch + l31 * EPROW + 8 * half;               // + 16 g4 : this lane's 8 bytes of group g4
        const int rd_row = lane >> 2, rd_ch = lane & 3;                  // read side: row (and row + 16), 16-byte chunk
        const int rd_off = patch + rd_row * EPROW + 16 * rd_ch;
        __builtin_amdgcn_s_barrier();                    // every wave has issued (and received) its last reads of this image
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            u16 *yrow = y + ((size_t)n * Cout + co0 + wco + 32 * i + rd_row) * ldyo + pt0 + 8 * rd_ch;
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int tl = 32 * j + 8 * g4;      // + 4 half + e : position inside the wave tile
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g4 + e] + bia[i];
                    if (!full) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (pt0 + 4 * half + tl + e < Lo) ? v[e] : 0.f;
                    }
                    const u16 h0 = __builtin_bit_cast(u16, (__bf16)v[0]), h1 = __builtin_bit_cast(u16, (__bf16)v[1]);
                    const u16 h2 = __builtin_bit_cast(u16, (__bf16)v[2]), h3 = __builtin_bit_cast(u16, (__bf16)v[3]);
                    if (STATS) {                         // over the ROUNDED values: the tensor the BatchNorm passes read
                        const float r0 = __uint_as_float((unsigned)h0 << 16), r1 = __uint_as_float((unsigned)h1 << 16);
                        const float r2 = __uint_as_float((unsigned)h2 << 16), r3 = __uint_as_float((unsigned)h3 << 16);
                        s += (r0 + r1) + (r2 + r3);
                        q = __fmaf_rn(r0, r0, q); q = __fmaf_rn(r1, r1, q); q = __fmaf_rn(r2, r2, q); q = __fmaf_rn(r3, r3, q);
                    }
                    *reinterpret_cast<uint2 *>(&lds[wr_off + 16 * g4]) =
                        make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
                }
                // same wave, in-order LDS: the reads below see the writes above
                const uint4 o0 = *reinterpret_cast<const uint4 *>(&lds[rd_off]);
                const uint4 o1 = *reinterpret_cast<const uint4 *>(&lds[rd_off + 16 * EPROW]);
                const int tq = pt0 + 32 * j + 8 * rd_ch;                 // first of this lane's eight positions
                if (full || tq < ldyo) {                                  // ldyo % 8 == 0: eight positions are inside the row or outside
                    *reinterpret_cast<uint4 *>(yrow + 32 * j) = o0;
                    *reinterpret_cast<uint4 *>(yrow + (size_t)16 * ldyo + 32 * j) = o1;
                }
                __builtin_amdgcn_sched_barrier(0);       // one accumulator at a time
            }
            st_s[i] += s; st_q[i] += q;
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    };

    // ---- prologue -----------------------------------------------------------------------------------------
    if (RES) {
        const int npieces = KT * PPT;
        for (int p = wave; p < npieces; p += NW) dma_piece(p / PPT, p % PPT, p * 1024);
    } else {
#pragma unroll
        for (int s = 0; s < LA; ++s) dma_group(s);
    }
#pragma unroll
    for (int j = 0; j < XL; ++j) load_x(j);
#pragma unroll
    for (int j = 0; j < XL; ++j) commit_x(j, WAREA);
    ld_advance();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the asm DMA pieces are not in hipcc's books)
    __syncthreads();                                     // slots 0..3 / resident slice and x image 0 are complete
    ECG_STAMPR_AT(1);

    // ---- fragment addressing: ONE base register per operand and tap, everything else an instruction offset ----
    // (left to itself hipcc precomputes a separate address register for every (tap, sub-tile) read of the unrolled body
    // — 150 of them — and spills the accumulators)
    const int woff_lane = (wco + l31) * 32 + (((half ^ (l31 >> 3)) & 1) << 4);
    const int xoff_lane = WAREA + (wt + l31) * 32;
    unsigned hmask = 0;                                  // bit k: which 16-byte half of row (.. + l31 + k) holds this lane's channels
#pragma unroll
    for (int k = 0; k < 16; ++k) hmask |= (unsigned)((half ^ ((l31 + k) >> 3)) & 1) << k;
    auto ld_w = [&](int base, int i) __attribute__((always_inline)) {      // base: byte offset of a [CO_T][32 B] tap slice + woff_lane
        return *reinterpret_cast<const bf16x8 *>(&lds[base + i * 1024]);
    };
    auto x_base = [&](int par, int k) __attribute__((always_inline)) {     // tap k of x image `par`, sub-tile 0
        unsigned hm = hmask;
        asm volatile("" : "+v"(hm));                     // opaque: the base is computed where it is used (2 VALU), not hoisted
        return xoff_lane + par * XB + (int)(((hm >> k) & 1u) << 4);
    };
    auto ld_x = [&](int base, int k, int j) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8 *>(&lds[base + k * 32 + j * 1024]);
    };

    // MC == 2 (10 accumulators = 160 registers): the x fragments live in a ROLLING WINDOW of three register sets —
    // fragment j of body tap u sits in set (5u + j) % 3 and is read two MFMA pairs before its use — instead of five
    // sets prefetched a whole tap ahead: with the accumulators that is the difference between 215 and 260 registers.
    // MC == 1 (80 registers of accumulators): all five fragments of the next tap are prefetched (one MFMA per step is
    // too short to cover an LDS read two steps ahead).
    constexpr int XW = MC == 2 ? 3 : MT;
    bf16x8 xa[XW], wf[MC], wn[MC];
    int xb_carry = x_base(0, 0);                         // x fragment base of the tap about to be multiplied
    {
        const int xb0 = xb_carry;
#pragma unroll
        for (int i = 0; i < MC; ++i) wf[i] = ld_w(woff_lane, i);        // tap 0 of chunk 0: resident slice 0 / slot 0, tap 0
        if constexpr (MC == 2) { xa[0] = ld_x(xb0, 0, 0); xa[1] = ld_x(xb0, 0, 1); }
        else {
#pragma unroll
            for (int j = 0; j < MT; ++j) xa[j] = ld_x(xb0, 0, j);
        }
    }

    // tiles of this workgroup; a tile = nchunks / BODY_CH unrolled bodies (the host only sends layers whose chunk count
    // is a multiple of BODY_CH): no condition sits between the MFMAs of a tile, and the accumulators are re-initialised
    // at a loop boundary (a conditional epilogue inside the unrolled body made hipcc spill half of them)
    const int nbodies = nchunks / BODY_CH;
    auto run_chunk = [&](auto CB_, int body) __attribute__((always_inline)) {
            constexpr int cb = decltype(CB_)::value;
            {
                constexpr int par = cb & 1, npar = par ^ 1;
                int wres_lane = 0, wres_next_lane = 0;   // RES: this chunk's / the next chunk's first slice, + woff_lane
                if constexpr (RES) {
                    const int cc = ONECH ? 0 : body * BODY_CH + cb;
                    wres_lane = cc * (KK * CO_T * 32) + woff_lane;
                    wres_next_lane = ((cc + 1 == nchunks) ? 0 : cc + 1) * (KK * CO_T * 32) + woff_lane;
                }
                static_for<KK>([&](auto K_) __attribute__((always_inline)) {
                    constexpr int k = decltype(K_)::value;
                    constexpr int u = cb * KK + k;       // tap inside the body
                    // ---- group start: the next group's pieces have landed, the slot of the last one is free ----
                    if constexpr (!RES && u % GT == 0) {
                        // this wave issued, after its piece of group gi+1: the pieces of gi+2, gi+3 and the x loads of the
                        // last 3*GT taps.  Leave exactly those in flight (never vmcnt(0) inside the loop).
                        wait_vm<2 + x_dwords_before(u, GT, XL)>();
                        __builtin_amdgcn_s_barrier();
                        // the x commits of this chunk sit HERE, at the last group start before tap 13: hipcc waits for the
                        // x loads with counts that do not know the DMA pieces, i.e. it drains them — and at this point the
                        // pieces in flight are the two oldest ones (issued one and two groups ago)
                        if constexpr (k == commit_tap(cb, GT)) {
#pragma unroll
                            for (int j = 0; j < XL; ++j) commit_x(j, WAREA + npar * XB);
                        }
                        dma_group(((u / GT) + LA) % NSLOT);
                    }
                    if constexpr (RES && k == 0) __builtin_amdgcn_s_barrier();   // everybody has left the previous chunk's x image
                    if constexpr (k == 13) {             // the commits of taps KC0..12 become visible before tap 14 prefetches
                        wait_lgkm0();
                        __builtin_amdgcn_s_barrier();
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one tap = one scheduling region
                    // ---- staging: x of the NEXT chunk, loaded at the first taps, committed at taps KC0..12 ----
                    if constexpr (k < XL) load_x(k);
                    if constexpr (RES && k >= KC0 && k < KC0 + XL) commit_x(k - KC0, WAREA + npar * XB);
                    // ---- next tap's operand bases ----
                    constexpr int un = u + 1;            // next tap inside the body (may be the first tap of the next chunk)
                    constexpr int kn = (k + 1 < KK) ? k + 1 : 0;
                    int wbase_n;
                    if constexpr (RES) wbase_n = (k + 1 < KK) ? wres_lane + (k + 1) * (CO_T * 32) : wres_next_lane;
                    else wbase_n = woff_lane + ((un / GT) % NSLOT) * SLOTB + (un % GT) * (CO_T * 32);
                    const int xbase_c = xb_carry;                         // computed by the previous tap
                    const int xbase_n = x_base((k + 1 < KK) ? par : npar, kn);
                    xb_carry = xbase_n;
                    // ---- MFMAs of this tap ----
                    if constexpr (MC == 2) {
                        static_for<MT>([&](auto J_) __attribute__((always_inline)) {
                            constexpr int j = decltype(J_)::value;
                            constexpr int sj = (5 * u + j) % 3;                  // register set of fragment j
                            // the fragment needed two steps from now goes into the set that was last used one step ago;
                            // the next tap's weight fragments are read at steps 1 and 2
                            if constexpr (j + 2 < MT) xa[(5 * u + j + 2) % 3] = ld_x(xbase_c, k, j + 2);
                            else xa[(5 * u + j + 2) % 3] = ld_x(xbase_n, kn, j + 2 - MT);   // fragments 0, 1 of the next tap
                            if constexpr (j == 1 || j == 2) wn[j - 1] = ld_w(wbase_n, j - 1);
                            // Fragment j was read two steps ago: everything issued since (this step's reads and the last
                            // step's) may stay in flight.  Left alone hipcc waits lgkmcnt(0) on every second step, i.e. for
                            // the reads it has just issued (82 of them per 30 taps).
                            constexpr int nread[5] = {1, 2, 2, 1, 1};             // LDS reads issued at step j
                            wait_lgkm<nread[j] + nread[(j + 4) % 5]>();
                            __builtin_amdgcn_sched_barrier(0);   // (hipcc hoists a register-only MFMA above a bare s_waitcnt)
                            acc[0][j] = mfma_step<(u & 1)>(xa[sj], wf[0], acc[0][j]);
                            acc[1][j] = mfma_step<(u & 1)>(xa[sj], wf[1], acc[1][j]);
                            __builtin_amdgcn_sched_barrier(0);
                        });
                        wf[0] = wn[0]; wf[1] = wn[1];
                    } else {
                        wn[0] = ld_w(wbase_n, 0);
#pragma unroll
                        for (int j = 0; j < MT; ++j) {
                            // issued since fragment j was read: the other MT - 1 fragments and the next weight fragment
                            wait_lgkm<MT>();
                            __builtin_amdgcn_sched_barrier(0);
                            acc[0][j] = mfma_step<(u & 1)>(xa[j], wf[0], acc[0][j]);
                            xa[j] = ld_x(xbase_n, kn, j);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        wf[0] = wn[0];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                ld_advance();
            }
    };
    if constexpr (ONECH) {
        // one chunk per tile: tiles alternate between the two x images; the second half of a pair is skipped (uniformly)
        // when the workgroup's tile count is odd — nothing is live across an epilogue but zeroed accumulators
        for (int tile = q0; tile < q1; tile += 2) {
            run_chunk(std::integral_constant<int, 0>{}, 0);
            if (tile == q0) ECG_STAMPR_AT(2);
            epilogue(cn, ctt, WAREA);
            if (tile == q0) ECG_STAMPR_AT(3);
            if (++ctt == tiles_t) { ctt = 0; ++cn; }
            if (tile + 1 < q1) {
                run_chunk(std::integral_constant<int, 1>{}, 0);
                epilogue(cn, ctt, WAREA + XB);
                if (++ctt == tiles_t) { ctt = 0; ++cn; }
            }
        }
    } else {
        for (int tile = q0; tile < q1; ++tile) {
            for (int body = 0; body < nbodies; ++body)
                static_for<BODY_CH>([&](auto CB_) __attribute__((always_inline)) { run_chunk(CB_, body); });
            if (tile == q0) ECG_STAMPR_AT(2);
            epilogue(cn, ctt, WAREA + ((BODY_CH - 1) & 1) * XB);
            if (tile == q0) ECG_STAMPR_AT(3);
            if (++ctt == tiles_t) { ctt = 0; ++cn; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the ring keeps issuing to the end: let everything land
#ifdef ECG_STAMP
    if (g_stamps_r && threadIdx.x == 0) g_stamps_r[(size_t)blockIdx.x * 8 + 5] = (unsigned long long)(q1 - q0);
#endif

    if (!STATS) ECG_STAMPR_AT(4);
    if (STATS) {
        float *red = reinterpret_cast<float *>(xlds);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const float s = st_s[i] + __shfl_xor(st_s[i], 32, 64);
            const float q = st_q[i] + __shfl_xor(st_q[i], 32, 64);
            if (half == 0) {
                red[(wave * (CO_T / WCO) + 32 * i + l31) * 2] = s;
                red[(wave * (CO_T / WCO) + 32 * i + l31) * 2 + 1] = q;
            }
        }
        __syncthreads();
        for (int e = tid; e < CO_T * 2; e += NT) {
            const int col = e >> 1, w = e & 1;
            const int wrow = col / (CO_T / WCO), lc2 = col - wrow * (CO_T / WCO);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < WT; ++j) s += red[((wrow * WT + j) * (CO_T / WCO) + lc2) * 2 + w];
            partials[((size_t)(co0 + col) * P_stride + g) * 2 + w] = s;
        }
        ECG_STAMPR_AT(4);
    }
}

}  // namespace ring

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct RingPlan { bool ok; int co_t, t_t, res_ch, G; bool xf32; };

#ifndef ECG_BF16_RING
#define ECG_BF16_RING 1      // compile-time A/B knob (make VARIANT=noring EXTRA="-DECG_BF16_RING=0"): no environment is read
#endif
static constexpr int ring_enabled() { return ECG_BF16_RING; }

// Which shapes the ring kernel takes: bf16 in and out (x_bf16 / y_bf16 of the callers), K = 15, odd pad, even ldx,
// ldy % 4 == 0, and rows long enough that 640-step tiles waste no more than the 256-step tiles of the old kernel.
// xf32: the input is the fp32 network input (one 16-channel chunk: C_in <= 16), rounded to bf16 while it is staged.
RingPlan bf16_ring_plan(int N, int Cin, int Cout, int Lo, int K, int pad, int ldx, int ldyo, bool xf32) {
    RingPlan p{false, 0, 0, 0, 0, xf32};
    if (!ring_enabled() || K != 15 || (pad & 1) == 0 || Cin % 4 || Cout % 32 || ldx % 2 || ldyo % 8) return p;
    const int nch = (Cin + 15) / 16;
    if (xf32) {
        if (nch != 1) return p;
        p.co_t = 32; p.t_t = 512; p.res_ch = 1;
    } else if (Cout % 128 == 0) { p.co_t = 128; p.t_t = 640; p.res_ch = 0; }
    else if (Cout % 64 == 0) { p.co_t = 64; p.t_t = 1280; p.res_ch = nch <= 2 ? 2 : 0; }
    else { p.co_t = 32; p.t_t = 1280; p.res_ch = nch <= 4 ? 4 : -1; }
    if (p.res_ch < 0) return p;
    // a tile is a whole number of unrolled loop bodies: 2 chunks (resident weights, 128-channel ring) or 4 (64-channel ring)
    const int body_ch = (p.res_ch == 0 && p.co_t == 64) ? 4 : 2;
    if (!xf32 && nch % body_ch) return p;
    if (ring_enabled() != 2) {          // 2 = force (tests of short rows); 1 = only where the tiles fit the row
        const long long new_pad = (long long)cdiv(Lo, p.t_t) * p.t_t, old_pad = (long long)cdiv(Lo, 256) * 256;
        if (new_pad > old_pad + old_pad / 50) return p;
    }
    const int CT = Cout / p.co_t;
    const long long ntiles = (long long)N * cdiv(Lo, p.t_t);
    long long G = (xf32 ? 512 : 256) / CT;                // (the fp32-input variant runs two workgroups per CU)
    if (G < 1) G = 1;
    if (G > ntiles) G = ntiles;
    const long long per = (ntiles + G - 1) / G;
    p.G = (int)((ntiles + per - 1) / per);
    p.ok = true;
    return p;
}

int bf16_ring_launch(const RingPlan &p, const void *x, int ldx, const void *wb, const float *bias, void *y, int ldyo,
                     float *partials, int P_stride, int N, int Cin, int Cout, int L, int Lo, int pad, hipStream_t st) {
    using namespace ring;
    const int tiles_t = cdiv(Lo, p.t_t);
    dim3 grid((unsigned)((Cout / p.co_t) * p.G)), block(NT);
    const void *xh = x;
    const u16 *w = static_cast<const u16 *>(wb);
    u16 *yh = static_cast<u16 *>(y);
#define ECG_RING(CO, WCO_, WT_, STATS_, RES_)                                                                        \
    hipLaunchKernelGGL((conv1d_bf16_ring_kernel<CO, WCO_, WT_, 5, STATS_, RES_>), grid, block, 0, st, xh, w, bias, yh, \
                       partials, Cin, Cout, L, Lo, pad, tiles_t, N, p.G, P_stride, ldx, ldyo)
    if (p.xf32) {
        if (partials)
            hipLaunchKernelGGL((conv1d_bf16_ring_kernel<32, 1, 8, 2, true, 1, true>), grid, block, 0, st, xh, w, bias, yh,
                               partials, Cin, Cout, L, Lo, pad, tiles_t, N, p.G, P_stride, ldx, ldyo);
        else
            hipLaunchKernelGGL((conv1d_bf16_ring_kernel<32, 1, 8, 2, false, 1, true>), grid, block, 0, st, xh, w, bias, yh,
                               partials, Cin, Cout, L, Lo, pad, tiles_t, N, p.G, P_stride, ldx, ldyo);
    } else if (p.co_t == 128) { if (partials) ECG_RING(128, 2, 4, true, 0); else ECG_RING(128, 2, 4, false, 0); }
    else if (p.co_t == 64 && p.res_ch) { if (partials) ECG_RING(64, 1, 8, true, 2); else ECG_RING(64, 1, 8, false, 2); }
    else if (p.co_t == 64) { if (partials) ECG_RING(64, 1, 8, true, 0); else ECG_RING(64, 1, 8, false, 0); }
    else { if (partials) ECG_RING(32, 1, 8, true, 4); else ECG_RING(32, 1, 8, false, 4); }
#undef ECG_RING
    return check_launch("conv1d_bf16_ring_kernel");
}

}  // namespace ecg

#ifdef ECG_STAMP
extern "C" __attribute__((visibility("default"))) int ecg_debug_set_stamp_buffer_ring(unsigned long long *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(ecg::g_stamps_r), &buf, sizeof(buf));
}
#endif
