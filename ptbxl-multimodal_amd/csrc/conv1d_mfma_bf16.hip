// conv1d_mfma_bf16.hip — mixed-precision Conv1d forward / input-grad: bf16 operands on the
// gfx950 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate), fp32 activations in HBM.
// This is the opt-in path of BASELINE.json config 5 ("AF binary, 12x5000, bf16 mixed precision");
// the fp32 MFMA kernels of conv1d_mfma.hip remain the default and the parity path.  The weight
// gradient of this mode is in conv1d_wgrad_bf16.hip.
//
// Same im2col-free structure as the fp32 kernel, with a 16-channel reduction block per MFMA:
//   D[co][t] += sum_{ci in chunk of 16} W[tap][co][ci] * X[t + tap][ci]       (one MFMA per tap)
//   MFMA A = weight fragment: lane (co = l&31, h = l>>5) holds ci 8h..8h+7  -> LDS [tap][co][16]
//   MFMA B = x fragment:      lane (t  = l&31, h)       holds ci 8h..8h+7  -> LDS [pos][16]
// so both fragments are single 16-byte LDS reads.  Weights are pre-packed to bf16 in that order
// ([chunk][tap][C_out][16]) and stream global -> LDS by DMA; the x tile is converted fp32 -> bf16
// (round to nearest even) while it is staged, with its zero padding applied by AND masks.
//
// Replaces the ATen work behind ConvBlock.net[0] (reference src/models/ecg_cnn.py:13); the
// reference itself has no mixed precision (`amp: true` in its YAML is a dead key).
#include "common.h"

namespace ecg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// in-kernel stamps of the diagnostic build (make STAMP=1), as in conv1d_mfma.hip
#ifdef ECG_STAMP
__device__ unsigned long long *g_stamps_b = nullptr;
#define ECG_STAMPB_AT(slot) do { if (g_stamps_b && threadIdx.x == 0) { \
    g_stamps_b[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    if ((slot) == 0) g_stamps_b[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime(); \
    if ((slot) == 4) g_stamps_b[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define ECG_STAMPB_AT(slot) do { } while (0)
#endif

constexpr int kKB = 15;      // kernel size staged by this path
constexpr int kCB = 16;      // input channels per MFMA (its K dimension)

__device__ __forceinline__ int acc_row_b(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// s_waitcnt vmcnt(N) with lgkmcnt / expcnt left at "no wait"
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(((N >> 4) << 14) | 0x0F70 | (N & 15));
}

// s_waitcnt lgkmcnt(n) with vmcnt / expcnt left at "no wait" (n is a compile-time constant after unrolling)
__device__ __forceinline__ void wait_lgkm(int n) {
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(0xC07F); break;
        case 1: __builtin_amdgcn_s_waitcnt(0xC17F); break;
        case 2: __builtin_amdgcn_s_waitcnt(0xC27F); break;
        case 3: __builtin_amdgcn_s_waitcnt(0xC37F); break;
        case 4: __builtin_amdgcn_s_waitcnt(0xC47F); break;
        case 5: __builtin_amdgcn_s_waitcnt(0xC57F); break;
        case 6: __builtin_amdgcn_s_waitcnt(0xC67F); break;
        case 7: __builtin_amdgcn_s_waitcnt(0xC77F); break;
        case 8: __builtin_amdgcn_s_waitcnt(0xC87F); break;
        default: break;                                   // more than the counter is worth tracking: let the compiler decide
    }
}

__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    const u16 a = __builtin_bit_cast(u16, (__bf16)lo), b = __builtin_bit_cast(u16, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

// w [Co][Ci][K] fp32 -> wb_fwd [ceil(Ci/16)][K][Co][16] and wb_bwd [ceil(Co/16)][K][Ci][16]
// (tap-flipped, roles of the channel axes swapped), zero-filled past the channel count.
__global__ void pack_weights_bf16_kernel(const float *__restrict__ w, u16 *__restrict__ wb_fwd,
                                         u16 *__restrict__ wb_bwd, int Co, int Ci, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nf = ((Ci + kCB - 1) / kCB) * K * Co * kCB;
    const int nb = ((Co + kCB - 1) / kCB) * K * Ci * kCB;
    if (wb_fwd && idx < nf) {
        const int j = idx % kCB, co = (idx / kCB) % Co, k = (idx / (kCB * Co)) % K, c = idx / (kCB * Co * K);
        const int ci = c * kCB + j;
        const float v = ci < Ci ? w[((size_t)co * Ci + ci) * K + k] : 0.f;
        wb_fwd[idx] = __builtin_bit_cast(u16, (__bf16)v);
    }
    if (wb_bwd && idx < nb) {
        const int j = idx % kCB, ci = (idx / kCB) % Ci, k = (idx / (kCB * Ci)) % K, c = idx / (kCB * Ci * K);
        const int co = c * kCB + j;
        const float v = co < Co ? w[((size_t)co * Ci + ci) * K + (K - 1 - k)] : 0.f;
        wb_bwd[idx] = __builtin_bit_cast(u16, (__bf16)v);
    }
}

// PERSISTENT workgroups: grid = (Cout/CO_T) * G; workgroup (tile_co, g) owns the (n, t-tile) tiles
// [ntiles*g/G, ntiles*(g+1)/G) of its C_out tile; WCO x WT waves (4 or 8) per workgroup.
//
// With bf16 operands an MFMA is 16x faster than the fp32 one, so what the fp32 kernel could ignore decides here:
//   * one tile per workgroup (measured, 12x5000, in-kernel stamps): first operand fetch 2.6-3.8 us + store of the
//     tile 3-5.3 us against 0.5-20 us of MFMA work per tile — 25 % (block 3) to 95 % (block 0) of a workgroup's life
//     with the matrix pipe idle.  So the chunk pipeline runs FLAT across the tiles of a workgroup (the "next chunk"
//     may be chunk 0 of the next tile: only the first tile pays a prologue), and a finished tile is stored
//     fire-and-forget: the wave issues its stores (bias, BN statistics on the way) and goes straight on to the next
//     tile, whose operands are already staged; the stores drain under its MFMAs.  (Parking the finished accumulators
//     in a second register set to interleave the stores with the next tile's MFMAs was tried: 256 VGPRs + scratch.)
//   * weight chunk [15][CO_T][16] bf16 per (tile, 16 input channels): L2 -> LDS by DMA.  A 64 x 128 tile needs
//     ~31 B/clk/CU of it with four workgroups per CU — the L2 -> LDS rate a CU sustains (~30 B/clk).  Tiles are
//     therefore as large as the layer allows: 128 x 256 (eight waves, one workgroup per CU) re-streams 4x fewer
//     weight bytes per MFMA, 64 x 256 2x fewer.
//   * LDS reads: one 16-byte fragment per lane feeds MC*MT MFMAs per MC+MT reads; MC = MT = 2 needs 128 B/clk/CU
//     (half the LDS peak), MC = 1, MT = 2 needs 192.  Rows are 32 bytes ([16 channels] bf16) and a lane reads one
//     16-byte half of a row: read plainly, the 16 lanes ds_read_b128 serves per cycle ({0-3,12-15,20-27}, ...)
//     hit every bank twice.  The two halves of row r are therefore stored swapped when bit 3 of r is set (the rows a
//     lane group pairs up are 8 or 24 apart, for any tap shift): conflict-free, one LDS cycle per 16 lanes.
//   * BN statistics stay in registers until the workgroup ends (16-lane DPP row sums kept in lanes r / r+16 of one
//     VGPR per 32 channels): one (sum, sum^2) partial per (channel, workgroup), P = G.
//   * XH: the input itself is bf16 ([N][C_in][ldx] u16, rows zero-filled past L, ldx even) — the input-gradient
//     conv reads the dY the BatchNorm backward wrote in bf16.  An item is then a PAIR of positions (2i-1, 2i) of four
//     channels: four aligned 4-byte loads (half the bytes, half the loads of the fp32 input) feeding two LDS rows.
//   * WRES: layers with at most two 16-channel chunks (blocks 0 and 1 forward) keep their whole weight slice
//     resident — image 0 holds chunk 0, image 1 chunk 1 (or chunk 0 again), loaded once per workgroup; only x tiles
//     stream after that.
//   * YH: the OUTPUT is stored as bf16 ([N][C_out][ldyo] u16, row stride ldyo >= Lo) — bf16 activation storage of the
//     train step: the BatchNorm passes then read half the bytes.  The statistics are taken over the ROUNDED values,
//     i.e. exactly over the tensor the BatchNorm passes will read.
template <int CO_T, int T_T, int WCO, int WT, bool STATS, bool XH = false, bool WRES = false, bool YH = false>
__global__ __launch_bounds__(64 * WCO * WT) __attribute__((amdgpu_waves_per_eu(2))) void conv1d_mfma_bf16_fwd_kernel(
    const float *__restrict__ x, const u16 *__restrict__ wb, const float *__restrict__ bias,
    float *__restrict__ y, float *__restrict__ partials, int Cin, int Cout, int L, int Lo, int pad,
    int tiles_t, int N, int G, int ldx, int ldyo) {
    constexpr int NW = WCO * WT, NT = 64 * NW;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int KK = kKB;
    constexpr int MC = CO_T / WCO / 32, MT = T_T / WT / 32;
    static_assert(MC >= 1 && MT >= 1, "wave tile must hold at least one 32x32 accumulator");
    constexpr int SPAN = T_T + KK - 1;                   // x-tile positions
    constexpr int WBYTES = KK * CO_T * kCB * 2;          // bytes of one weight chunk [K][CO_T][16] bf16
    constexpr int NDMA = (WBYTES + 1023) / 1024;         // 1 KB wave-instructions per chunk
    constexpr int DPW = (NDMA + NW - 1) / NW;
    constexpr int WPADB = NDMA * 1024;                   // every wave issues DPW pieces unconditionally: pieces past the
                                                         // slice repeat its last piece (same bytes, same place)
    constexpr int NPAIR = SPAN / 2 + 1;                  // XH: position pairs (2i-1, 2i), i = 0 .. SPAN/2
    constexpr int XITEMS = XH ? NPAIR * 4 : SPAN * 4;    // (pos or pos pair, quarter of 4 channels)
    constexpr int XL = (XITEMS + NT - 1) / NT;
    constexpr int XBYTES = ((SPAN * kCB * 2 + 8 + 15) / 16) * 16;   // every thread commits XL items unconditionally: items
                                                         // past the tile all land in one dummy slot behind it
    constexpr int IMGB = WPADB + XBYTES;
    constexpr int NOPS = XL + DPW + XL;                  // DMA pieces, commits, loads
    constexpr int OPS = (NOPS + KK - 1) / KK;            // staging operations per tap step
    constexpr int REDB = STATS ? NW * (CO_T / WCO) * 2 * 4 : 0;
    static_assert(REDB <= IMGB, "stat scratch aliases image 0");
    static_assert(2 * IMGB <= 160 * 1024, "LDS");

    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * IMGB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    ECG_STAMPB_AT(0);
    // XCD-aware workgroup order (as conv1d_mfma.hip): the C_out tiles of one tile range read the same x panels
    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int CT = Cout / CO_T;
    const int tile_co = wg % CT, g = wg / CT;
    const int ntiles = N * tiles_t;
    const int q0 = (int)((long long)ntiles * g / G), q1 = (int)((long long)ntiles * (g + 1) / G);
    const int co0 = tile_co * CO_T;
    const int wco = (wave / WT) * (CO_T / WCO), wt = (wave % WT) * (T_T / WT);
    const int nchunks = (Cin + kCB - 1) / kCB;
    const int total = (q1 - q0) * nchunks;               // flat chunks of this workgroup
    if (total <= 0) return;                              // uniform: more workgroups than tiles

    f32x16 acc[MC][MT];
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // bias of the 16 accumulator rows of each 32-channel group, lane-indexed (lane r + 32*half holds row (r, half)),
    // loaded before the main loop: a global load between the epilogue's stores would wait for their round trips
    float p_b[MC], st_s[MC], st_q[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        p_b[i] = bias ? bias[co0 + wco + 32 * i + acc_row_b(l31 & 15, half)] : 0.f;
        st_s[i] = 0.f; st_q[i] = 0.f;
    }

    // ---- loop-invariant staging geometry ---------------------------------------------------
    // weight piece j of this wave: byte offset inside a chunk [K][Cout][16] bf16 -> (k, co, lane part); pieces past
    // the slice re-read its last piece
    int woff[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int e = min((min(j * NW + wave, NDMA - 1) * 64 + lane) * 16, WBYTES - 16);   // byte in the LDS image
        const int k = e / (CO_T * kCB * 2), rem = e - k * (CO_T * kCB * 2);  // rem = co_local*32 + stored half*16
        const int col = rem >> 5, sh = (rem >> 4) & 1;                       // LDS half sh of row col holds half sh ^ bit3(col)
        woff[j] = (k * Cout + co0) * kCB * 2 + col * 32 + ((sh ^ ((col >> 3) & 1)) << 4);   // byte offset in the global chunk
    }
    // fp32 input: item = (pos, quarter); bf16 input: item = (pair i -> positions 2i-1 and 2i, quarter)
    int xdst[XL], xdst1[XL], xq4[XL], xpm[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        const int it = tid + NT * j;
        const int itc = min(it, XITEMS - 1);
        const int per = XH ? NPAIR : SPAN;
        const int q = itc / per, pi = itc - q * per;
        const int pos = XH ? 2 * pi - 1 : pi, pos1 = 2 * pi;
        xq4[j] = 4 * q;                 // first channel of the quarter inside the chunk
        xpm[j] = pos - pad;             // sequence index (of the first position) relative to the tile origin
        // quarter q = channels 4q..4q+3 of position pos: half (q >> 1), stored swapped when bit 3 of pos is set;
        // positions outside the tile (and the items past the last one) land in the dummy slot behind it
        auto dst = [&](int ps) {
            return (it < XITEMS && ps >= 0 && ps < SPAN) ? ps * 32 + ((((q >> 1) ^ (ps >> 3)) & 1) << 4) + (q & 1) * 8
                                                        : SPAN * kCB * 2;
        };
        xdst[j] = dst(pos);
        xdst1[j] = dst(pos1);
    }
    float xreg[XL][4];                // fp32 input: four channels of one position; bf16 input: four dwords = two positions each
    unsigned xok = 0, xok1 = 0;       // bit j: (first / second position of) item j in registers is real data

    // ---- stage coordinates: uniform, advanced by additions only ---------------------------------
    int cn = q0 / tiles_t, ctt = q0 - cn * tiles_t, cc = 0;       // compute stage: (n, t tile, chunk)
    // load stage: sample base (in floats; a bf16 sample is Cin*ldx/2 floats), t tile, chunk
    const size_t xstep_n = XH ? (size_t)Cin * ldx / 2 : (size_t)Cin * L;
    const float *xld = x + (size_t)cn * xstep_n;
    int ltt = ctt, lc = 0, lleft = total;
    auto ld_advance = [&]() {                      // stays on the last chunk once everything is loaded
        if (--lleft > 0) {
            if (++lc == nchunks) {
                lc = 0;
                if (++ltt == tiles_t) { ltt = 0; xld += xstep_n; }
            }
        } else lleft = 1;
    };

    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wb);
    const size_t chunk_bytes = (size_t)KK * Cout * kCB * 2;
    auto dma_w = [&](int j, const unsigned char *wchunk, unsigned char *img) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(wchunk + woff[j]),
            (__attribute__((address_space(3))) void *)(img + min(j * NW + wave, NDMA - 1) * 1024), 16, 0, 0);
    };
    auto load_x = [&](int j) {                           // item j of the x tile of the load stage
        const int ci = lc * kCB + xq4[j];
        const int sidx = ltt * T_T + xpm[j];
        if (XH) {
            // positions (sidx, sidx + 1), sidx even (T_T even, pad odd — the host checks): one aligned dword per channel.
            // Rows are zero-filled from L to ldx, so only the row's own bounds need a mask.
            const u16 *xh = reinterpret_cast<const u16 *>(xld);
            const int sc = min(max(sidx, 0), ldx - 2);
            const size_t off = (size_t)min(ci, Cin - 4) * ldx + sc;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                xreg[j][u] = __uint_as_float(*reinterpret_cast<const unsigned *>(xh + off + (size_t)u * ldx));
            const unsigned cok = ci < Cin ? 1u : 0u;
            const unsigned ok0 = ((sidx >= 0) && (sidx < ldx)) ? cok : 0u, ok1 = ((sidx + 1 >= 0) && (sidx + 1 < ldx)) ? cok : 0u;
            xok = (xok & ~(1u << j)) | (ok0 << j);
            xok1 = (xok1 & ~(1u << j)) | (ok1 << j);
        } else {
            const int off = min(ci, Cin - 4) * L + min(max(sidx, 0), L - 1);      // Cin % 4 == 0: a quarter is all valid or all padding
#pragma unroll
            for (int u = 0; u < 4; ++u) xreg[j][u] = xld[off + u * L];
            const unsigned ok = ((sidx >= 0) && (sidx < L) && (ci < Cin)) ? 1u : 0u;
            xok = (xok & ~(1u << j)) | (ok << j);
        }
    };
    auto commit_x = [&](int j, unsigned char *img) {
        const unsigned keep = 0u - ((xok >> j) & 1u);
        if (XH) {
            const unsigned keep1 = 0u - ((xok1 >> j) & 1u);
            const unsigned w0 = __float_as_uint(xreg[j][0]), w1 = __float_as_uint(xreg[j][1]);
            const unsigned w2 = __float_as_uint(xreg[j][2]), w3 = __float_as_uint(xreg[j][3]);
            // low halves = first position, high halves = second position, of channels 4q .. 4q+3
            const uint2 a = make_uint2(((w0 & 0xFFFFu) | (w1 << 16)) & keep, ((w2 & 0xFFFFu) | (w3 << 16)) & keep);
            const uint2 b = make_uint2(((w0 >> 16) | (w1 & 0xFFFF0000u)) & keep1, ((w2 >> 16) | (w3 & 0xFFFF0000u)) & keep1);
            *reinterpret_cast<uint2 *>(img + WPADB + xdst[j]) = a;
            *reinterpret_cast<uint2 *>(img + WPADB + xdst1[j]) = b;
        } else {
            const unsigned lo = pack2_bf16(xreg[j][0], xreg[j][1]) & keep;
            const unsigned hi = pack2_bf16(xreg[j][2], xreg[j][3]) & keep;
            *reinterpret_cast<uint2 *>(img + WPADB + xdst[j]) = make_uint2(lo, hi);
        }
    };

    // ---- epilogue of a finished tile: one (accumulator row r, channel group i) item per call ---------
    float *ytile = y;                    // (n, co0, t0) of the finished tile (YH: the same in u16 elements)
    int pt0 = 0;
    const int ylane = (wco + 4 * half) * ldyo + wt + l31;
    auto epilogue_item = [&](int it) {
        const int i = it >> 4, r = it & 15;
        const int bi = __float_as_int(p_b[i]);
        const float blo = __int_as_float(__builtin_amdgcn_readlane(bi, r));
        const float bhi = __int_as_float(__builtin_amdgcn_readlane(bi, r + 32));
        const float bv = half ? bhi : blo;
        const int yoff = (32 * i + (r & 3) + 8 * (r >> 2)) * ldyo + ylane;
        float *yr = ytile + yoff;
        u16 *yhr = reinterpret_cast<u16 *>(ytile) + yoff;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            float v = acc[i][j][r] + bv;
            if (pt0 + wt + 32 * j + l31 < Lo) {
                if (YH) {
                    const u16 h = __builtin_bit_cast(u16, (__bf16)v);
                    yhr[32 * j] = h;
                    v = __uint_as_float((unsigned)h << 16);
                } else yr[32 * j] = v;
                if (STATS) { s += v; q = __fmaf_rn(v, v, q); }
            }
        }
        if (STATS) {
            const bool mine = (l31 & 15) == r;           // lanes r and r+16 of each half keep row (r, half)
            s = row16_sum(s);
            q = row16_sum(q);
            st_s[i] += mine ? s : 0.f;
            st_q[i] += mine ? q : 0.f;
        }
    };

    // prologue: flat chunk 0 -> image 0; x of flat chunk 1 -> registers
#pragma unroll
    for (int j = 0; j < DPW; ++j) dma_w(j, wbase, lds);
    if (WRES) {               // weights of flat chunk 1 (chunk 1, or chunk 0 again for a one-chunk layer) -> image 1, for good
#pragma unroll
        for (int j = 0; j < DPW; ++j) dma_w(j, wbase + (size_t)(nchunks > 1 ? 1 : 0) * chunk_bytes, lds + IMGB);
    }
#pragma unroll
    for (int j = 0; j < XL; ++j) load_x(j);
#pragma unroll
    for (int j = 0; j < XL; ++j) commit_x(j, lds);
    ld_advance();
#pragma unroll
    for (int j = 0; j < XL; ++j) load_x(j);
    ld_advance();
    __syncthreads();
    ECG_STAMPB_AT(1);

    for (int q = 0; q < total; ++q) {        // one flat chunk: 15 tap steps
        const unsigned char *ws = lds + (q & 1) * IMGB, *xs = ws + WPADB;
        unsigned char *nxt = lds + ((q + 1) & 1) * IMGB;
        const unsigned char *wnext = wbase + (size_t)((cc + 1 == nchunks) ? 0 : cc + 1) * chunk_bytes;   // weights of flat chunk q+1
        auto ld = [&](int k, bf16x8 *a, bf16x8 *b) {
            const int ha = (half ^ (l31 >> 3)) & 1;              // CO_T, wco, 32*i are multiples of 32: bit 3 of the row is bit 3 of l31
            const int hb = (half ^ ((l31 + k) >> 3)) & 1;        // wt, 32*i likewise: bit 3 of the position is bit 3 of l31 + k
#pragma unroll
            for (int i = 0; i < MC; ++i)
                a[i] = *reinterpret_cast<const bf16x8 *>(ws + (k * CO_T + wco + 32 * i + l31) * 32 + ha * 16);
#pragma unroll
            for (int i = 0; i < MT; ++i)
                b[i] = *reinterpret_cast<const bf16x8 *>(xs + (wt + 32 * i + l31 + k) * 32 + hb * 16);
        };
        bf16x8 a_c[MC], b_c[MT], a_n[MC], b_n[MT];
        ld(0, a_c, b_c);
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            ld(k + 1 < KK ? k + 1 : 0, a_n, b_n);
            // The MFMAs below need the fragments read ONE step ago; the MC + MT reads just issued may stay in flight.
            // Left alone, hipcc waits lgkmcnt(0) on every second step and after every LDS-DMA issue — the full LDS
            // latency of the reads just issued, in front of MFMAs that do not need them.  The explicit counted wait
            // sits BEFORE the staging operation so that the compiler sees the operands complete at that point.
            wait_lgkm(MC + MT);
#pragma unroll
            for (int o = k * OPS; o < (k + 1) * OPS; ++o) {      // weight DMA pieces first (longest latency), x commits, x loads;
                if (o < DPW) { if (!WRES) dma_w(o, wnext, nxt); }   // all UNCONDITIONAL: stages past the end restage valid data
                else if (o < DPW + XL) {
                    // The commits need the x loads of the PREVIOUS chunk; the only vector-memory operations issued since
                    // are this chunk's DPW weight DMA pieces (vmcnt counts in issue order).  Left alone, hipcc puts
                    // s_waitcnt vmcnt(0) here (it loses the count over the loop's back edge): every wave then waits
                    // for the L2 round trip of the DMA pieces it issued a few taps ago, in the middle of the chunk.
                    if (o == DPW) wait_vm<WRES ? 0 : DPW>();
                    commit_x(o - DPW, nxt);
                }
                else if (o < NOPS) load_x(o - XL - DPW);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c[i], b_c[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MC; ++i) a_c[i] = a_n[i];
#pragma unroll
            for (int i = 0; i < MT; ++i) b_c[i] = b_n[i];
        }
        ld_advance();
        // End of chunk: image q&1 is free again once everybody is here, and image (q+1)&1 is complete once every wave's
        // DMA pieces and commits have landed.  __syncthreads() would ALSO wait (vmcnt(0)) for the x loads of flat chunk
        // q+2 issued a few steps ago — an HBM round trip of 1-2 us in front of every barrier, half a chunk's time.  They
        // are the NEWEST 4*XL vector-memory operations of this wave (vmcnt counts in issue order), so wait for everything
        // older and let them fly; they are consumed by the commits of the next chunk.
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(4 * XL) : "memory");
        if (q == 0) ECG_STAMPB_AT(2);
        if (++cc == nchunks) {                     // tile complete: store it and go straight on
            cc = 0;
            ytile = YH ? reinterpret_cast<float *>(reinterpret_cast<u16 *>(y) + ((size_t)cn * Cout + co0) * ldyo + ctt * T_T)
                       : y + ((size_t)cn * Cout + co0) * ldyo + ctt * T_T;
            pt0 = ctt * T_T;
            // only the x loads of the chunk after next are in flight here; the epilogue does not touch their registers
#pragma unroll
            for (int e = 0; e < 16 * MC; ++e) epilogue_item(e);
#pragma unroll
            for (int a = 0; a < MC; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
            if (++ctt == tiles_t) { ctt = 0; ++cn; }
        }
    }
    ECG_STAMPB_AT(3);
    if (STATS) {
        float *red = reinterpret_cast<float *>(lds);      // all images are dead (last chunk's barrier)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const float s = st_s[i] + __shfl_xor(st_s[i], 16, 64);
            const float q = st_q[i] + __shfl_xor(st_q[i], 16, 64);
            if (l31 < 16) {
                const int lc2 = 32 * i + acc_row_b(l31, half);
                red[(wave * (CO_T / WCO) + lc2) * 2] = s;
                red[(wave * (CO_T / WCO) + lc2) * 2 + 1] = q;
            }
        }
        __syncthreads();
        for (int e = tid; e < CO_T * 2; e += NT) {
            const int col = e >> 1, w = e & 1;
            const int wrow = col / (CO_T / WCO), lc2 = col - wrow * (CO_T / WCO);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < WT; ++j) s += red[((wrow * WT + j) * (CO_T / WCO) + lc2) * 2 + w];
            partials[((size_t)(co0 + col) * G + g) * 2 + w] = s;
        }
    }
#ifdef ECG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    ECG_STAMPB_AT(4);
}

bool bf16_fwd_supported(int Cin, int Cout, int K, int pad) {
    (void)pad;
    return K == kKB && Cin % 4 == 0 && Cout % 32 == 0;
}

// Tile choice: the largest C_out tile the layer has, 256 time steps when the row is long enough to fill them
// (at 12x1000 the last block's rows are 125 long: a 256-wide tile would idle half its lanes); G persistent
// workgroups per C_out tile fill the resident slots (LDS decides how many fit a CU) with equal tile counts.
struct Bf16Cfg { int co_t, t_t, G; };
static Bf16Cfg bf16_cfg(int N, int Cout, int Lo) {
    const bool wide = Lo > 160;
    Bf16Cfg c;
    int per_cu;                                       // resident workgroups per CU (LDS: 2 images each)
    if (Cout % 128 == 0 && wide) { c = {128, 256, 0}; per_cu = 1; }
    else if (Cout % 64 == 0) { c = {64, wide ? 256 : 128, 0}; per_cu = 2; }
    else { c = {32, 256, 0}; per_cu = 2; }            // (VGPRs allow two 4-wave workgroups per CU)
    const int CT = Cout / c.co_t;
    const long long ntiles = (long long)N * cdiv(Lo, c.t_t);
    long long G = (256LL * per_cu) / CT;
    if (G > ntiles) G = ntiles;
    if (G < 1) G = 1;
    const long long per = (ntiles + G - 1) / G;       // tiles of the busiest workgroup
    c.G = (int)((ntiles + per - 1) / per);            // fewest workgroups with that maximum
    return c;
}

int bf16_fwd_stat_partials(int N, int Cout, int Lo) { return bf16_cfg(N, Cout, Lo).G; }

size_t bf16_packed_elems(int Cred, int Cout, int K) {       // reduction channels padded to 16
    return (size_t)((Cred + kCB - 1) / kCB) * K * Cout * kCB;
}

template <int CO_T, int T_T, int WCO, int WT>
static void launch_bf16(const void *x, int ldx, bool xh, const u16 *wb, const float *bias, float *y, int ldyo, bool yh,
                        float *partials, int N, int Cin, int Cout, int L, int Lo, int pad, int G, hipStream_t st) {
    const int tiles_t = cdiv(Lo, T_T);
    dim3 grid((unsigned)((size_t)(Cout / CO_T) * G)), block(64 * WCO * WT);
    const float *xf = static_cast<const float *>(x);
#define ECG_BF(STATS, XH, WRES, YH)                                                                                    \
    hipLaunchKernelGGL((conv1d_mfma_bf16_fwd_kernel<CO_T, T_T, WCO, WT, STATS, XH, WRES, YH>), grid, block, 0, st, xf, \
                       wb, bias, y, partials, Cin, Cout, L, Lo, pad, tiles_t, N, G, ldx, ldyo)
    // y / dx always leave as bf16 rows (round 5: the mixed-precision step has ONE form); x is the previous block's bf16
    // activation / this block's bf16 dY (xh), or the fp32 network input of the first block
    (void)yh;
    if (xh) { // bf16 in, bf16 out: the train-mode forward of an inner block (statistics), or its input gradient (none)
        if (!partials) ECG_BF(false, true, false, true);
        else if (Cin <= 2 * kCB) ECG_BF(true, true, true, true);
        else ECG_BF(true, true, false, true);
    } else { // fp32 network input (train-mode forward: always with statistics)
        if (Cin <= 2 * kCB) ECG_BF(true, false, true, true);
        else ECG_BF(true, false, false, true);
    }
#undef ECG_BF
}

// conv1d_bf16_ring.hip: the round-3 kernel for long rows with bf16 activations on both sides
struct RingPlan { bool ok; int co_t, t_t, res_ch, G; bool xf32; };
RingPlan bf16_ring_plan(int N, int Cin, int Cout, int Lo, int K, int pad, int ldx, int ldyo, bool xf32 = false);
int bf16_ring_launch(const RingPlan &p, const void *x, int ldx, const void *wb, const float *bias, void *y, int ldyo,
                     float *partials, int P_stride, int N, int Cin, int Cout, int L, int Lo, int pad, hipStream_t st);

// x: fp32 [N][Cin][L] (xh false, ldx ignored) or bf16 [N][Cin][ldx] with rows zero-filled past L (xh true);
// y: fp32 [N][Cout][ldyo] or (yh, with statistics) bf16 [N][Cout][ldyo]
static int bf16_fwd_any(const void *x, int ldx, bool xh, const void *wb, const float *bias, float *y, int ldyo, bool yh,
                        float *partials, int N, int Cin, int Cout, int L, int K, int pad, hipStream_t st) {
    const int Lo = L + 2 * pad - K + 1;
    const u16 *w = static_cast<const u16 *>(wb);
    if (yh) {
        const RingPlan rp = bf16_ring_plan(N, Cin, Cout, Lo, K, pad, ldx, ldyo, !xh);
        // The kernel is chosen by SHAPE only, so that ecg_conv1d_fwd_bf16_yh_stat_partials (which sees no pointers) always
        // agrees with the launch about the number of partials; operands the chosen kernel cannot address are refused.
        // (fp32 input: the rows are [L] floats read as aligned pairs; torch allocations and whole-sample slices satisfy both)
        if (rp.ok)
            ECG_REQUIRE(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(wb)) & 15) == 0 &&
                            (reinterpret_cast<uintptr_t>(x) & (xh ? 3 : 7)) == 0,
                        "conv1d bf16 (long rows): y / packed weights must be 16-byte aligned, x %d-byte aligned", xh ? 4 : 8);
        if (rp.ok) return bf16_ring_launch(rp, x, ldx, wb, bias, y, ldyo, partials, rp.G, N, Cin, Cout, L, Lo, pad, st);
    }
    const Bf16Cfg c = bf16_cfg(N, Cout, Lo);
    if (c.co_t == 128) launch_bf16<128, 256, 2, 4>(x, ldx, xh, w, bias, y, ldyo, yh, partials, N, Cin, Cout, L, Lo, pad, c.G, st);
    else if (c.co_t == 64 && c.t_t == 256) launch_bf16<64, 256, 1, 4>(x, ldx, xh, w, bias, y, ldyo, yh, partials, N, Cin, Cout, L, Lo, pad, c.G, st);
    else if (c.co_t == 64) launch_bf16<64, 128, 2, 2>(x, ldx, xh, w, bias, y, ldyo, yh, partials, N, Cin, Cout, L, Lo, pad, c.G, st);
    else launch_bf16<32, 256, 1, 4>(x, ldx, xh, w, bias, y, ldyo, yh, partials, N, Cin, Cout, L, Lo, pad, c.G, st);
    return check_launch("conv1d_mfma_bf16_fwd_kernel");
}

int bf16_pack(const float *w, void *wb_fwd, void *wb_bwd, int Co, int Ci, int K, hipStream_t st) {
    const size_t n = bf16_packed_elems(Ci, Co, K), m = bf16_packed_elems(Co, Ci, K);
    const size_t total = n > m ? n : m;
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(cdiv((long long)total, 256)), dim3(256), 0, st, w,
                       static_cast<u16 *>(wb_fwd), static_cast<u16 *>(wb_bwd), Co, Ci, K);
    return check_launch("pack_weights_bf16_kernel");
}

}  // namespace ecg

using namespace ecg;

static int check_bf16_shape(const char *who, int N, int Cin, int Cout, int L, int K, int pad) {
    ECG_REQUIRE(N > 0 && N <= 65535 && Cin > 0 && Cout > 0 && L > 0, "%s: bad shape", who);
    ECG_REQUIRE(K == kKB && pad >= 0 && pad < K && L + 2 * pad - K + 1 > 0, "%s: kernel size must be 15", who);
    return ECG_OK;
}

namespace ecg {
bool wgrad_bf16_tk_supported(int Cin, int Cout, int K, int pad);      // conv1d_wgrad_bf16_tk.hip
}  // namespace ecg

ECG_API int ecg_conv1d_bf16_supported(int C_in, int C_out, int K, int pad) {
    // bit 0: forward (C_in % 4 == 0, C_out % 32 == 0); bit 1: input-grad (roles swapped);
    // bit 2: weight-grad, the time-on-K kernel (K == 15, pad == 7, C_out % 32 == 0)
    return (bf16_fwd_supported(C_in, C_out, K, pad) ? 1 : 0) |
           (bf16_fwd_supported(C_out, C_in, K, K - 1 - pad) ? 2 : 0) |
           (wgrad_bf16_tk_supported(C_in, C_out, K, pad) ? 4 : 0);
}

ECG_API size_t ecg_conv1d_bf16_packed_elems(int C_reduce, int C_result, int K) {
    return bf16_packed_elems(C_reduce, C_result, K);
}

ECG_API int ecg_conv1d_pack_weights_bf16(const float *w, void *wb_fwd, void *wb_bwd, int C_out,
                                         int C_in, int K, ecg_stream_t stream) {
    ECG_REQUIRE(w && (wb_fwd || wb_bwd) && C_out > 0 && C_in > 0 && K >= 1 && K <= 31,
                "pack_weights_bf16: bad argument");
    return bf16_pack(w, wb_fwd, wb_bwd, C_out, C_in, K, as_stream(stream));
}

// partials per channel that ecg_conv1d_fwd_bf16_yh writes for these arguments (it picks its kernel by the operand
// types and row strides too)
ECG_API int ecg_conv1d_fwd_bf16_yh_stat_partials(int N, int C_in, int C_out, int L, int K, int pad, int x_bf16, int ldx,
                                                 int ldy) {
    const int Lo = L + 2 * pad - K + 1;
    {
        const RingPlan rp = bf16_ring_plan(N, C_in, C_out, Lo, K, pad, x_bf16 ? ldx : L, ldy, !x_bf16);
        if (rp.ok) return rp.G;
    }
    return bf16_fwd_stat_partials(N, C_out, Lo);
}

// time steps per workgroup tile of the ring kernel when a conv with bf16 tensors on both sides (C_red reduction channels,
// C_res result channels, L_out result positions, `pad` as the kernel sees it) takes it, 0 when the round-2 kernel runs
ECG_API int ecg_conv1d_bf16_ring_tile(int N, int C_red, int C_res, int L_out, int K, int pad, int ld_in, int ld_out) {
    const RingPlan rp = bf16_ring_plan(N, C_red, C_res, L_out, K, pad, ld_in, ld_out);
    return rp.ok ? rp.t_t : 0;
}

// Train-mode forward with bf16 ACTIVATION STORAGE: y is written as bf16 [N][C_out][ldy] (ldy >= Lo, even), the
// BatchNorm statistics partials ([C_out][P][2] = (sum, sum of squares), P from ecg_conv1d_fwd_bf16_yh_stat_partials) are taken
// over the rounded values.  The consumers are ecg_bn_stats_relu_pool_fwd_h / ecg_bn_stats_relu_pool_gap_fwd_yh and
// ecg_bn_relu_pool_bwd_h.
// x: fp32 [N][C_in][L] (x_bf16 == 0, ldx ignored) or bf16 [N][C_in][ldx] with rows zero-filled from L to ldx, ldx even,
// pad odd (x_bf16 != 0: the previous block's ecg_bn_stats_relu_pool_fwd_h wrote it that way).
ECG_API int ecg_conv1d_fwd_bf16_yh(const void *x, int x_bf16, int ldx, const void *wb_fwd, const float *bias,
                                   void *y_bf16, int ldy, float *stat_partials, int N, int C_in, int C_out, int L, int K,
                                   int pad, ecg_stream_t stream) {
    int rc = check_bf16_shape("conv1d_fwd_bf16_yh", N, C_in, C_out, L, K, pad);
    if (rc) return rc;
    ECG_REQUIRE(x && wb_fwd && y_bf16 && stat_partials, "conv1d_fwd_bf16_yh: null pointer");
    ECG_REQUIRE(bf16_fwd_supported(C_in, C_out, K, pad), "conv1d_fwd_bf16_yh: needs C_in %% 4 == 0, C_out %% 32 == 0");
    ECG_REQUIRE(ldy >= L + 2 * pad - K + 1 && ldy % 2 == 0 && (reinterpret_cast<uintptr_t>(y_bf16) & 3) == 0,
                "conv1d_fwd_bf16_yh: needs an even row stride >= Lo and a 4-byte aligned y");
    ECG_REQUIRE(!x_bf16 || (ldx >= L && ldx % 2 == 0 && (pad & 1) == 1 && (reinterpret_cast<uintptr_t>(x) & 3) == 0),
                "conv1d_fwd_bf16_yh: a bf16 x needs an even row stride >= L, odd pad and a 4-byte aligned base");
    return bf16_fwd_any(x, x_bf16 ? ldx : L, x_bf16 != 0, wb_fwd, bias, static_cast<float *>(y_bf16), ldy, true,
                        stat_partials, N, C_in, C_out, L, K, pad, as_stream(stream));
}

// Input gradient from a dY that is itself bf16 — [N][C_out][ldy], rows zero-filled from Lo to ldy, ldy even, as
// ecg_bn_relu_pool_bwd_h writes it — to dx as bf16 [N][C_in][ldx] (ldx even, >= L; the row padding is left unwritten or
// zeroed): the dp that the previous block's ecg_bn_relu_pool_bwd_h reads
ECG_API int ecg_conv1d_bwd_data_bf16hh(const void *dy_bf16, int ldy, const void *wb_bwd, void *dx_bf16, int ldx, int N,
                                       int C_in, int C_out, int L, int K, int pad, ecg_stream_t stream) {
    int rc = check_bf16_shape("conv1d_bwd_data_bf16hh", N, C_in, C_out, L, K, pad);
    if (rc) return rc;
    ECG_REQUIRE(dy_bf16 && wb_bwd && dx_bf16, "conv1d_bwd_data_bf16hh: null pointer");
    const int Lo = L + 2 * pad - K + 1, padb = K - 1 - pad;
    ECG_REQUIRE(bf16_fwd_supported(C_out, C_in, K, padb), "conv1d_bwd_data_bf16hh: needs C_out %% 4 == 0, C_in %% 32 == 0");
    ECG_REQUIRE(ldy >= Lo && ldy % 2 == 0 && (padb & 1) == 1 && (reinterpret_cast<uintptr_t>(dy_bf16) & 3) == 0,
                "conv1d_bwd_data_bf16hh: needs an even row stride >= Lo, odd K-1-pad and a 4-byte aligned dY");
    ECG_REQUIRE(ldx >= L && ldx % 2 == 0 && (reinterpret_cast<uintptr_t>(dx_bf16) & 3) == 0,
                "conv1d_bwd_data_bf16hh: needs an even dx row stride >= L and a 4-byte aligned dx");
    return bf16_fwd_any(dy_bf16, ldy, true, wb_bwd, nullptr, static_cast<float *>(dx_bf16), ldx, true, nullptr, N, C_out,
                        C_in, Lo, K, padb, as_stream(stream));
}

#ifdef ECG_STAMP
extern "C" __attribute__((visibility("default"))) int ecg_debug_set_stamp_buffer_bf16(unsigned long long *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(ecg::g_stamps_b), &buf, sizeof(buf));
}
#endif
