// conv1d_api.hip — C-ABI entry points for Conv1d and the shape dispatch between the
// MFMA implicit-GEMM kernels (conv1d_mfma.hip) and the direct VALU kernels (conv1d_direct.hip).
#include "common.h"

namespace ecg {
// conv1d_direct.hip
int direct_fwd_stat_partials(int N, int Lo);
int direct_fwd(const float *x, const float *wp, const float *bias, float *y, float *partials,
               int N, int Cin, int Cout, int L, int K, int pad, hipStream_t st);
size_t direct_wgrad_ws_floats(int N, int Cin, int Cout, int K);
int direct_wgrad(const float *dy, const float *x, float *dw, float *db, float *ws, int N, int Cin,
                 int Cout, int L, int K, int pad, hipStream_t st);
int pack_weights(const float *w, float *w_fwd, float *w_bwd, int Co, int Ci, int K,
                 hipStream_t st);
int pack_weights_grouped(const float *const *w, float *const *w_fwd, float *const *w_bwd, void *const *wb_fwd,
                         void *const *wb_bwd, const int *Co, const int *Ci, const int *K, int count, hipStream_t st);
// conv1d_mfma.hip
bool mfma_fwd_supported(int Cin, int Cout, int K, int pad);
int mfma_fwd_stat_partials(int N, int Cin, int Cout, int Lo);
int mfma_fwd(const float *x, int ldx, const float *wp, const float *bias, float *y, float *partials,
             int N, int Cin, int Cout, int L, int K, int pad, hipStream_t st);
int mfma_fwd_eval_pool(const float *x, const float *wp, const float *bias, const float *gamma,
                       const float *beta, const float *mean, const float *var, float eps, float *p,
                       int N, int Cin, int Cout, int L, int K, int pad, hipStream_t st, int gap);
bool mfma_fwd_eval_gap_supported(int Cin, int Cout, int L, int K, int pad);
bool mfma_wgrad_supported(int Cin, int Cout, int K, int pad);
size_t mfma_wgrad_ws_floats(int N, int Cin, int Cout, int L, int K, int pad);
bool mfma_wgrad_dma_supported(int Cin, int Cout, int K);
int mfma_multiplies_per_pair(int op, int Cin, int Cout, int K, int pad);
int mfma_wgrad(const float *dy, int ldy, const float *x, float *dw, float *db, float *ws, int N,
               int Cin, int Cout, int L, int K, int pad, hipStream_t st);

static int check_conv_shape(int N, int Cin, int Cout, int L, int K, int pad) {
    ECG_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && L > 0, "conv1d: N=%d C_in=%d C_out=%d L=%d must be > 0",
                N, Cin, Cout, L);
    ECG_REQUIRE(K >= 1 && K <= 31, "conv1d: kernel size %d outside [1,31]", K);
    ECG_REQUIRE(pad >= 0 && pad < K, "conv1d: padding %d outside [0,K)", pad);
    ECG_REQUIRE(L + 2 * pad - K + 1 > 0, "conv1d: empty output (L=%d K=%d pad=%d)", L, K, pad);
    ECG_REQUIRE(N <= 65535, "conv1d: N=%d exceeds grid.z limit 65535", N);
    return ECG_OK;
}
}  // namespace ecg

using namespace ecg;

ECG_API int ecg_conv1d_pack_weights(const float *w, float *w_fwd, float *w_bwd, int C_out,
                                    int C_in, int K, ecg_stream_t stream) {
    ECG_REQUIRE(w && (w_fwd || w_bwd), "pack_weights: null pointer");
    ECG_REQUIRE(C_out > 0 && C_in > 0 && K >= 1 && K <= 31, "pack_weights: bad shape");
    return pack_weights(w, w_fwd, w_bwd, C_out, C_in, K, as_stream(stream));
}

ECG_API int ecg_pack_weights_grouped(const float *const *w, float *const *w_fwd,
                                     float *const *w_bwd, const int *C_out, const int *C_in,
                                     const int *K, int count, ecg_stream_t stream) {
    ECG_REQUIRE(w && w_fwd && w_bwd && C_out && C_in && K, "pack_weights_grouped: null table");
    ECG_REQUIRE(count >= 1 && count <= 16, "pack_weights_grouped: count=%d outside [1,16]", count);
    for (int q = 0; q < count; ++q) {
        ECG_REQUIRE(w[q] && (w_fwd[q] || w_bwd[q]), "pack_weights_grouped: problem %d has null pointers", q);
        ECG_REQUIRE(C_out[q] > 0 && C_in[q] > 0 && K[q] >= 1 && K[q] <= 31,
                    "pack_weights_grouped: problem %d has a bad shape", q);
    }
    return pack_weights_grouped(w, w_fwd, w_bwd, nullptr, nullptr, C_out, C_in, K, count, as_stream(stream));
}

ECG_API int ecg_pack_weights_grouped_mixed(const float *const *w, float *const *w_fwd, float *const *w_bwd,
                                           void *const *wb_fwd, void *const *wb_bwd, const int *C_out,
                                           const int *C_in, const int *K, int count, ecg_stream_t stream) {
    ECG_REQUIRE(w && w_fwd && w_bwd && wb_fwd && wb_bwd && C_out && C_in && K, "pack_weights_grouped_mixed: null table");
    ECG_REQUIRE(count >= 1 && count <= 16, "pack_weights_grouped_mixed: count=%d outside [1,16]", count);
    for (int q = 0; q < count; ++q) {
        ECG_REQUIRE(w[q] && (w_fwd[q] || w_bwd[q] || wb_fwd[q] || wb_bwd[q]),
                    "pack_weights_grouped_mixed: problem %d has null pointers", q);
        ECG_REQUIRE(C_out[q] > 0 && C_in[q] > 0 && K[q] >= 1 && K[q] <= 31,
                    "pack_weights_grouped_mixed: problem %d has a bad shape", q);
        ECG_REQUIRE(K[q] <= 15 || !(wb_fwd[q] || wb_bwd[q]),
                    "pack_weights_grouped_mixed: problem %d asks for bf16 operands with K=%d > 15", q, K[q]);
    }
    return pack_weights_grouped(w, w_fwd, w_bwd, wb_fwd, wb_bwd, C_out, C_in, K, count, as_stream(stream));
}

ECG_API int ecg_conv1d_fwd_stat_partials(int N, int C_in, int C_out, int L, int K, int pad) {
    int Lo = L + 2 * pad - K + 1;
    if (mfma_fwd_supported(C_in, C_out, K, pad)) return mfma_fwd_stat_partials(N, C_in, C_out, Lo);
    return direct_fwd_stat_partials(N, Lo);
}

ECG_API int ecg_conv1d_fwd(const float *x, const float *w_fwd, const float *bias, float *y,
                           float *stat_partials, int N, int C_in, int C_out, int L, int K, int pad,
                           ecg_stream_t stream) {
    int rc = check_conv_shape(N, C_in, C_out, L, K, pad);
    if (rc) return rc;
    ECG_REQUIRE(x && w_fwd && y, "conv1d_fwd: null pointer");
    if (mfma_fwd_supported(C_in, C_out, K, pad))
        return mfma_fwd(x, L, w_fwd, bias, y, stat_partials, N, C_in, C_out, L, K, pad, as_stream(stream));
    return direct_fwd(x, w_fwd, bias, y, stat_partials, N, C_in, C_out, L, K, pad, as_stream(stream));
}

// Row stride the fused BatchNorm backward should give dY for this layer: rows padded to a
// multiple of 64 floats (pad zero-filled) let the weight-gradient kernel stream dY by LDS-DMA.
// Returns Lo (dense rows) when the shape is not served by the MFMA kernels that understand it.
ECG_API int ecg_conv1d_dy_row_stride(int N, int C_in, int C_out, int L, int K, int pad, int need_dx) {
    (void)N;
    const int Lo = L + 2 * pad - K + 1;
    if (Lo <= 0) return Lo;
    // the input gradient (when it is wanted at all: not for the first layer) must understand the stride too
    if (mfma_wgrad_dma_supported(C_in, C_out, K) && (!need_dx || mfma_fwd_supported(C_out, C_in, K, K - 1 - pad)))
        return cdiv(Lo, 64) * 64;
    return Lo;
}

ECG_API int ecg_conv1d_multiplies_per_output_pair(int op, int C_in, int C_out, int K, int pad) {
    if (op < 0 || op > 3 || K < 1) return 0;
    return mfma_multiplies_per_pair(op, C_in, C_out, K, pad);
}

// Input gradient = forward conv of dy with the tap-flipped, channel-transposed weights
// (w_bwd [K][C_out][C_in]) and padding K-1-pad: roles of C_in / C_out swap.
ECG_API int ecg_conv1d_bwd_data_ld(const float *dy, int ldy, const float *w_bwd, float *dx, int N,
                                   int C_in, int C_out, int L, int K, int pad, ecg_stream_t stream) {
    int rc = check_conv_shape(N, C_in, C_out, L, K, pad);
    if (rc) return rc;
    ECG_REQUIRE(dy && w_bwd && dx, "conv1d_bwd_data: null pointer");
    const int Lo = L + 2 * pad - K + 1, padb = K - 1 - pad;
    ECG_REQUIRE(ldy >= Lo, "conv1d_bwd_data: dY row stride %d < row length %d", ldy, Lo);
    if (mfma_fwd_supported(C_out, C_in, K, padb))
        return mfma_fwd(dy, ldy, w_bwd, nullptr, dx, nullptr, N, C_out, C_in, Lo, K, padb, as_stream(stream));
    ECG_REQUIRE(ldy == Lo, "conv1d_bwd_data: this shape needs dense dY rows (stride %d != %d); "
                "use the stride ecg_conv1d_dy_row_stride returns", ldy, Lo);
    return direct_fwd(dy, w_bwd, nullptr, dx, nullptr, N, C_out, C_in, Lo, K, padb, as_stream(stream));
}

ECG_API int ecg_conv1d_bwd_data(const float *dy, const float *w_bwd, float *dx, int N, int C_in,
                                int C_out, int L, int K, int pad, ecg_stream_t stream) {
    return ecg_conv1d_bwd_data_ld(dy, L + 2 * pad - K + 1, w_bwd, dx, N, C_in, C_out, L, K, pad, stream);
}

ECG_API size_t ecg_conv1d_bwd_weight_ws_floats(int N, int C_in, int C_out, int L, int K, int pad) {
    if (mfma_wgrad_supported(C_in, C_out, K, pad)) return mfma_wgrad_ws_floats(N, C_in, C_out, L, K, pad);
    return direct_wgrad_ws_floats(N, C_in, C_out, K);
}

ECG_API int ecg_conv1d_bwd_weight_bias_ld(const float *dy, int ldy, const float *x, float *dw,
                                          float *db, float *ws, int N, int C_in, int C_out, int L,
                                          int K, int pad, ecg_stream_t stream) {
    int rc = check_conv_shape(N, C_in, C_out, L, K, pad);
    if (rc) return rc;
    ECG_REQUIRE(dy && x && dw && ws, "conv1d_bwd_weight_bias: null pointer");
    const int Lo = L + 2 * pad - K + 1;
    ECG_REQUIRE(ldy >= Lo, "conv1d_bwd_weight_bias: dY row stride %d < row length %d", ldy, Lo);
    if (mfma_wgrad_supported(C_in, C_out, K, pad))
        return mfma_wgrad(dy, ldy, x, dw, db, ws, N, C_in, C_out, L, K, pad, as_stream(stream));
    ECG_REQUIRE(ldy == Lo, "conv1d_bwd_weight_bias: this shape needs dense dY rows (stride %d != %d); "
                "use the stride ecg_conv1d_dy_row_stride returns", ldy, Lo);
    return direct_wgrad(dy, x, dw, db, ws, N, C_in, C_out, L, K, pad, as_stream(stream));
}

ECG_API int ecg_conv1d_bwd_weight_bias(const float *dy, const float *x, float *dw, float *db,
                                       float *ws, int N, int C_in, int C_out, int L, int K, int pad,
                                       ecg_stream_t stream) {
    return ecg_conv1d_bwd_weight_bias_ld(dy, L + 2 * pad - K + 1, x, dw, db, ws, N, C_in, C_out, L, K,
                                         pad, stream);
}

ECG_API int ecg_conv1d_bn_relu_pool_eval_supported(int C_in, int C_out, int K, int pad) {
    return mfma_fwd_supported(C_in, C_out, K, pad) ? 1 : 0;
}

ECG_API int ecg_conv1d_bn_relu_pool_eval_fwd(const float *x, const float *w_fwd, const float *bias,
                                             const float *gamma, const float *beta,
                                             const float *running_mean, const float *running_var,
                                             float eps, float *p, int N, int C_in, int C_out, int L,
                                             int K, int pad, ecg_stream_t stream) {
    int rc = check_conv_shape(N, C_in, C_out, L, K, pad);
    if (rc) return rc;
    ECG_REQUIRE(x && w_fwd && gamma && beta && running_mean && running_var,
                "conv1d_bn_relu_pool_eval_fwd: null pointer");
    ECG_REQUIRE(mfma_fwd_supported(C_in, C_out, K, pad),
                "conv1d_bn_relu_pool_eval_fwd: shape not covered by the fused kernel "
                "(query ecg_conv1d_bn_relu_pool_eval_supported and use the unfused sequence)");
    const int Lo = L + 2 * pad - K + 1;
    if (Lo / 2 == 0) return ECG_OK;
    ECG_REQUIRE(p, "conv1d_bn_relu_pool_eval_fwd: null output");
    return mfma_fwd_eval_pool(x, w_fwd, bias, gamma, beta, running_mean, running_var, eps, p, N,
                              C_in, C_out, L, K, pad, as_stream(stream), 0);
}

ECG_API int ecg_conv1d_bn_relu_pool_gap_eval_supported(int C_in, int C_out, int L, int K, int pad) {
    return mfma_fwd_eval_gap_supported(C_in, C_out, L, K, pad) ? 1 : 0;
}

ECG_API int ecg_conv1d_bn_relu_pool_gap_eval_fwd(const float *x, const float *w_fwd, const float *bias,
                                                 const float *gamma, const float *beta,
                                                 const float *running_mean, const float *running_var,
                                                 float eps, float *g, int N, int C_in, int C_out, int L,
                                                 int K, int pad, ecg_stream_t stream) {
    int rc = check_conv_shape(N, C_in, C_out, L, K, pad);
    if (rc) return rc;
    ECG_REQUIRE(x && w_fwd && gamma && beta && running_mean && running_var && g,
                "conv1d_bn_relu_pool_gap_eval_fwd: null pointer");
    ECG_REQUIRE(mfma_fwd_eval_gap_supported(C_in, C_out, L, K, pad),
                "conv1d_bn_relu_pool_gap_eval_fwd: shape not covered (the conv output row must fit one time tile; "
                "query ecg_conv1d_bn_relu_pool_gap_eval_supported)");
    return mfma_fwd_eval_pool(x, w_fwd, bias, gamma, beta, running_mean, running_var, eps, g, N,
                              C_in, C_out, L, K, pad, as_stream(stream), 1);
}
