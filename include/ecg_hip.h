/*
 * ecg_hip.h — C ABI of libecg_hip.so: the MI355X (gfx950) kernels behind the
 * ptbxl-multimodal 1D-CNN train/eval path.
 *
 * The reference (cyu0330/ptbxl-multimodal) has no FFI of its own: its boundary is the
 * Python API of src/models + src/training, and below that stock torch.nn modules
 * dispatching to ATen.  Each entry point here replaces the ATen work dispatched by one
 * reference call site (cited per function as <reference file>:<line>); the Python
 * host (ptbxl-multimodal_amd/ecg_hip/) binds them with ctypes from inside
 * torch.autograd.Function.forward/backward.  See INTEGRATION.md for the binding.
 *
 * Conventions
 *   - All tensors are device pointers to contiguous float32 unless noted; activations
 *     are NCL.  The library never allocates or frees tensor memory: outputs and
 *     workspaces are caller-owned (query the *_ws_floats() helpers for sizes).
 *   - Every entry point returns 0 on success and a nonzero ECG_E* code otherwise; the
 *     message is available from ecg_last_error() (thread-local).  No exception crosses
 *     the ABI.  Shapes are validated on the host before any launch.
 *   - Entry points are re-entrant, keep no global mutable state (the only thing cached is
 *     the CU count of a device), read no environment variable and launch on the
 *     stream handed in (a hipStream_t passed as void*; NULL = the null stream).  They
 *     never synchronise the device.  Forward is called on the Python main thread,
 *     backward on torch's autograd engine thread.
 *   - Conv1d support envelope: stride 1, dilation 1, groups 1, 1 <= K <= 31,
 *     0 <= pad < K (the reference uses K=15, pad=7 only: src/models/ecg_cnn.py:13).
 */
#ifndef ECG_HIP_H
#define ECG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ecg_stream_t; /* hipStream_t */

enum {
    ECG_OK = 0,
    ECG_EINVAL = 1,   /* bad shape / null pointer / unsupported configuration */
    ECG_ELAUNCH = 2,  /* hipGetLastError() after a launch */
    ECG_ENODEV = 3    /* no gfx950 device visible */
};

/* ABI version, 100*major + minor. */
int ecg_version(void);
/* Message of the last nonzero return on the calling thread ("" if none). */
const char *ecg_last_error(void);
/* 0 if device 0..n-1 is a gfx950 the code object can run on, ECG_ENODEV otherwise. */
int ecg_check_device(void);

/* ------------------------------------------------------------------------------------
 * Conv1d — replaces aten::convolution / aten::convolution_backward dispatched by
 * ConvBlock.net[0] = nn.Conv1d(in,out,k,padding=k//2): src/models/ecg_cnn.py:13,
 * src/models/ecg_multimodal.py:9 (backward via loss.backward(), src/training/loop.py:33).
 * ---------------------------------------------------------------------------------- */

/* Packed weights: w [C_out][C_in][K] (state_dict layout) ->
 *   w_fwd [K][C_in][C_out]            (forward operand)
 *   w_bwd [K][C_out][C_in], tap-flipped: w_bwd[k][co][ci] = w[co][ci][K-1-k]  (input-grad operand)
 * Either output may be NULL.  Each holds C_out*C_in*K floats. */
int ecg_conv1d_pack_weights(const float *w, float *w_fwd, float *w_bwd,
                            int C_out, int C_in, int K, ecg_stream_t stream);

/* count (<= 16) packs in ONE launch; problem q == ecg_conv1d_pack_weights(w[q], w_fwd[q], w_bwd[q],
 * C_out[q], C_in[q], K[q]).  A Linear weight transpose is the K = 1 case (w_fwd[ci][co] = w[co][ci]).
 * All six arguments are HOST arrays of length count; w_fwd[q] or w_bwd[q] may be NULL. */
int ecg_pack_weights_grouped(const float *const *w, float *const *w_fwd, float *const *w_bwd,
                             const int *C_out, const int *C_in, const int *K, int count,
                             ecg_stream_t stream);

/* The same launch, also writing the bf16 MFMA operands of the opt-in mixed-precision mode: problem q additionally
 * == ecg_conv1d_pack_weights_bf16(w[q], wb_fwd[q], wb_bwd[q], ...) (K[q] <= 15 when either is given).  Any of the
 * four destinations of a problem may be NULL, not all.  One launch per train step instead of one per conv layer
 * plus the fp32 one (the reference has no counterpart: ATen reads nn.Conv1d.weight as it stands,
 * src/models/ecg_cnn.py:13). */
int ecg_pack_weights_grouped_mixed(const float *const *w, float *const *w_fwd, float *const *w_bwd,
                                   void *const *wb_fwd, void *const *wb_bwd, const int *C_out,
                                   const int *C_in, const int *K, int count, ecg_stream_t stream);

/* Number P of per-channel (sum, sum-of-squares) partials ecg_conv1d_fwd writes per output
 * channel for this shape; stat_partials must hold C_out*P*2 floats. */
int ecg_conv1d_fwd_stat_partials(int N, int C_in, int C_out, int L, int K, int pad);

/* y[n,co,t] = bias[co] + sum_ci sum_k w[co,ci,k] * x[n,ci,t+k-pad]   (zero padding)
 * x [N][C_in][L], w_fwd packed as above, bias [C_out] or NULL, y [N][C_out][Lo],
 * Lo = L + 2*pad - K + 1.
 * stat_partials (nullable): [C_out][P][2] per-producer (sum y, sum y^2) over the producer's
 * outputs, consumed by ecg_bn_finalize (train-mode BatchNorm statistics fused in the epilogue). */
int ecg_conv1d_fwd(const float *x, const float *w_fwd, const float *bias, float *y,
                   float *stat_partials, int N, int C_in, int C_out, int L, int K, int pad,
                   ecg_stream_t stream);

/* dx[n,ci,s] = sum_co sum_k dy[n,co,s-k+pad] * w[co,ci,k];  dy [N][C_out][Lo], dx [N][C_in][L] */
int ecg_conv1d_bwd_data(const float *dy, const float *w_bwd, float *dx,
                        int N, int C_in, int C_out, int L, int K, int pad, ecg_stream_t stream);

/* Row-padded dY.  The backward of a ConvBlock hands dY from the BatchNorm backward to the two
 * conv gradients; it is an internal tensor, so its layout is ours to choose.  When
 * ecg_conv1d_dy_row_stride returns ldy > Lo (a multiple of 64 floats), give dY that row stride
 * ([N][C_out][ldy], zeros in [Lo, ldy) — ecg_bn_relu_pool_bwd_ld / _gap_bwd_ld write them) and the
 * weight gradient streams it global -> LDS by DMA instead of through registers.  The _ld entry
 * points accept any ldy >= Lo where the MFMA kernels apply and require ldy == Lo elsewhere; the
 * plain entry points are the ldy == Lo case.  need_dx: whether ecg_conv1d_bwd_data_ld will be called on
 * this dY too (not for the first layer) — the stride must then be one the input-gradient kernel reads. */
int ecg_conv1d_dy_row_stride(int N, int C_in, int C_out, int L, int K, int pad, int need_dx);

/* Multiplies the fp32 kernels issue per output PAIR (2m, 2m+1) and (c_in, c_out): 30 = the direct form (2 x 15 taps), 23 = the
 * two-phase fast-FIR split (round 5): three half-rate products over the even / odd taps and samples, the third on DIFFERENCES
 * (w[2j] - w[2j-1]) * (x[2m+2j] - x[2m+2j+1]) — exact for neighbouring samples of like sign and magnitude — combined in fp32 in
 * the epilogue (forward, input gradient) or in the double-precision slab reduce (weight gradient).  Same operands, same
 * results to a few ulp of the accumulated magnitude (tests/test_gpu_ops.py: error against float64 no larger than the CPU fp32
 * path's); measurement code uses this to report the matrix-pipe utilisation beside the algorithmic rate.
 * op: 0 = ecg_conv1d_fwd with the statistics epilogue (training), 1 = ecg_conv1d_bwd_data[_ld] and ecg_conv1d_fwd without
 * statistics, 2 = ecg_conv1d_bwd_weight_bias_ld on row-padded dY, 3 = the one-launch inference blocks
 * (ecg_conv1d_bn_relu_pool[_gap]_eval_fwd). */
int ecg_conv1d_multiplies_per_output_pair(int op, int C_in, int C_out, int K, int pad);
int ecg_conv1d_bwd_data_ld(const float *dy, int ldy, const float *w_bwd, float *dx,
                           int N, int C_in, int C_out, int L, int K, int pad, ecg_stream_t stream);
int ecg_conv1d_bwd_weight_bias_ld(const float *dy, int ldy, const float *x, float *dw, float *db,
                                  float *ws, int N, int C_in, int C_out, int L, int K, int pad,
                                  ecg_stream_t stream);

/* Workspace (floats) for ecg_conv1d_bwd_weight_bias. */
size_t ecg_conv1d_bwd_weight_ws_floats(int N, int C_in, int C_out, int L, int K, int pad);
/* dw[co,ci,k] = sum_n sum_t dy[n,co,t]*x[n,ci,t+k-pad] (state_dict layout [C_out][C_in][K]);
 * db[co] = sum_n sum_t dy[n,co,t] (nullable).  Deterministic: split partial slabs in ws are
 * summed in a fixed order, no float atomics. */
int ecg_conv1d_bwd_weight_bias(const float *dy, const float *x, float *dw, float *db, float *ws,
                               int N, int C_in, int C_out, int L, int K, int pad,
                               ecg_stream_t stream);

/* ---- mixed precision (opt-in; BASELINE.json config 5: AF binary 12x5000, bf16) ----------------
 * ONE form (round 5): a training ConvBlock runs its three convs with bf16 operands on the matrix cores (fp32 accumulate)
 * and keeps every tensor BETWEEN its kernels as bf16 rows [N][C][ld] — what torch.autocast stores too; the BatchNorm passes
 * are HBM-bound and these are their operands:
 *   y   conv output        bf16 [N][C][ldy], ldy % 8 == 0 (ecg_conv1d_fwd_bf16_yh; the BatchNorm statistics are taken over
 *                          the ROUNDED values, i.e. over the tensor the passes read)
 *   p   pooled activation  bf16 [N][C][ldp], rows zero-filled from L/2 to ldp (ecg_bn_stats_relu_pool_fwd_h) — read by the
 *                          next block's ecg_conv1d_fwd_bf16_yh (x_bf16 != 0) and by its weight gradient
 *   dY  conv output grad   bf16 [N][C][ldt], rows zero-filled to a multiple of 128 (ecg_bn_relu_pool_bwd_h) — read by the
 *                          weight gradient (ecg_conv1d_bwd_weight_bias_bf16_ncl) and the input gradient (..._bf16hh)
 *   dp  gradient of p      bf16 [N][C][ldp] (ecg_conv1d_bwd_data_bf16hh of the next block) — read by ecg_bn_relu_pool_bwd_h
 * Parameters, statistics, parameter gradients, the network input (fp32, read by block 0 and rounded while it is staged) and
 * the tail stay fp32.  Everything that does not fit (eval / frozen BatchNorm, other kernel sizes or channel counts, an fp32
 * input that needs an input gradient) runs the fp32 entry points above: there is no second bf16 form.  (Rounds 2-4 also
 * shipped fp32-activation variants and a sample-on-K weight gradient with its "n16" operand copies; removed in round 5.)
 * Packed weights are bf16: wb_fwd [ceil(C_in/16)][K][C_out][16], wb_bwd [ceil(C_out/16)][K][C_in][16]
 * (tap-flipped), ecg_conv1d_bf16_packed_elems(C_reduce, C_result, K) 2-byte elements each.
 * K must be 15; forward needs C_in % 4 == 0 and C_out % 32 == 0, the input-grad the same with the roles swapped, the weight
 * gradient K == 15, pad == 7, C_out % 32 == 0.  ecg_conv1d_bf16_supported returns a bit mask: 1 = forward, 2 = input-grad,
 * 4 = weight-grad.  Same reference call sites as the fp32 entry points: src/models/ecg_cnn.py:13-16 and their autograd. */
int ecg_conv1d_bf16_supported(int C_in, int C_out, int K, int pad);
size_t ecg_conv1d_bf16_packed_elems(int C_reduce, int C_result, int K);
int ecg_conv1d_pack_weights_bf16(const float *w, void *wb_fwd, void *wb_bwd, int C_out, int C_in,
                                 int K, ecg_stream_t stream);
/* Train-mode forward: x fp32 [N][C_in][L] (x_bf16 == 0, ldx ignored) or bf16 [N][C_in][ldx] with rows zero-filled from L to
 * ldx, ldx even, pad odd (x_bf16 != 0) -> y_bf16 [N][C_out][ldy] (+ bias) and the BatchNorm statistics partials
 * [C_out][P][2] = (sum, sum of squares) over the rounded values. */
int ecg_conv1d_fwd_bf16_yh(const void *x, int x_bf16, int ldx, const void *wb_fwd, const float *bias, void *y_bf16,
                           int ldy, float *stat_partials, int N, int C_in, int C_out, int L, int K, int pad,
                           ecg_stream_t stream);
/* How many (sum, sum^2) partials per channel ecg_conv1d_fwd_bf16_yh writes for exactly these arguments — ALWAYS size
 * stat_partials with this query.  Long rows with bf16 on both sides (x_bf16 != 0, ldy % 8 == 0) take the round-3 ring kernel
 * (csrc/conv1d_bf16_ring.hip: 640 / 1280 time steps per workgroup, weights through an LDS-DMA ring, transposed
 * accumulators), and so does the fp32 network input (x_bf16 == 0) of at most 16 channels on rows of even length where
 * 512-step tiles pad no more than 256-step ones (one chunk per tile, weights resident, two workgroups per CU); everything
 * else the round-2 kernel of csrc/conv1d_mfma_bf16.hip, with other workgroup counts. */
int ecg_conv1d_fwd_bf16_yh_stat_partials(int N, int C_in, int C_out, int L, int K, int pad, int x_bf16, int ldx, int ldy);
/* Which kernel a conv with bf16 tensors on both sides takes (ecg_conv1d_fwd_bf16_yh with x_bf16 != 0: C_red = C_in,
 * C_res = C_out, pad; ecg_conv1d_bwd_data_bf16hh: C_red = C_out, C_res = C_in, pad = K-1-pad): the ring kernel's time
 * steps per workgroup tile (640 / 1280), or 0 = the round-2 kernel of conv1d_mfma_bf16.hip.  Tests and profiles. */
int ecg_conv1d_bf16_ring_tile(int N, int C_red, int C_res, int L_out, int K, int pad, int ld_in, int ld_out);
/* The BatchNorm passes on bf16 [N][C][ld] tensors (csrc/bn_relu_pool_h.hip; 16 bytes per lane in and out):
 *   ecg_bn_stats_relu_pool_fwd_h: statistics combine + BN + ReLU + MaxPool(2): y_bf16 [N][C][ldy] -> p_bf16 [N][C][ldp], rows
 *     zero-filled from L/2 to ldp (ldy, ldp multiples of 8); mean / invstd are outputs, running statistics and the counter
 *     are updated as by ecg_bn_stats_relu_pool_fwd.
 *   ecg_bn_stats_relu_pool_gap_fwd_yh: the same with the global average pool of the last block folded in:
 *     y_bf16 [N][C][ldy] (ldy even) -> out fp32 [N][C] (csrc/bn_relu_pool.hip).
 *   ecg_bn_relu_pool_bwd_h: reduction pass + dx pass: dp bf16 [N][C][ldp] (dp_kind 0), fp32 dg [N][C] of the fused global
 *     average pool (1) or fp32 dp [N][C][ldp] (2) -> dy_bf16 [N][C][ldy]
 *     with rows zero-filled from L to ldy (give ldy = ecg_conv1d_bf16_tk_dy_stride(L)), dgamma, dbeta; ws from
 *     ecg_bn_relu_pool_bwd_ws_floats.  Same reference call site: autograd of ConvBlock.net[1..3], src/models/ecg_cnn.py:14-16. */
int ecg_bn_stats_relu_pool_fwd_h(const float *stat_partials, int P, long long count, float *running_mean,
                                 float *running_var, long long *num_batches_tracked, float momentum, float eps,
                                 const void *y_bf16, int ldy, const float *gamma, const float *beta, float *mean,
                                 float *invstd, void *p_bf16, int ldp, int N, int C, int L, ecg_stream_t stream);
int ecg_bn_stats_relu_pool_gap_fwd_yh(const float *stat_partials, int P, long long count, float *running_mean,
                                      float *running_var, long long *num_batches_tracked, float momentum, float eps,
                                      const void *y_bf16, int ldy, const float *gamma, const float *beta, float *mean,
                                      float *invstd, float *out, int N, int C, int L, ecg_stream_t stream);
int ecg_bn_relu_pool_bwd_h(const void *y_bf16, int ldyy, const void *dp, int dp_kind, int ldp, const float *gamma,
                           const float *beta, const float *mean, const float *invstd, void *dy_bf16, int ldy,
                           float *dgamma, float *dbeta, float *ws, int N, int C, int L, int train,
                           ecg_stream_t stream);
/* Input gradient from the bf16 dY (ldy even, K-1-pad odd) to dx as bf16 [N][C_in][ldx] (ldx even, >= L): the dp the previous
 * block's ecg_bn_relu_pool_bwd_h reads. */
int ecg_conv1d_bwd_data_bf16hh(const void *dy_bf16, int ldy, const void *wb_bwd, void *dx_bf16, int ldx, int N,
                               int C_in, int C_out, int L, int K, int pad, ecg_stream_t stream);
/* Weight + bias gradient on the bf16 tensors the other two convs read — TIME on the MFMA's K axis (round 4,
 * csrc/conv1d_wgrad_bf16_tk.hip): dy_bf16 [N][C_out][ldy] with ldy = ecg_conv1d_bf16_tk_dy_stride(Lo) (rows zero-filled to a
 * multiple of 128), x either bf16 [N][C_in][ldx] (x_is_bf16 != 0: ldx % 8 == 0, rows zero-filled past L — the previous
 * block's pooled activation as ecg_bn_stats_relu_pool_fwd_h writes it) or the fp32 network input [N][C_in][ldx] (L % 8 == 0),
 * rounded to bf16 while it is staged.  K == 15, pad == 7, C_out % 32 == 0 (ecg_conv1d_bf16_tk_supported).  Exact on the
 * bf16-rounded operands up to fp32 accumulation order; split partial slabs in ws are summed in a fixed order.  Same
 * reference call site as ecg_conv1d_bwd_weight_bias. */
int ecg_conv1d_bf16_tk_supported(int C_in, int C_out, int K, int pad);
int ecg_conv1d_bf16_tk_dy_stride(int Lo);
size_t ecg_conv1d_bwd_weight_bf16_ncl_ws_floats(int N, int C_in, int C_out, int L, int K, int pad);
int ecg_conv1d_bwd_weight_bias_bf16_ncl(const void *dy_bf16, int ldy, const void *x, int x_is_bf16, int ldx,
                                        float *dw, float *db, float *ws, int N, int C_in, int C_out, int L,
                                        int K, int pad, ecg_stream_t stream);

/* ------------------------------------------------------------------------------------
 * BatchNorm1d / ReLU / MaxPool1d(2) — ConvBlock.net[1..3]: src/models/ecg_cnn.py:14-16.
 * ---------------------------------------------------------------------------------- */

/* Standalone statistics partials of y [N][C][L] in the same [C][P][2] layout
 * (used when the conv epilogue did not produce them). Returns P via the helper. */
int ecg_bn_stat_partials_count(int N, int C, int L);
int ecg_bn_stat_partials(const float *y, float *stat_partials, int N, int C, int L,
                         ecg_stream_t stream);

/* Train-mode statistics from partials (summed in double, fixed order):
 * mean, biased var -> invstd = 1/sqrt(var+eps);  running_mean/var (nullable) updated with
 * `momentum` and the UNBIASED variance;  *num_batches_tracked (nullable, device int64) += 1. */
int ecg_bn_finalize(const float *stat_partials, int P, long long count,
                    float *mean, float *invstd, float *running_mean, float *running_var,
                    long long *num_batches_tracked, int C, float momentum, float eps,
                    ecg_stream_t stream);

/* invstd[c] = 1/sqrt(var[c] + eps)  (eval mode: from running_var) */
int ecg_bn_invstd(const float *var, float *invstd, int C, float eps, ecg_stream_t stream);

/* p[n,c,j] = max(0, max(a[2j], a[2j+1])),  a = (y-mean)*(invstd*gamma)+beta,  Lp = L/2 */
int ecg_bn_relu_pool_fwd(const float *y, const float *gamma, const float *beta,
                         const float *mean, const float *invstd, float *p,
                         int N, int C, int L, ecg_stream_t stream);
/* ecg_bn_finalize + BN-apply + ReLU + MaxPool(2) in ONE launch: the statistics combine is folded into the streaming
 * pass (every workgroup re-derives mean / invstd of its channel from the P partials — same arithmetic and bits as
 * ecg_bn_finalize — one workgroup per channel stores them and updates running statistics / counter).  mean and
 * invstd are OUTPUTS.  mode 0: out = p [N][C][L/2]; mode 1: + global average pool, out = g [N][C]. */
int ecg_bn_stats_relu_pool_fwd(const float *stat_partials, int P, long long count, float *running_mean,
                               float *running_var, long long *num_batches_tracked, float momentum, float eps,
                               const float *y, const float *gamma, const float *beta, float *mean,
                               float *invstd, float *out, int N, int C, int L, int mode, ecg_stream_t stream);

/* ONE-LAUNCH form of ecg_bn_relu_pool_bwd_ld / ecg_bn_relu_pool_gap_bwd_ld (same reference call site: autograd of
 * ConvBlock.net[1..3], src/models/ecg_cnn.py:14-16): when the (dp, y) slice of a block fits the register file of the
 * device, C x S <= #CUs workgroups load their slice once, the S workgroups of a channel exchange their partial sums, and
 * dY is written from registers (one read of the operands instead of two, one launch instead of two).
 *   ..._splits(N, C, L, ldy)          S >= 1 when the shape takes this form on the current device, 0 when it does not
 *                                     (then call the two-pass entry points).  The CALLER decides which form to use.
 *   ..._counter_uints(N, C, L, ldy)   uint32 words of `counters` the launch needs (0 when S <= 1).
 *   counters                          CALLER-OWNED exchange words (8-byte aligned): all zero before the first launch, left
 *                                     all zero by every launch, so one buffer serves any sequence of launches that are
 *                                     ordered on a stream; launches that may run CONCURRENTLY need separate buffers.  The
 *                                     library allocates nothing and keeps no state between calls.
 *   The exchange: each partial sum travels with its own "valid" tag in one 64-bit word (a single agent-scope atomic store;
 *   pollers use agent-scope atomic loads), so no ordering between different locations is assumed — the self-reset included:
 *   a workgroup counts itself out only after its own two words have been SEEN tagged (by its pollers, or, on the
 *   self-service path, read back by the publishing thread), so the last one out clears words that are all in place.  The
 *   wait is bounded:
 *   after `spin_polls` polls (< 0: the default, about 1 ms; 0: never wait) a workgroup recomputes its siblings' sums from
 *   memory itself, bit for bit what they publish — a co-tenant that keeps siblings off the device (another process, a
 *   communication kernel waiting for a late peer) costs time, never correctness.  Use the two-pass form while collectives
 *   run on the device during backward (ecg_hip.functional.declare_backward_collectives).
 *   gap != 0: dp is dg [N][C] (the global-average-pool form).  The two forms associate the partial sums differently
 *   (both deterministic). */
int ecg_bn_relu_pool_bwd_one_launch_splits(int N, int C, int L, int ldy);
size_t ecg_bn_relu_pool_bwd_one_launch_counter_uints(int N, int C, int L, int ldy);
int ecg_bn_relu_pool_bwd_one_launch(const float *y, const float *dp, const float *gamma, const float *beta,
                                    const float *mean, const float *invstd, float *dy, int ldy,
                                    float *dgamma, float *dbeta, uint32_t *counters, int N, int C, int L,
                                    int train, int gap, int spin_polls, ecg_stream_t stream);

size_t ecg_bn_relu_pool_bwd_ws_floats(int N, int C, int L);
/* Backward of the fused tail: dp [N][C][L/2] -> dy [N][C][L], dgamma[C], dbeta[C].
 * Arg-max and ReLU mask are recomputed from y (first element wins a tie, as max_pool1d).
 * train != 0: batch-statistics backward (native_batch_norm_backward);  train == 0: dy = da*gamma*invstd. */
int ecg_bn_relu_pool_bwd(const float *y, const float *dp, const float *gamma, const float *beta,
                         const float *mean, const float *invstd,
                         float *dy, float *dgamma, float *dbeta, float *ws,
                         int N, int C, int L, int train, ecg_stream_t stream);
/* Same, dy rows at stride ldy >= L with the pad [L, ldy) zero-filled. */
int ecg_bn_relu_pool_bwd_ld(const float *y, const float *dp, const float *gamma, const float *beta,
                            const float *mean, const float *invstd,
                            float *dy, int ldy, float *dgamma, float *dbeta, float *ws,
                            int N, int C, int L, int train, ecg_stream_t stream);

/* Inference: a whole ConvBlock in ONE launch — Conv1d, eval-mode BatchNorm1d (running statistics)
 * folded into the epilogue, ReLU and MaxPool1d(2); only the pooled activation p [N][C_out][Lo/2]
 * is written (reference test path scripts/06_ecg_baseline_test.py:69-106).  Covered shapes:
 * ecg_conv1d_bn_relu_pool_eval_supported() != 0 (K = 15, C_in % 4 == 0, C_out % 32 == 0). */
int ecg_conv1d_bn_relu_pool_eval_supported(int C_in, int C_out, int K, int pad);
int ecg_conv1d_bn_relu_pool_eval_fwd(const float *x, const float *w_fwd, const float *bias,
                                     const float *gamma, const float *beta,
                                     const float *running_mean, const float *running_var,
                                     float eps, float *p, int N, int C_in, int C_out, int L,
                                     int K, int pad, ecg_stream_t stream);
/* The same with the global average pool behind it (last block of the backbone): g [N][C_out] =
 * mean_j max(0, max(a[2j], a[2j+1])), nothing else is written.  Covered when the conv output row fits
 * one time tile of the kernel (Lo <= 126 for C_out % 64 == 0, <= 254 otherwise — the tile distance of the fast-FIR kernel:
 * 12x1000 windows end at 125; longer windows use ecg_conv1d_fwd + ecg_bn_relu_pool_gap_fwd). */
int ecg_conv1d_bn_relu_pool_gap_eval_supported(int C_in, int C_out, int L, int K, int pad);
int ecg_conv1d_bn_relu_pool_gap_eval_fwd(const float *x, const float *w_fwd, const float *bias,
                                         const float *gamma, const float *beta,
                                         const float *running_mean, const float *running_var,
                                         float eps, float *g, int N, int C_in, int C_out, int L,
                                         int K, int pad, ecg_stream_t stream);

/* Last block of the backbone fused with AdaptiveAvgPool1d(1) (src/models/ecg_cnn.py:46,62):
 * g[n,c] = mean_j max(0, max(a[2j], a[2j+1])) — the pooled tensor is never materialised.
 * Backward takes dg [N][C] (gradient of g); same workspace as ecg_bn_relu_pool_bwd. */
int ecg_bn_relu_pool_gap_fwd(const float *y, const float *gamma, const float *beta,
                             const float *mean, const float *invstd, float *g,
                             int N, int C, int L, ecg_stream_t stream);
int ecg_bn_relu_pool_gap_bwd(const float *y, const float *dg, const float *gamma, const float *beta,
                             const float *mean, const float *invstd,
                             float *dy, float *dgamma, float *dbeta, float *ws,
                             int N, int C, int L, int train, ecg_stream_t stream);
int ecg_bn_relu_pool_gap_bwd_ld(const float *y, const float *dg, const float *gamma,
                                const float *beta, const float *mean, const float *invstd,
                                float *dy, int ldy, float *dgamma, float *dbeta, float *ws,
                                int N, int C, int L, int train, ecg_stream_t stream);

/* Unfused leaves (used when a caller hooks an inner module, e.g. Grad-CAM on net[0]:
 * scripts/00_demo_inference.py:36-37). */
int ecg_bn_apply_fwd(const float *y, const float *gamma, const float *beta, const float *mean,
                     const float *invstd, float *out, int N, int C, int L, ecg_stream_t stream);
size_t ecg_bn_bwd_ws_floats(int N, int C, int L);
int ecg_bn_bwd(const float *y, const float *dout, const float *gamma, const float *mean,
               const float *invstd, float *dy, float *dgamma, float *dbeta, float *ws,
               int N, int C, int L, int train, ecg_stream_t stream);
int ecg_relu_fwd(const float *x, float *out, size_t n, ecg_stream_t stream);
int ecg_relu_bwd(const float *out, const float *dout, float *dx, size_t n, ecg_stream_t stream);
int ecg_maxpool2_fwd(const float *x, float *p, int rows, int L, ecg_stream_t stream);
int ecg_maxpool2_bwd(const float *x, const float *dp, float *dx, int rows, int L, ecg_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Tail — AdaptiveAvgPool1d(1), Linear, FiLM, BCE-with-logits.
 * ---------------------------------------------------------------------------------- */

/* g[r] = mean_t p[r][t], r in [0, rows)  — src/models/ecg_cnn.py:46,62 */
int ecg_gap_fwd(const float *p, float *g, int rows, int L, ecg_stream_t stream);
int ecg_gap_bwd(const float *dg, float *dp, int rows, int L, ecg_stream_t stream);

/* y[m,o] = act(b[o] + sum_i x[m,i]*w[o,i]);  act = ReLU if relu else identity.
 * nn.Linear: src/models/ecg_cnn.py:47,50; src/models/ecg_multimodal.py:35,52-55,85-86 */
int ecg_linear_fwd(const float *x, const float *w, const float *b, float *y,
                   int M, int In, int Out, int relu, ecg_stream_t stream);
size_t ecg_linear_bwd_ws_floats(int M, int In, int Out);
/* dy is the gradient w.r.t. y (post-activation); y is only read when relu != 0.
 * dx, dw, db are each nullable. */
int ecg_linear_bwd(const float *x, const float *w, const float *y, const float *dy,
                   float *dx, float *dw, float *db, float *ws,
                   int M, int In, int Out, int relu, ecg_stream_t stream);

/* zc = (1 + tanh(film[:, :F])) * z + film[:, F:]  — src/models/ecg_multimodal.py:92-96 */
int ecg_film_fwd(const float *z, const float *film, float *zc, int M, int F, ecg_stream_t stream);
int ecg_film_bwd(const float *z, const float *film, const float *dzc, float *dz, float *dfilm,
                 int M, int F, ecg_stream_t stream);

/* loss[0] = mean(max(x,0) - x*t + log1p(exp(-|x|)))  — src/training/loop.py:32, loop_demo.py:10,33
 * dx (nullable) = (sigmoid(x) - t) / numel  (the gradient for d loss = 1).
 * running_sum (nullable, device double): running_sum[0] += loss * weight — the epoch-loss
 * bookkeeping of the loops (loop.py:36 weight = batch size, loop_demo.py:38 weight = 1). */
int ecg_bce_logits_fwd(const float *x, const float *target, float *loss, float *dx,
                       int numel, double *running_sum, double weight, ecg_stream_t stream);
/* prob = sigmoid(x) — src/training/loop.py:63 */
int ecg_sigmoid_fwd(const float *x, float *prob, size_t n, ecg_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused tail — everything after the global average pool in three launches.
 * Dimensions: M samples; F0 = backbone width (256); F = feat_dim; D = demo_dim (5);
 * H1 = first demographic layer (64); H = demo_hidden_dim; C = num_labels.
 * ---------------------------------------------------------------------------------- */

/* wT[c][r] = w[r][c]  (w [rows][cols]).  ecg_tail_fwd consumes Linear weights transposed. */
int ecg_transpose(const float *w, float *wT, int rows, int cols, ecg_stream_t stream);

/* z = g Wp^T + bp                                   (ECGCNN.proj / ECGBackbone.proj)
 * xd != NULL (ECGMultimodal.forward, src/models/ecg_multimodal.py:88-99):
 *   h1 = relu(xd W0^T + b0); h2 = relu(h1 W2^T + b2); film = h2 Wf^T + bf;
 *   zc = (1 + tanh(film[:, :F])) * z + film[:, F:];  logits = zc Wh^T + bh
 * xd == NULL (ECGCNN.forward, src/models/ecg_cnn.py:63-64):  logits = z Wh^T + bh
 * WpT [F0][F] and WfT [H][2F] are TRANSPOSED weights (ecg_transpose); the small W0 [H1][D],
 * W2 [H][H1] and Wh [C][F] are in state_dict layout.
 * Outputs z [M][F], logits [M][C] and, on the demographic path, h1, h2, film [M][2F], zc. */
int ecg_tail_fwd(const float *g, const float *xd, const float *WpT, const float *bp,
                 const float *W0, const float *b0, const float *W2, const float *b2,
                 const float *WfT, const float *bf, const float *Wh, const float *bh,
                 float *z, float *h1, float *h2, float *film, float *zc, float *logits,
                 int M, int F0, int F, int D, int H1, int H, int C, ecg_stream_t stream);

/* Per-sample backward chain of the tail.  Weights in state_dict layout (Wp [F][F0], W0 [H1][D],
 * W2 [H][H1], Wf [2F][H], Wh [C][F]).  dz_extra (nullable) is added to d z (gradient arriving
 * at z from outside, e.g. return_features).  Outputs: dz [M][F], dg [M][F0]; with demo != 0
 * also dzc [M][F], dfilm [M][2F], dh2m [M][H], dh1m [M][H1] (ReLU-masked) and dxd [M][D] (nullable). */
int ecg_tail_bwd_chain(const float *dlogits, const float *dz_extra, const float *z,
                       const float *h1, const float *h2, const float *film,
                       const float *Wp, const float *W0, const float *W2, const float *Wf,
                       const float *Wh, float *dzc, float *dz, float *dfilm, float *dh2m,
                       float *dh1m, float *dg, float *dxd, int M, int F0, int F, int D,
                       int H1, int H, int C, int demo, ecg_stream_t stream);

/* count (<= 8) Linear weight/bias gradients in one launch:
 * dW[q][o][i] = sum_m G[q][m][o] * X[q][m][i];  db[q][o] = sum_m G[q][m][o] (db[q] nullable).
 * G, X, dW, db, Out, In are HOST arrays of length count. */
int ecg_linear_wgrad_grouped(const float *const *G, const float *const *X, float *const *dW,
                             float *const *db, const int *Out, const int *In, int count,
                             int M, ecg_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Optimizer — torch.optim.AdamW defaults over one flat fp32 buffer
 * (scripts/03_train_ecg_baseline.py:130-133).  grad_scale multiplies g first (1/world).
 * ---------------------------------------------------------------------------------- */
int ecg_adamw_step(float *p, const float *g, float *m, float *v, size_t n, int step,
                   float lr, float beta1, float beta2, float eps, float weight_decay,
                   float grad_scale, ecg_stream_t stream);

/* Same update with the step counter on the device (step_dev[0] = number of steps taken so far,
 * incremented by the call): nothing host-side changes between steps, so the whole train step can
 * be captured once in a hipGraph and replayed (ecg_hip.graph.GraphedTrainStep). */
int ecg_adamw_step_graph(float *p, const float *g, float *m, float *v, size_t n, int *step_dev,
                         float lr, float beta1, float beta2, float eps, float weight_decay,
                         float grad_scale, ecg_stream_t stream);

/* ---- input pipeline: the step before the model -----------------------
 * Reference: `_load_ecg` = wfdb.rdsamp + float32 cast + transpose (src/datasets/ptbxl.py:14-41) and
 * `_normalize` (ptbxl.py:122-127), repeated in ptbxl_ecg_multimodal.py and ptbxl_af.py.
 * Both entry points reproduce the reference's float32 arithmetic bit for bit (its per-lead sums run
 * left to right because it normalises a transposed view).
 *
 * ecg_wfdb16_physical: WFDB format-16 digital samples d [B][T][leads] (the .dat layout: time-major,
 * leads interleaved) -> physical units out [B][leads][T] = float32((d - baseline) / gain), gain and
 * baseline per (window, lead) from the .hea signal lines; sample -32768 (format-16 "invalid") -> NaN.
 * leads <= 16. */
int ecg_wfdb16_physical(const int16_t *d, const double *gain, const int *baseline, float *out,
                        int B, int T, int leads, ecg_stream_t stream);
/* Both steps in one launch where a whole window fits in 64 KB of LDS (12 leads: T up to about 1340; longer
 * windows run the two entry points around this one back to back): d [B][T][leads] -> z-scored
 * out [B][leads][T]; stats [B*leads][2] receives (mean, std + 1e-6) of the physical signal.  HBM
 * traffic is the algorithmic 2 B in + 4 B out per sample. */
int ecg_wfdb16_zscore(const int16_t *d, const double *gain, const int *baseline, float *out,
                      float *stats, int B, int T, int leads, ecg_stream_t stream);
/* Per-lead z-score, (x-mean)/(std+1e-6) with population std.  x [rows][T] -> out [rows][T] (in place
 * allowed); stats [rows][2] receives (mean, std + 1e-6) per row. */
int ecg_zscore_rows(const float *x, float *out, float *stats, int rows, int T, ecg_stream_t stream);
/* HOST side of the same step — what torch's DataLoader collate does for the reference (one `Dataset.__getitem__` per record,
 * src/datasets/ptbxl.py:25,122-127, stacked into a batch): rows `rows[0..n)` of a row-major host table (row_bytes each, e.g.
 * the int16 records of a memory-mapped pack) copied to consecutive rows of `dst` (the pinned staging slot).  Plain memcpy
 * per row, no thread, no allocation, no device call: the caller runs one call per gather thread on disjoint shares
 * (ecg_hip/pack.py; ctypes drops the GIL for the duration, which a per-record numpy copy does not). */
int ecg_host_gather_rows(const void *src, size_t row_bytes, const long long *rows, int n, void *dst);

#ifdef __cplusplus
}
#endif
#endif /* ECG_HIP_H */
