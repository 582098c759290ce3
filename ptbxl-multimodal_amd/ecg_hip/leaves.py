"""The unfused leaf Functions: one autograd node per reference layer (Conv1d, BatchNorm1d, ReLU, MaxPool1d(2),
AdaptiveAvgPool1d(1), Linear, FiLM).  They run when a caller has hooked an inner module of a ConvBlock or of the tail
(Grad-CAM: reference scripts/00_demo_inference.py:36-37, src/explain/grad_cam_1d.py:36), so that hooks see the same tensors
the reference's modules would hand them; the fused train / inference path is ecg_hip/functional.py.  fp32 only."""
import torch

from . import _lib as L
from .functional import (_bn_momentum, _call, _contig, _empty, _f32, _query, _st, bn_batch_stats, bn_eval_stats,
                         conv1d_backward_raw, conv1d_forward_raw, conv1d_pack)


class Conv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, pad):
        x, w = _contig(x), _contig(w)
        Co, _, K = w.shape
        w_fwd, w_bwd = conv1d_pack(w, need_bwd=ctx.needs_input_grad[0])
        y, _, _ = conv1d_forward_raw(x, w_fwd, b, Co, K, pad, want_stats=False)
        ctx.save_for_backward(x, w)
        ctx.w_bwd, ctx.pad, ctx.has_bias = w_bwd, pad, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = conv1d_backward_raw(x, _contig(dy), w.shape, ctx.w_bwd, ctx.pad,
                                         ctx.needs_input_grad[0], need_db=ctx.has_bias)
        return dx, dw, db, None


class BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, nbt, training, momentum, eps):
        y = _contig(y)
        N, C, Lo = y.shape
        use_batch = training or running_mean is None
        if use_batch:
            rm, rv, cnt = (running_mean, running_var, nbt) if training else (None, None, None)
            mean, invstd = bn_batch_stats(y, None, 0, rm, rv, cnt, _bn_momentum(momentum, nbt), eps)
        else:
            mean, invstd = bn_eval_stats(running_mean, running_var, eps)
        out = torch.empty_like(y)
        _call("ecg_bn_apply_fwd", _f32(y), _f32(gamma), _f32(beta), _f32(mean), _f32(invstd),
              _f32(out), N, C, Lo, _st())
        ctx.save_for_backward(y, gamma, mean, invstd)
        ctx.batch_stats = use_batch
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, mean, invstd = ctx.saved_tensors
        N, C, Lo = y.shape
        dy = torch.empty_like(y)
        dgamma, dbeta = _empty(y, C), _empty(y, C)
        ws = _empty(y, _query("ecg_bn_bwd_ws_floats", N, C, Lo))
        _call("ecg_bn_bwd", _f32(y), _f32(_contig(dout)), _f32(gamma), _f32(mean), _f32(invstd),
              _f32(dy), _f32(dgamma), _f32(dbeta), _f32(ws), N, C, Lo,
              1 if ctx.batch_stats else 0, _st())
        return dy, dgamma, dbeta, None, None, None, None, None, None


class ReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _contig(x)
        out = torch.empty_like(x)
        _call("ecg_relu_fwd", _f32(x), _f32(out), x.numel(), _st())
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dx = torch.empty_like(out)
        _call("ecg_relu_bwd", _f32(out), _f32(_contig(dout)), _f32(dx), out.numel(), _st())
        return dx


class MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _contig(x)
        Lin = x.shape[-1]
        rows = x.numel() // Lin
        p = _empty(x, *x.shape[:-1], Lin // 2)
        _call("ecg_maxpool2_fwd", _f32(x), _f32(p), rows, Lin, _st())
        ctx.save_for_backward(x)
        return p

    @staticmethod
    def backward(ctx, dp):
        (x,) = ctx.saved_tensors
        Lin = x.shape[-1]
        dx = torch.empty_like(x)
        _call("ecg_maxpool2_bwd", _f32(x), _f32(_contig(dp)), _f32(dx), x.numel() // Lin, Lin, _st())
        return dx


class GapFn(torch.autograd.Function):
    """AdaptiveAvgPool1d(1): [N,C,L] -> [N,C,1] (reference src/models/ecg_cnn.py:46)."""

    @staticmethod
    def forward(ctx, p):
        p = _contig(p)
        N, C, Lp = p.shape
        g = _empty(p, N, C, 1)
        _call("ecg_gap_fwd", _f32(p), _f32(g), N * C, Lp, _st())
        ctx.shape = (N, C, Lp)
        return g

    @staticmethod
    def backward(ctx, dg):
        N, C, Lp = ctx.shape
        dp = _empty(dg, N, C, Lp)
        _call("ecg_gap_bwd", _f32(_contig(dg)), _f32(dp), N * C, Lp, _st())
        return dp


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b); act = ReLU when relu (reference src/models/ecg_multimodal.py:52-55)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x, w = _contig(x), _contig(w)
        if x.dim() != 2:
            raise L.EcgHipError(f"linear: expected a [M, In] input, got {tuple(x.shape)}")
        M, In = x.shape
        Out = w.shape[0]
        y = _empty(x, M, Out)
        _call("ecg_linear_fwd", _f32(x), _f32(w), _f32(b), _f32(y), M, In, Out, int(relu), _st())
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.relu, ctx.has_bias = bool(relu), b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        M, In = x.shape
        Out = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        db = _empty(x, Out) if ctx.has_bias and ctx.needs_input_grad[2] else None
        _call("ecg_linear_bwd", _f32(x), _f32(w), _f32(y), _f32(_contig(dy)), _f32(dx), _f32(dw),
              _f32(db), None, M, In, Out, int(ctx.relu), _st())
        return dx, dw, db, None


class FilmFn(torch.autograd.Function):
    """zc = (1 + tanh(film[:, :F])) * z + film[:, F:] (reference src/models/ecg_multimodal.py:92-96)."""

    @staticmethod
    def forward(ctx, z, film):
        z, film = _contig(z), _contig(film)
        M, F = z.shape
        if film.shape != (M, 2 * F):
            raise L.EcgHipError(f"film: expected film of shape {(M, 2 * F)}, got {tuple(film.shape)}")
        zc = torch.empty_like(z)
        _call("ecg_film_fwd", _f32(z), _f32(film), _f32(zc), M, F, _st())
        ctx.save_for_backward(z, film)
        return zc

    @staticmethod
    def backward(ctx, dzc):
        z, film = ctx.saved_tensors
        M, F = z.shape
        dz, dfilm = torch.empty_like(z), torch.empty_like(film)
        _call("ecg_film_bwd", _f32(z), _f32(film), _f32(_contig(dzc)), _f32(dz), _f32(dfilm), M, F, _st())
        return dz, dfilm
