"""torch.autograd.Function wrappers over the C ABI (include/ecg_hip.h).

Each Function's forward/backward is a straight sequence of libecg_hip.so launches on
torch's current stream; torch only owns the memory.  Hand-written backward formulas are
the ones restated (and pinned against the reference) in oracle/ecg_oracle.c.
"""
import torch

from . import _lib as L

_call, _query, _f32, _st = L.call, L.query, L.f32, L.stream


def _empty(ref, *shape):
    return torch.empty(shape, dtype=torch.float32, device=ref.device)


# --------------------------------------------------------------------------------------
# Gradient sinks.  FlatAdamW keeps ONE flat gradient buffer; when it registers a view of that buffer for
# a parameter, the backward kernels write that parameter's gradient straight into it and autograd's
# AccumulateGrad adopts the view as `.grad` (no copy) — the optimizer's per-step gather (`torch.cat` of
# ~28 tensors) and the all-reduce staging disappear.  A sink is only handed out while the parameter's
# `.grad` is None (zero_grad(set_to_none=True), the default): if a gradient is already there, autograd will
# ADD the new one to it, and the kernel must not have overwritten the accumulated value.
# --------------------------------------------------------------------------------------
import weakref as _weakref

_grad_sinks = {}          # parameter data_ptr -> (weakref(parameter), flat view)


def register_grad_sinks(params, views):
    for p, v in zip(params, views):
        _grad_sinks[p.data_ptr()] = (_weakref.ref(p), v)


def unregister_grad_sinks(keys):
    for k in keys:
        _grad_sinks.pop(k, None)


_sinks_handed = set()     # sinks already given to a kernel in the backward pass that is running now


def _forget_handed_sinks():
    _sinks_handed.clear()


def _grad_out(key, ref, *shape):
    """Destination of a parameter gradient: a FRESH view of the registered sink (AccumulateGrad only adopts
    a tensor nobody else holds) or a new tensor.  A sink is handed out AT MOST ONCE per backward pass: a
    parameter used twice in one graph (model called twice before backward(), shared weights) gets a private
    tensor for its second gradient, which autograd then adds to the first — two kernels writing the same
    sink would leave 2*dw2 instead of dw1 + dw2.  The set is cleared by an engine callback when the pass ends — and,
    because the engine runs no callbacks for a backward pass that RAISED, again whenever a new step begins
    (backward_from_loss, FlatAdamW.zero_grad / step): one failed backward must not disable the sinks for good."""
    ent = _grad_sinks.get(key) if key is not None else None
    if ent is not None and key not in _sinks_handed:
        p, v = ent[0](), ent[1]
        if p is not None and p.grad is None and v.device == ref.device and v.numel() == _numel(shape):
            try:
                if not _sinks_handed:
                    torch.autograd.Variable._execution_engine.queue_callback(_forget_handed_sinks)
                _sinks_handed.add(key)
            except RuntimeError:         # not inside a backward pass (direct call of a backward formula): no tracking
                return torch.empty(shape, dtype=torch.float32, device=ref.device)
            return v.view(shape)
    return torch.empty(shape, dtype=torch.float32, device=ref.device)


# --------------------------------------------------------------------------------------
# The one-launch BatchNorm backward (csrc/bn_relu_pool.hip, bn_bwd_resident_kernel) and collectives.
# Its workgroups exchange partial sums through a bounded wait, which is only instant while every workgroup of the grid
# is resident.  A communication kernel that runs DURING backward and waits for a late peer keeps its CUs; this kernel's
# workgroups would then run out their wait and take the slow self-service path on every call.  The decision is taken
# PER CALL, from what is known about the parameters of the block whose backward is running — no process-wide switch:
#   * the owner of the parameters (FlatAdamW, FlatGradDDP) declares "busy" (its hooks issue all-reduces under backward)
#     or "quiet" (its exchange starts after the last backward kernel);
#   * parameters nobody has declared are quiet in a single-rank process and BUSY as soon as torch.distributed runs more
#     than one rank — so a model wrapped in stock torch.nn.parallel.DistributedDataParallel (bucketed all-reduces under
#     backward, invisible from here) gets the two-pass form without having to ask for it.
# ECG_BN_BWD_RESIDENT=0 keeps the two-pass form everywhere; ECG_HIP_REHEARSE_ON_ONE_GPU=1 (several ranks sharing ONE
# device: workgroups of different processes would wait for each other's CUs) does the same.  Read here, on the host side
# of the ABI: the library reads no environment.
# --------------------------------------------------------------------------------------
import os as _os

_bn_one_launch = (_os.environ.get("ECG_BN_BWD_RESIDENT", "1") != "0"
                  and _os.environ.get("ECG_HIP_REHEARSE_ON_ONE_GPU") != "1")
_bn_spin_polls = int(_os.environ.get("ECG_BN_BWD_RESIDENT_SPIN", "-1"))     # tests: 0 = nobody waits (self-service path)
_backward_collectives = {}      # id(parameter) -> (weakref(parameter), busy, owner): keyed by the OBJECT, not its address —
                                # FlatAdamW / adopt_stock_adamw re-home parameter storage after a wrapper has spoken
_collectives_by_ptr = None      # {data_ptr now: busy}, rebuilt lazily after a declaration or a re-homing
_bn_counters = {}               # (device index, raw stream) -> zeroed int32 tensor: the kernel's exchange words


def set_bn_backward_one_launch(on, spin_polls=None):
    """Process default for the one-launch BatchNorm backward (True unless the environment says otherwise); returns the
    previous setting.  `spin_polls` (tests) bounds the inter-workgroup wait: 0 forces the self-service path."""
    global _bn_one_launch, _bn_spin_polls
    prev, _bn_one_launch = _bn_one_launch, bool(on)
    if spin_polls is not None:
        _bn_spin_polls = int(spin_polls)
    return prev


def declare_backward_collectives(params, busy, owner=None):
    """The owner of `params` says whether collectives can run on the device while their backward kernels do (True: the
    BatchNorm backward of their blocks keeps the two-pass form), that they cannot (False), or withdraws its statement
    (None; with an `owner` token only a statement made under the same token is withdrawn, so an optimizer going away
    does not take a live wrapper's statement with it).  `params` are the parameter tensors themselves (or their id()s,
    for a withdrawal from a finalizer).  Cheap; call it whenever the exchange mode changes."""
    global _collectives_by_ptr
    for p in params:
        k = p if isinstance(p, int) else id(p)
        if busy is None:
            ent = _backward_collectives.get(k)
            if ent is not None and (owner is None or ent[2] == owner):
                del _backward_collectives[k]
        elif not isinstance(p, int):
            _backward_collectives[k] = (_weakref.ref(p), bool(busy), owner)
    _collectives_by_ptr = None


def parameters_rehomed():
    """Parameter storage moved (flatten_tensors_): the address -> statement table is rebuilt at the next backward."""
    global _collectives_by_ptr
    _collectives_by_ptr = None


def _multi_rank():
    d = torch.distributed
    return d.is_available() and d.is_initialized() and d.get_world_size() > 1


def bn_backward_one_launch_allowed(key):
    """The per-call decision described above for the block whose conv weight has data_ptr `key`."""
    global _collectives_by_ptr
    if not _bn_one_launch:
        return False
    if _collectives_by_ptr is None:
        tab = {}
        for k, (ref, busy, _owner) in list(_backward_collectives.items()):
            p = ref()
            if p is None:
                del _backward_collectives[k]          # the parameter is gone (and its id may be reused)
            else:
                tab[p.data_ptr()] = busy
        _collectives_by_ptr = tab
    busy = _collectives_by_ptr.get(key)
    return (not _multi_rank()) if busy is None else (not busy)


def _bn_bwd_counters(ref, n_uints):
    """Caller-owned exchange words of the one-launch BatchNorm backward: zero before the first launch, left zero by every
    launch.  One buffer per (device, stream) — launches on one stream are ordered, launches on different streams get
    different buffers.  Inside a stream capture a FRESH zeroed buffer is allocated (a memset node + memory of the
    graph's private pool): a captured step never shares its words with eager launches or with another graph."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(n_uints, dtype=torch.int32, device=ref.device)
    key = (ref.device.index, _st())
    t = _bn_counters.get(key)
    if t is None or t.numel() < n_uints:
        t = torch.zeros(max(n_uints, 4352), dtype=torch.int32, device=ref.device)
        _bn_counters[key] = t
    return t


def _numel(shape):
    n = 1
    for d in shape:
        n *= int(d)
    return n


def _key(t):
    return None if t is None else t.data_ptr()


def _contig(t):
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------------------
# raw launches (no autograd) — shared by the Functions below
# --------------------------------------------------------------------------------------
def conv1d_pack(w, need_bwd=True):
    Co, Ci, K = w.shape
    w_fwd = torch.empty_like(w).view(K, Ci, Co)
    w_bwd = torch.empty_like(w).view(K, Co, Ci) if need_bwd else None
    _call("ecg_conv1d_pack_weights", _f32(w), _f32(w_fwd), _f32(w_bwd), Co, Ci, K, _st())
    return w_fwd, w_bwd


# --------------------------------------------------------------------------------------
# Conv precision: "fp32" (default, the parity path: exact-fp32 MFMA) or "bf16" (opt-in mixed precision, BASELINE.json
# config 5).  The mixed-precision step has ONE form (round 5): a TRAINING ConvBlock (batch statistics, gradients wanted,
# K = 15, pad = 7, channel counts the bf16 MFMA kernels tile) runs its three convs on bf16 operands with fp32 accumulation
# and keeps every tensor between its kernels as bf16 rows [N][C][ld] — y, the pooled activation handed to the next
# block, dY and the input gradient handed back — while parameters, statistics, gradients of parameters, the network
# input, the tail and the optimizer stay fp32.  A block that does not fit that form (frozen or eval-mode BatchNorm, no
# gradient wanted, other kernel sizes or channel counts) runs the exact fp32 kernels instead, and tensors crossing between the
# two forms are fp32: inference is always the fp32 one-launch path.
# ECG_HIP_CONV_PRECISION sets the process default.
# --------------------------------------------------------------------------------------
_conv_precision = _os.environ.get("ECG_HIP_CONV_PRECISION", "fp32")


def set_conv_precision(mode):
    global _conv_precision
    if mode not in ("fp32", "bf16"):
        raise ValueError("conv precision must be 'fp32' or 'bf16'")
    _conv_precision = mode


def get_conv_precision():
    return _conv_precision


class conv_precision:
    """Context manager: `with conv_precision("bf16"): model(x)`."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_conv_precision()
        set_conv_precision(self.mode)

    def __exit__(self, *exc):
        set_conv_precision(self.prev)
        return False


def conv1d_pack_bf16(w, need_bwd=True):
    Co, Ci, K = w.shape
    nf = _query("ecg_conv1d_bf16_packed_elems", Ci, Co, K)
    wb_fwd = torch.empty(nf, dtype=torch.bfloat16, device=w.device)
    wb_bwd = None
    if need_bwd:
        wb_bwd = torch.empty(_query("ecg_conv1d_bf16_packed_elems", Co, Ci, K), dtype=torch.bfloat16, device=w.device)
    _call("ecg_conv1d_pack_weights_bf16", _f32(w), L.ptr(wb_fwd), L.ptr(wb_bwd), Co, Ci, K, _st())
    return wb_fwd, wb_bwd


class WeightPacker:
    """All per-forward weight repacking of a model in one launch: conv weights -> (w_fwd, w_bwd)
    and the Linear weights the fused tail reads transposed.  Destination buffers and the
    host-side pointer tables are cached while the parameter storage stays put."""

    def __init__(self):
        self._key = None

    # the cache holds raw device pointers: never copy or pickle it along with the model
    def __deepcopy__(self, memo):
        return WeightPacker()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self._key = None

    def pack(self, convs, linears, need_bwd):
        """-> (conv_packs, linear_T); conv_packs[i] = (w_fwd, w_bwd) fp32 operands, or in bf16 mode, for a conv the
        bf16 kernels take, (w_fwd, w_bwd, wb_fwd, wb_bwd) with the fp32 pair None — the bf16 operands come out of the
        same launch (one launch per step instead of one per layer plus the fp32 one)."""
        srcs = [c.weight for c in convs] + [l.weight for l in linears]
        mixed = _conv_precision == "bf16" and bool(need_bwd)       # (without gradients every block runs the fp32 kernels)
        key = (need_bwd, mixed) + tuple((w.data_ptr(), tuple(w.shape)) for w in srcs)
        if key != self._key:
            self._key = key
            self.conv_packs, self.linear_T, fw, bw, hf, hb, co, ci, kk = [], [], [], [], [], [], [], [], []
            for i, c in enumerate(convs):
                w = _contig(c.weight)
                Co, Ci, K = w.shape
                want_bwd = need_bwd and i > 0                # block 0 has no input-grad
                # (bf16 operands for a conv whose geometry fits the mixed-precision form; whether the block really takes it
                # is decided per call — a block that does not repacks its fp32 operands itself, ConvBlockFn._weights)
                if mixed and _bf16_block_ok(Ci, Co, K, c.padding[0], 16, want_bwd):
                    wf = wb = None
                    hwf = torch.empty(_query("ecg_conv1d_bf16_packed_elems", Ci, Co, K), dtype=torch.bfloat16, device=w.device)
                    hwb = (torch.empty(_query("ecg_conv1d_bf16_packed_elems", Co, Ci, K), dtype=torch.bfloat16,
                                       device=w.device) if want_bwd else None)
                    self.conv_packs.append((None, None, hwf, hwb))
                else:
                    hwf = hwb = None
                    wf = torch.empty_like(w).view(K, Ci, Co)
                    wb = torch.empty_like(w).view(K, Co, Ci) if want_bwd else None
                    self.conv_packs.append((wf, wb))
                fw.append(wf); bw.append(wb); hf.append(hwf); hb.append(hwb); co.append(Co); ci.append(Ci); kk.append(K)
            for l in linears:
                w = _contig(l.weight)
                wt = torch.empty(w.shape[1], w.shape[0], dtype=w.dtype, device=w.device)
                self.linear_T.append(wt)
                fw.append(wt); bw.append(None); hf.append(None); hb.append(None)
                co.append(w.shape[0]); ci.append(w.shape[1]); kk.append(1)
            self._mixed = any(t is not None for t in hf)
            tabs = [L.ptr_table([_contig(w) for w in srcs]), L.ptr_table(fw), L.ptr_table(bw)]
            if self._mixed:
                tabs += [L.ptr_table(hf, any_dtype=True), L.ptr_table(hb, any_dtype=True)]
            self._tables = tuple(tabs) + (L.int_table(co), L.int_table(ci), L.int_table(kk), len(srcs))
        _call("ecg_pack_weights_grouped_mixed" if self._mixed else "ecg_pack_weights_grouped", *self._tables, _st())
        return self.conv_packs, self.linear_T


def conv1d_forward_raw(x, w_fwd, bias, Co, K, pad, want_stats):
    N, Ci, Lin = x.shape
    Lo = Lin + 2 * pad - K + 1
    if Lo <= 0:
        raise L.EcgHipError(f"conv1d: empty output for L={Lin}, K={K}, pad={pad}")
    y = _empty(x, N, Co, Lo)
    partials, P = None, 0
    if want_stats:
        P = _query("ecg_conv1d_fwd_stat_partials", N, Ci, Co, Lin, K, pad)
        partials = _empty(x, Co * P * 2)
    _call("ecg_conv1d_fwd", _f32(x), _f32(w_fwd), _f32(bias), _f32(y), _f32(partials),
          N, Ci, Co, Lin, K, pad, _st())
    return y, partials, P


def conv1d_backward_raw(x, dy, w_shape, w_bwd, pad, need_dx, need_db=True, ldy=None, sink_keys=(None, None)):
    """fp32 weight / bias / input gradient.  dy is [N, C_out, ldy] with ldy >= Lo (row-padded, zero pad) when ldy is given.
    (Weight gradient on a side stream under the next block's BatchNorm passes: measured 1.94 against 1.92 ms per step in round 1,
    1.604 against 1.580 in round 3 — two MFMA-bound kernels sharing the CUs lose what the hidden passes save; removed.)"""
    N, Ci, Lin = x.shape
    Co, _, K = w_shape
    if ldy is None:
        ldy = Lin + 2 * pad - K + 1
    dw = _grad_out(sink_keys[0], x, Co, Ci, K)
    db = _grad_out(sink_keys[1], x, Co) if need_db else None
    ws = _empty(x, max(1, _query("ecg_conv1d_bwd_weight_ws_floats", N, Ci, Co, Lin, K, pad)))
    _call("ecg_conv1d_bwd_weight_bias_ld", _f32(dy), ldy, _f32(x), _f32(dw), _f32(db), _f32(ws), N, Ci, Co, Lin, K, pad,
          _st())
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        _call("ecg_conv1d_bwd_data_ld", _f32(dy), ldy, _f32(w_bwd), _f32(dx), N, Ci, Co, Lin, K, pad, _st())
    return dx, dw, db


def bn_batch_stats(y, partials, P, running_mean, running_var, nbt, momentum, eps):
    """mean/invstd of y [N,C,L] (+ running-stat / counter update) from conv-epilogue partials,
    or from a standalone pass when partials is None."""
    N, C, Lo = y.shape
    if partials is None:
        P = _query("ecg_bn_stat_partials_count", N, C, Lo)
        partials = _empty(y, C * P * 2)
        _call("ecg_bn_stat_partials", _f32(y), _f32(partials), N, C, Lo, _st())
    mean, invstd = _empty(y, C), _empty(y, C)
    if nbt is not None and nbt.dtype != torch.int64:
        raise L.EcgHipError("num_batches_tracked must be int64")
    _call("ecg_bn_finalize", _f32(partials), P, N * Lo, _f32(mean), _f32(invstd),
          _f32(running_mean), _f32(running_var), L.ptr(nbt), C, float(momentum), float(eps), _st())
    return mean, invstd


def bn_eval_stats(running_mean, running_var, eps):
    invstd = torch.empty_like(running_var)
    _call("ecg_bn_invstd", _f32(running_var), _f32(invstd), running_var.numel(), float(eps), _st())
    return running_mean, invstd


def _bn_momentum(momentum, nbt):
    if momentum is None:      # cumulative moving average: factor = 1/(count+1) (host read)
        return 1.0 / (int(nbt.item()) + 1) if nbt is not None else 0.0
    return float(momentum)


# --------------------------------------------------------------------------------------
# Fused ConvBlock: Conv1d -> BatchNorm1d -> ReLU -> MaxPool1d(2)
# --------------------------------------------------------------------------------------
def _bf16_block_ok(Ci, Co, K, pad, Lin, need_dx):
    """Geometry test of the mixed-precision block form: the bf16 forward (bit 0 of ecg_conv1d_bf16_supported), the time-on-K
    weight gradient (bit 2: K == 15, pad == 7, C_out % 32 == 0 — which also gives the odd pad / odd K-1-pad the position-pair
    staging of bf16 rows needs), the bf16 input gradient when one is wanted (bit 1), and a pooled row that is not empty."""
    if K > 15:
        return False
    sup = _query("ecg_conv1d_bf16_supported", Ci, Co, K, pad)
    return bool((sup & 1) and (sup & 4) and (not need_dx or (sup & 2)) and Lin + 2 * pad - K + 1 >= 2)


_EVAL, _FP32, _BF16 = "eval", "fp32", "bf16"


class ConvBlockFn(torch.autograd.Function):
    """reference src/models/ecg_cnn.py:12-17 as 3 launches forward (conv + BN-statistics epilogue, statistics combine +
    BN-apply + ReLU + pool) and 3-4 backward.  ONE dispatch at the top picks the form of the block, recorded in ctx.mode and
    reused by backward:
      eval   inference (running statistics, no gradient wanted): conv + folded BN + ReLU + pool [+ GAP] in one launch
      bf16   the mixed-precision training form (set_conv_precision("bf16"), see _bf16_block_ok): bf16 rows between the kernels
      fp32   everything else — the parity path"""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, running_mean, running_var, nbt, training, momentum,
                eps, pad, gap=False, packed=None, grad_enabled=True, x_len=None, next_bf16=False):
        """Returns p: fp32 [N][C_out][Lo/2] ([N][C_out] with `gap`), or — bf16 form with `next_bf16` (the caller's promise that
        the block consuming p takes the bf16 form too) — bf16 [N][C_out][ldp] with rows zero-filled past Lo/2.  `x_len` is
        the true row length when x itself is such a bf16 activation [N][C_in][ldx]."""
        x, w = _contig(x), _contig(w)
        Co, Ci, K = w.shape
        x_h = x.dtype == torch.bfloat16
        use_batch = training or running_mean is None
        # grad mode is always off INSIDE forward and needs_input_grad ignores torch.no_grad(): the caller
        # samples torch.is_grad_enabled() and hands it in, so that inference takes the one-launch kernel
        need_grad = grad_enabled and any(ctx.needs_input_grad)
        need_dx = bool(need_grad and ctx.needs_input_grad[0])
        Lin = x_len if x_h else x.shape[2]
        Lo = Lin + 2 * pad - K + 1
        if Lo <= 0:
            raise L.EcgHipError(f"conv1d: empty output for L={Lin}, K={K}, pad={pad}")
        mode = _FP32
        if _conv_precision == "bf16" and use_batch and need_grad and _bf16_block_ok(Ci, Co, K, pad, Lin, need_dx):
            mode = _BF16
        elif not use_batch and not need_grad:
            mode = _EVAL
        if x_h and mode != _BF16:
            # the producer wrote p as bf16 because this block looked able to read it: the precision was switched, or a
            # BatchNorm frozen, between the two calls
            raise L.EcgHipError("ConvBlock: input arrived as a bf16 activation but this block does not take the mixed-"
                                "precision form (conv precision / BatchNorm mode changed between blocks?)")
        ctx.mode, ctx.pad, ctx.gap, ctx.batch_stats, ctx.Lin, ctx.Lo = mode, pad, gap, use_batch, Lin, Lo
        ctx.sink_keys = (_key(w), _key(b), _key(gamma), _key(beta))
        if mode == _EVAL:
            p = ConvBlockFn._fwd_eval(x, w, b, gamma, beta, running_mean, running_var, eps, pad, gap, packed)
            if p is not None:
                return p
            ctx.mode = mode = _FP32            # (a shape the one-launch kernel does not tile: the three-launch sequence)
        if mode == _BF16:
            return ConvBlockFn._fwd_bf16(ctx, x, w, b, gamma, beta, running_mean, running_var, nbt, training, momentum, eps,
                                         packed, need_dx, next_bf16)
        return ConvBlockFn._fwd_fp32(ctx, x, w, b, gamma, beta, running_mean, running_var, nbt, training, momentum, eps,
                                     packed, need_grad, need_dx)

    # ---- weights: from the model's grouped repack when it holds what this form needs, else packed here -------------------
    @staticmethod
    def _weights(w, packed, bf16, need_bwd):
        if bf16:
            if packed is not None and len(packed) == 4 and packed[2] is not None and (packed[3] is not None or not need_bwd):
                return packed[2], packed[3]
            return conv1d_pack_bf16(w, need_bwd=need_bwd)
        if packed is not None and packed[0] is not None and (packed[1] is not None or not need_bwd):
            return packed[0], packed[1]
        return conv1d_pack(w, need_bwd=need_bwd)

    # ---- eval: one launch, y never written --------------------------------------------------------------------------
    @staticmethod
    def _fwd_eval(x, w, b, gamma, beta, running_mean, running_var, eps, pad, gap, packed):
        N, Ci, Lin = x.shape
        Co, _, K = w.shape
        Lo = Lin + 2 * pad - K + 1
        if not gap and _query("ecg_conv1d_bn_relu_pool_eval_supported", Ci, Co, K, pad):
            w_fwd, _ = ConvBlockFn._weights(w, packed, False, False)
            p = _empty(x, N, Co, Lo // 2)
            _call("ecg_conv1d_bn_relu_pool_eval_fwd", _f32(x), _f32(w_fwd), _f32(b), _f32(gamma), _f32(beta),
                  _f32(running_mean), _f32(running_var), float(eps), _f32(p), N, Ci, Co, Lin, K, pad, _st())
            return p
        if gap and _query("ecg_conv1d_bn_relu_pool_gap_eval_supported", Ci, Co, Lin, K, pad):
            w_fwd, _ = ConvBlockFn._weights(w, packed, False, False)
            g = _empty(x, N, Co)
            _call("ecg_conv1d_bn_relu_pool_gap_eval_fwd", _f32(x), _f32(w_fwd), _f32(b), _f32(gamma), _f32(beta),
                  _f32(running_mean), _f32(running_var), float(eps), _f32(g), N, Ci, Co, Lin, K, pad, _st())
            return g
        return None

    # ---- fp32: the parity path ----------------------------------------------------------------------------------------
    @staticmethod
    def _fwd_fp32(ctx, x, w, b, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, packed, need_grad,
                  need_dx):
        N, Ci, Lin = x.shape
        Co, _, K = w.shape
        pad, gap, use_batch, Lo = ctx.pad, ctx.gap, ctx.batch_stats, ctx.Lo
        w_fwd, w_bwd = ConvBlockFn._weights(w, packed, False, need_dx)
        y, partials, P = conv1d_forward_raw(x, w_fwd, b, Co, K, pad, want_stats=use_batch)
        p = _empty(x, N, Co) if gap else _empty(x, N, Co, Lo // 2)
        if use_batch:
            # statistics combine + BN-apply + ReLU + pool in ONE launch (ecg_bn_finalize folded into the streaming pass)
            rm, rv, cnt = (running_mean, running_var, nbt) if training else (None, None, None)
            if cnt is not None and cnt.dtype != torch.int64:
                raise L.EcgHipError("num_batches_tracked must be int64")
            mean, invstd = _empty(x, Co), _empty(x, Co)
            _call("ecg_bn_stats_relu_pool_fwd", _f32(partials), P, N * Lo, _f32(rm), _f32(rv), L.ptr(cnt),
                  _bn_momentum(momentum, nbt), float(eps), _f32(y), _f32(gamma), _f32(beta), _f32(mean), _f32(invstd),
                  _f32(p), N, Co, Lo, 1 if gap else 0, _st())
        else:
            mean, invstd = bn_eval_stats(running_mean, running_var, eps)
            # (last block: AdaptiveAvgPool1d(1) folded in, the pooled tensor never exists)
            _call("ecg_bn_relu_pool_gap_fwd" if gap else "ecg_bn_relu_pool_fwd", _f32(y), _f32(gamma), _f32(beta),
                  _f32(mean), _f32(invstd), _f32(p), N, Co, Lo, _st())
        ctx.save_for_backward(x, w, y, gamma, beta, mean, invstd)
        ctx.w_bwd = w_bwd
        return p

    @staticmethod
    def _bwd_fp32(ctx, dp):
        x, w, y, gamma, beta, mean, invstd = ctx.saved_tensors
        N, Co, Lo = y.shape
        if dp is None:          # (only possible with set_materialize_grads(False): p unused downstream)
            dp = torch.zeros(N, Co, *(() if ctx.gap else (Lo // 2,)), device=y.device)
        dp = _contig(dp)
        Ci, Lin, K = x.shape[1], x.shape[2], w.shape[2]
        need_dx = ctx.needs_input_grad[0]
        kw, kb, kg, kbe = ctx.sink_keys
        dgamma, dbeta = _grad_out(kg, y, Co), _grad_out(kbe, y, Co)
        # dY never leaves this function: give it the row stride the conv gradients stream best
        # (rows padded to 64 floats, zero pad -> LDS-DMA in the weight gradient)
        ldy = _query("ecg_conv1d_dy_row_stride", N, Ci, Co, Lin, K, ctx.pad, int(bool(need_dx)))
        dy = _empty(y, N, Co, ldy)
        if bn_backward_one_launch_allowed(kw) and _query("ecg_bn_relu_pool_bwd_one_launch_splits", N, Co, Lo, ldy):
            # operands resident in registers: one launch, one read of (dp, y); the exchange words are ours
            n_u = _query("ecg_bn_relu_pool_bwd_one_launch_counter_uints", N, Co, Lo, ldy)
            cnt = _bn_bwd_counters(y, n_u) if n_u else None
            _call("ecg_bn_relu_pool_bwd_one_launch", _f32(y), _f32(dp), _f32(gamma), _f32(beta), _f32(mean),
                  _f32(invstd), _f32(dy), ldy, _f32(dgamma), _f32(dbeta), L.ptr(cnt), N, Co, Lo,
                  1 if ctx.batch_stats else 0, 1 if ctx.gap else 0, _bn_spin_polls, _st())
        else:
            ws = _empty(y, _query("ecg_bn_relu_pool_bwd_ws_floats", N, Co, Lo))
            _call("ecg_bn_relu_pool_gap_bwd_ld" if ctx.gap else "ecg_bn_relu_pool_bwd_ld", _f32(y), _f32(dp),
                  _f32(gamma), _f32(beta), _f32(mean), _f32(invstd), _f32(dy), ldy, _f32(dgamma),
                  _f32(dbeta), _f32(ws), N, Co, Lo, 1 if ctx.batch_stats else 0, _st())
        dx, dw, db = conv1d_backward_raw(x, dy, w.shape, ctx.w_bwd, ctx.pad, need_dx, ldy=ldy, sink_keys=(kw, kb))
        return dx, dw, db, dgamma, dbeta

    # ---- bf16: the mixed-precision training form (plain bf16 rows end to end) -----------------------------------------
    @staticmethod
    def _fwd_bf16(ctx, x, w, b, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, packed, need_dx,
                  next_bf16):
        N, Ci = x.shape[0], x.shape[1]
        Co, _, K = w.shape
        pad, gap, Lin, Lo = ctx.pad, ctx.gap, ctx.Lin, ctx.Lo
        x_h = x.dtype == torch.bfloat16
        ctx.x_cast = False
        if not x_h and (need_dx or Lin % 8 != 0 or x.data_ptr() % 16 != 0):
            # an fp32 input the kernels cannot take as it is — an input gradient is wanted (it comes back as bf16 rows), the
            # row length is not a multiple of 8, or the tensor is not 16-byte aligned (the network input read in place by
            # block 0 is the aligned, gradient-free case): make the bf16 rows here, with torch ops; backward widens dx again
            xh = torch.zeros(N, Ci, (Lin + 7) & ~7, dtype=torch.bfloat16, device=x.device)
            xh[:, :, :Lin].copy_(x)
            x, x_h, ctx.x_cast = xh, True, True
        ldx = x.shape[2] if x_h else 0
        w_fwd, w_bwd = ConvBlockFn._weights(w, packed, True, need_dx)
        # y as bf16 [N][Co][ldyh] (what torch.autocast would keep): the BatchNorm passes are HBM-bound and y is their
        # largest operand; the statistics are taken over the rounded values
        ldyh = (Lo + 7) & ~7
        y = torch.empty(N, Co, ldyh, dtype=torch.bfloat16, device=x.device)
        P = _query("ecg_conv1d_fwd_bf16_yh_stat_partials", N, Ci, Co, Lin, K, pad, 1 if x_h else 0, ldx, ldyh)
        partials = _empty(x, Co * P * 2)
        _call("ecg_conv1d_fwd_bf16_yh", L.ptr(x), 1 if x_h else 0, ldx, L.ptr(w_fwd), _f32(b), L.ptr(y), ldyh,
              _f32(partials), N, Ci, Co, Lin, K, pad, _st())
        rm, rv, cnt = (running_mean, running_var, nbt) if training else (None, None, None)
        if cnt is not None and cnt.dtype != torch.int64:
            raise L.EcgHipError("num_batches_tracked must be int64")
        mean, invstd = _empty(x, Co), _empty(x, Co)
        ldp = 0
        if gap:
            p = _empty(x, N, Co)
            _call("ecg_bn_stats_relu_pool_gap_fwd_yh", _f32(partials), P, N * Lo, _f32(rm), _f32(rv), L.ptr(cnt),
                  _bn_momentum(momentum, nbt), float(eps), L.ptr(y), ldyh, _f32(gamma), _f32(beta), _f32(mean), _f32(invstd),
                  _f32(p), N, Co, Lo, _st())
        else:
            # p as bf16 rows [N][Co][ldp], zero-filled past Lo/2: the next conv's forward reads half the bytes, and its input
            # gradient comes back as bf16 of the same shape
            ldp = (Lo // 2 + 7) & ~7
            p = torch.empty(N, Co, ldp, dtype=torch.bfloat16, device=x.device)
            _call("ecg_bn_stats_relu_pool_fwd_h", _f32(partials), P, N * Lo, _f32(rm), _f32(rv), L.ptr(cnt),
                  _bn_momentum(momentum, nbt), float(eps), L.ptr(y), ldyh, _f32(gamma), _f32(beta), _f32(mean),
                  _f32(invstd), L.ptr(p), ldp, N, Co, Lo, _st())
            if not next_bf16:
                # the consumer is not a mixed-precision block (a model that mixes geometries, a frozen BatchNorm downstream):
                # hand it fp32 rows; its gradient comes back as fp32 rows (dp_kind 2 of the backward pass)
                p, ldp = p[:, :, :Lo // 2].float(), 0
        ctx.save_for_backward(x, w, y, gamma, beta, mean, invstd)
        ctx.w_bwd, ctx.ldyh, ctx.ldp = w_bwd, ldyh, ldp
        ctx.set_materialize_grads(False)
        return p

    @staticmethod
    def _bwd_bf16(ctx, dp):
        x, w, y, gamma, beta, mean, invstd = ctx.saved_tensors
        N, Co, Lo, Lin = y.shape[0], y.shape[1], ctx.Lo, ctx.Lin
        if dp is None:
            dp = (torch.zeros(N, Co, ctx.ldp, dtype=torch.bfloat16, device=y.device) if ctx.ldp else
                  torch.zeros(N, Co, *(() if ctx.gap else (Lo // 2,)), device=y.device))
        dp = _contig(dp)
        if ctx.ldp and (dp.dtype != torch.bfloat16 or dp.shape[2] != ctx.ldp):
            raise L.EcgHipError("ConvBlock backward: the gradient of a bf16 activation must be bf16 of the same shape")
        Ci, K = x.shape[1], w.shape[2]
        x_h = x.dtype == torch.bfloat16
        need_dx = ctx.needs_input_grad[0]
        kw, kb, kg, kbe = ctx.sink_keys
        dgamma, dbeta = _grad_out(kg, y, Co), _grad_out(kbe, y, Co)
        ws = _empty(y, _query("ecg_bn_relu_pool_bwd_ws_floats", N, Co, Lo))
        # BatchNorm backward (reduction + dx, 16 bytes per lane) writes dY ONCE, as bf16 [N][Co][ldt] with rows zero-filled
        # to a multiple of 128; the time-on-K weight gradient and the input gradient both read it
        ldt = _query("ecg_conv1d_bf16_tk_dy_stride", Lo)
        dyh = torch.empty(N, Co, ldt, dtype=torch.bfloat16, device=y.device)
        # dp: the fp32 gradient of the fused global average pool [N][Co], bf16 rows [N][Co][ldp], or fp32 rows [N][Co][Lo/2]
        dp_kind, dp_ld = (1, 0) if ctx.gap else ((0, ctx.ldp) if ctx.ldp else (2, Lo // 2))
        _call("ecg_bn_relu_pool_bwd_h", L.ptr(y), ctx.ldyh, L.ptr(dp), dp_kind, dp_ld, _f32(gamma), _f32(beta), _f32(mean),
              _f32(invstd), L.ptr(dyh), ldt, _f32(dgamma), _f32(dbeta), _f32(ws), N, Co, Lo, 1, _st())
        dw, db = _grad_out(kw, x, Co, Ci, K), _grad_out(kb, x, Co)
        ws2 = _empty(y, max(1, _query("ecg_conv1d_bwd_weight_bf16_ncl_ws_floats", N, Ci, Co, Lin, K, ctx.pad)))
        _call("ecg_conv1d_bwd_weight_bias_bf16_ncl", L.ptr(dyh), ldt, L.ptr(x), 1 if x_h else 0, x.shape[2], _f32(dw),
              _f32(db), _f32(ws2), N, Ci, Co, Lin, K, ctx.pad, _st())
        dx = None
        if need_dx:         # (x is bf16 rows here: the previous block's activation — dx is its dp — or the cast of an fp32 input)
            dx = torch.empty_like(x)
            _call("ecg_conv1d_bwd_data_bf16hh", L.ptr(dyh), ldt, L.ptr(ctx.w_bwd), L.ptr(dx), x.shape[2], N, Ci, Co,
                  Lin, K, ctx.pad, _st())
            if ctx.x_cast:
                dx = dx[:, :, :Lin].float()
        return dx, dw, db, dgamma, dbeta

    @staticmethod
    def backward(ctx, dp):
        grads = ConvBlockFn._bwd_bf16(ctx, dp) if ctx.mode == _BF16 else ConvBlockFn._bwd_fp32(ctx, dp)
        return grads + (None,) * 12


# --------------------------------------------------------------------------------------
# Unfused leaves (the hook-compatible path: Conv1dFn, BatchNormFn, ReLUFn, MaxPool2Fn, GapFn, LinearFn, FilmFn) live in
# ecg_hip/leaves.py and are re-exported at the end of this module.
# --------------------------------------------------------------------------------------
class BceWithLogitsFn(torch.autograd.Function):
    """mean BCE-with-logits; the gradient is produced by the same launch as the loss."""

    @staticmethod
    def forward(ctx, logits, target, running, weight):
        logits, target = _contig(logits), _contig(target)
        if logits.shape != target.shape:
            raise ValueError(f"Target size ({tuple(target.shape)}) must be the same as input size ({tuple(logits.shape)})")
        if running is not None and (running.dtype != torch.float64 or running.numel() != 1):
            raise L.EcgHipError("running loss accumulator must be a float64 scalar tensor")
        loss = _empty(logits, 1)
        dx = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        _call("ecg_bce_logits_fwd", _f32(logits), _f32(target), _f32(loss), _f32(dx), logits.numel(),
              L.ptr(running), float(weight), _st())
        ctx.save_for_backward(dx)
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        (dx,) = ctx.saved_tensors
        one = _unit_grads.get(dloss.device)
        if one is not None and dloss.data_ptr() == one.data_ptr():
            return dx, None, None, None          # seeded by backward_from_loss(): d(loss)/d(loss) = 1 exactly
        return dx * dloss, None, None, None


# `loss.backward()` makes autograd fill a fresh ones_like(loss) and BceWithLogitsFn multiply its gradient by
# it: two tiny launches per step.  The training loops seed the backward pass with this cached constant 1
# instead; the multiply is then skipped (identical values, the tensor is never written).
_unit_grads = {}


def backward_from_loss(loss):
    """Same as `loss.backward()` for a scalar loss, without the ones_like fill and the x1 multiply."""
    one = _unit_grads.get(loss.device)
    if one is None:
        one = _unit_grads[loss.device] = torch.ones((), dtype=loss.dtype, device=loss.device)
    _sinks_handed.clear()                  # left over only if an earlier backward pass raised (no engine callback then)
    loss.backward(one)


# --------------------------------------------------------------------------------------
# functional entry points
# --------------------------------------------------------------------------------------
def conv_block(x, conv, bn, gap=False, packed=None):
    """Fused Conv1d -> BatchNorm1d -> ReLU -> MaxPool1d(2) [-> AdaptiveAvgPool1d(1).squeeze(-1)]."""
    return conv_block_chain(x, conv, bn, gap, packed)[0]


def conv_block_chain(x, conv, bn, gap=False, packed=None, carry=None, next_conv=None, next_bn=None):
    """conv_block for a chain of blocks: returns (p, carry).  In the mixed-precision form p is a bf16 activation
    [N][C][ld] (rows zero-filled past the pooled length) when the NEXT block of the chain (next_conv / next_bn given) takes
    that form too; the carry is then the true pooled length, to be handed to the next call — only the next block of the
    chain may consume such a tensor."""
    nxt = False
    if next_conv is not None and _conv_precision == "bf16" and not gap:
        x_len = carry if x.dtype == torch.bfloat16 else x.shape[2]
        Lo = x_len + 2 * conv.padding[0] - conv.kernel_size[0] + 1
        nxt = bool(next_bn is not None and (next_bn.training or next_bn.running_mean is None) and torch.is_grad_enabled()
                   and _bf16_block_ok(conv.out_channels, next_conv.out_channels, next_conv.kernel_size[0],
                                      next_conv.padding[0], Lo // 2, True))
    p = ConvBlockFn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                          bn.num_batches_tracked, bn.training, bn.momentum, bn.eps, conv.padding[0], gap, packed,
                          torch.is_grad_enabled(), carry, nxt)
    if p.dtype == torch.bfloat16:
        x_len = carry if x.dtype == torch.bfloat16 else x.shape[2]
        return p, (x_len + 2 * conv.padding[0] - conv.kernel_size[0] + 1) // 2
    return p, None


class TailFn(torch.autograd.Function):
    """Everything after the global average pool as one forward launch (+2 weight transposes) and
    two backward launches: proj [-> demographic MLP -> film_gen -> FiLM] -> head.
    reference src/models/ecg_cnn.py:63-64 and src/models/ecg_multimodal.py:88-99."""

    @staticmethod
    def forward(ctx, g, xd, Wp, bp, W0, b0, W2, b2, Wf, bf, Wh, bh, WpT=None, WfT=None):
        g = _contig(g)
        M, F0 = g.shape
        F, C = Wp.shape[0], Wh.shape[0]
        demo = xd is not None
        if WpT is None:
            WpT = _empty(g, F0, F)
            _call("ecg_transpose", _f32(_contig(Wp)), _f32(WpT), F, F0, _st())
        z, logits = _empty(g, M, F), _empty(g, M, C)
        D = H1 = H = 0
        xdc = h1 = h2 = film = zc = None
        if demo:
            xdc = _contig(xd)
            D, H1, H = xdc.shape[1], W0.shape[0], W2.shape[0]
            if WfT is None:
                WfT = _empty(g, H, 2 * F)
                _call("ecg_transpose", _f32(_contig(Wf)), _f32(WfT), 2 * F, H, _st())
            h1, h2, film, zc = _empty(g, M, H1), _empty(g, M, H), _empty(g, M, 2 * F), _empty(g, M, F)
        _call("ecg_tail_fwd", _f32(g), _f32(xdc), _f32(WpT), _f32(bp), _f32(W0), _f32(b0), _f32(W2),
              _f32(b2), _f32(WfT), _f32(bf), _f32(_contig(Wh)), _f32(bh), _f32(z), _f32(h1), _f32(h2),
              _f32(film), _f32(zc), _f32(logits), M, F0, F, D, H1, H, C, _st())
        ctx.save_for_backward(g, xdc, Wp, W0, W2, Wf, Wh, z, h1, h2, film, zc)
        ctx.dims = (M, F0, F, D, H1, H, C, demo)
        ctx.sink_keys = tuple(_key(t) for t in (Wp, bp, W0, b0, W2, b2, Wf, bf, Wh, bh))
        ctx.set_materialize_grads(False)       # an unused `z` output must not cost a zero-fill
        return logits, z

    @staticmethod
    def backward(ctx, dlogits, dz_extra):
        g, xd, Wp, W0, W2, Wf, Wh, z, h1, h2, film, zc = ctx.saved_tensors
        M, F0, F, D, H1, H, C, demo = ctx.dims
        if dlogits is None:
            dlogits = torch.zeros(M, C, dtype=torch.float32, device=g.device)
        dlogits = _contig(dlogits)
        dz, dg = _empty(g, M, F), _empty(g, M, F0)
        dzc = dfilm = dh2m = dh1m = dxd = None
        if demo:
            dzc, dfilm = _empty(g, M, F), _empty(g, M, 2 * F)
            dh2m, dh1m = _empty(g, M, H), _empty(g, M, H1)
            if ctx.needs_input_grad[1]:
                dxd = _empty(g, M, D)
        dze = None if dz_extra is None else _contig(dz_extra)
        _call("ecg_tail_bwd_chain", _f32(dlogits), _f32(dze), _f32(z), _f32(h1), _f32(h2), _f32(film),
              _f32(Wp), _f32(W0), _f32(W2), _f32(Wf), _f32(Wh), _f32(dzc), _f32(dz), _f32(dfilm),
              _f32(dh2m), _f32(dh1m), _f32(dg), _f32(dxd), M, F0, F, D, H1, H, C, int(demo), _st())
        k = ctx.sink_keys                       # (Wp, bp, W0, b0, W2, b2, Wf, bf, Wh, bh)
        dWp, dbp = _grad_out(k[0], g, *Wp.shape), _grad_out(k[1], g, F)
        dWh, dbh = _grad_out(k[8], g, *Wh.shape), _grad_out(k[9], g, C)
        if demo:
            dW0, db0 = _grad_out(k[2], g, *W0.shape), _grad_out(k[3], g, H1)
            dW2, db2 = _grad_out(k[4], g, *W2.shape), _grad_out(k[5], g, H)
            dWf, dbf = _grad_out(k[6], g, *Wf.shape), _grad_out(k[7], g, 2 * F)
            Gs, Xs = [dz, dfilm, dh2m, dh1m, dlogits], [g, h2, h1, xd, zc]
            dWs, dbs = [dWp, dWf, dW2, dW0, dWh], [dbp, dbf, db2, db0, dbh]
        else:
            dW0 = db0 = dW2 = db2 = dWf = dbf = None
            Gs, Xs, dWs, dbs = [dz, dlogits], [g, z], [dWp, dWh], [dbp, dbh]
        _call("ecg_linear_wgrad_grouped", L.ptr_table(Gs), L.ptr_table(Xs), L.ptr_table(dWs),
              L.ptr_table(dbs), L.int_table([w.shape[0] for w in dWs]),
              L.int_table([w.shape[1] for w in dWs]), len(Gs), M, _st())
        return dg, dxd, dWp, dbp, dW0, db0, dW2, db2, dWf, dbf, dWh, dbh, None, None


def tail(g, x_demo, proj, head, mlp0=None, mlp2=None, film_gen=None, transposed=None):
    """(logits, z) of the fused tail; x_demo/mlp0/mlp2/film_gen are None for ECGCNN.
    `transposed` = [proj.weight^T (, film_gen.weight^T)] when a WeightPacker already made them."""
    WpT = transposed[0] if transposed else None
    if x_demo is None:
        return TailFn.apply(g, None, proj.weight, proj.bias, None, None, None, None, None, None,
                            head.weight, head.bias, WpT, None)
    WfT = transposed[1] if transposed else None
    return TailFn.apply(g, x_demo, proj.weight, proj.bias, mlp0.weight, mlp0.bias, mlp2.weight,
                        mlp2.bias, film_gen.weight, film_gen.bias, head.weight, head.bias, WpT, WfT)


def film(z, film_raw):
    """(1 + tanh(gamma)) * z + beta with [gamma | beta] = film_raw (reference src/models/ecg_multimodal.py:92-96)."""
    if not z.is_cuda:
        gamma, beta = film_raw.chunk(2, dim=-1)
        return (1.0 + torch.tanh(gamma)) * z + beta
    return FilmFn.apply(z, film_raw)


def binary_cross_entropy_with_logits(logits, target, running=None, weight=1.0):
    """Drop-in for F.binary_cross_entropy_with_logits(logits, y) (mean reduction only).
    `running` (optional float64 scalar on the device) receives `+= loss * weight` inside the same
    launch: the loops' epoch-loss bookkeeping without a host sync or extra kernels.
    CPU tensors take stock torch (the reference's own call, src/training/loop.py:32)."""
    if not logits.is_cuda:
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, target)
        if running is not None:
            running += loss.detach().double() * weight
        return loss
    return BceWithLogitsFn.apply(logits, target, running, weight)


def sigmoid(x):
    if not x.is_cuda:
        return torch.sigmoid(x.detach())
    x = _contig(x.detach())
    out = torch.empty_like(x)
    _call("ecg_sigmoid_fwd", _f32(x), _f32(out), x.numel(), _st())
    return out


def zscore_per_lead(x, out=None, return_stats=False):
    """(x - mean)/(std + 1e-6) per lead row, population std, bit-identical to the reference's float32
    numpy arithmetic (src/datasets/ptbxl.py:122-127).  `out` may alias x (in place)."""
    x = _contig(x)
    T = x.shape[-1]
    rows = x.numel() // T
    if out is None:
        out = torch.empty_like(x)
    stats = _empty(x, rows, 2)
    _call("ecg_zscore_rows", _f32(x), _f32(out), _f32(stats), rows, T, _st())
    return (out, stats) if return_stats else out


def wfdb16_to_windows(d, gain, baseline, normalize=True, return_stats=False):
    """WFDB format-16 samples d int16 [B, T, leads] (time-major, as stored in .dat) + per-(window, lead)
    gain (float64) / baseline (int32) [B, leads]  ->  fp32 windows [B, leads, T], per-lead z-scored:
    `_load_ecg` + `_normalize` of the reference (src/datasets/ptbxl.py:14-41,122-127) on the GPU, one
    launch.  normalize=False stops at the physical signal."""
    if d.dtype != torch.int16 or gain.dtype != torch.float64 or baseline.dtype != torch.int32:
        raise L.EcgHipError("wfdb16_to_windows: d must be int16, gain float64, baseline int32")
    d, gain, baseline = _contig(d), _contig(gain), _contig(baseline)
    B, T, leads = d.shape
    if tuple(gain.shape) != (B, leads) or tuple(baseline.shape) != (B, leads):
        raise L.EcgHipError("wfdb16_to_windows: gain/baseline must be [B, leads]")
    x = torch.empty(B, leads, T, dtype=torch.float32, device=d.device)
    if not normalize:
        _call("ecg_wfdb16_physical", L.ptr(d), L.ptr(gain), L.ptr(baseline), _f32(x), B, T, leads, _st())
        return x
    stats = _empty(x, B * leads, 2)
    _call("ecg_wfdb16_zscore", L.ptr(d), L.ptr(gain), L.ptr(baseline), _f32(x), _f32(stats), B, T, leads, _st())
    return (x, stats) if return_stats else x


from .leaves import BatchNormFn, Conv1dFn, FilmFn, GapFn, LinearFn, MaxPool2Fn, ReLUFn  # noqa: E402,F401
