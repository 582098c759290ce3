"""Whole-train-step HIP graph.

The fp32 step at batch 256 is GPU-bound (the host enqueues ~50 launches in ~0.95 ms against
~1.9 ms of kernels), but at the reference's own batch sizes (64 / 32 windows of 12x1000) the
Python/ctypes enqueue time IS the step time.  `GraphedTrainStep` captures one complete step —
zero_grad, forward, BCE, backward, optimizer — into a hipGraph (torch.cuda.CUDAGraph) and
replays it with new data copied into static input buffers: one host call per step.

Requirements: static shapes; a capturable optimizer (`FlatAdamW.make_capturable()`, whose step
counter lives on the device); no hooks that synchronise.  BatchNorm counters, running statistics
and the optional running-loss accumulator are all updated by kernels, so replay keeps them right.
Not used with a process group (collectives are left out of the capture).

`LoopStepper` is how the UNCHANGED loop API gets this (opt-in: ECG_HIP_LOOP_GRAPH=1, see loop_graph_enabled):
`train_one_epoch[_demo]` ask it for every batch; once a batch shape has been seen twice it builds a GraphedTrainStep
for that shape and replays it, anything else (ragged last batch, a hooked model, a stock optimizer, several ranks)
runs the eager step.
"""
import os

import torch

from . import functional as hipF
from .optim import FlatAdamW


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_batch, warmup=3, loss_weight=1.0, running=None):
        """example_batch = (*inputs, target) on the GPU; its shapes become the static shapes.  Every replay adds
        `loss * loss_weight` to `running` (a float64 device scalar; a private one when None)."""
        if not isinstance(optimizer, FlatAdamW):
            raise TypeError("GraphedTrainStep needs ecg_hip.optim.FlatAdamW (device-side step counter)")
        if optimizer.world_size > 1:
            raise ValueError("GraphedTrainStep does not capture collectives: use it single-GPU")
        self.model, self.optimizer = model, optimizer.make_capturable()
        self.static = [t.clone() for t in example_batch]
        self.running = (torch.zeros((), dtype=torch.float64, device=self.static[0].device) if running is None
                        else running)
        self.loss_weight = float(loss_weight)
        self.loss = None
        model.train()
        # the warm-up iterations below are REAL optimizer steps on the example batch: snapshot everything they
        # advance (parameters, AdamW moments, step counter, BatchNorm running statistics and counters) and put
        # it back before capture, so that building the graphed step on a loaded checkpoint leaves it untouched
        opt = self.optimizer
        snap = [(t, t.detach().clone()) for t in (opt.flat_param, opt.flat_m, opt.flat_v, opt._step_dev)]
        snap += [(b, b.detach().clone()) for b in model.buffers()]
        snap.append((self.running, self.running.clone()))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up outside capture (allocator, pack caches)
            for _ in range(warmup):
                self._step_body()
            with torch.no_grad():
                for t, saved in snap:
                    t.copy_(saved)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # capture on the stream the warm-up ran on: the AccumulateGrad nodes created there keep that stream, and
        # a capture on another one makes autograd insert (and warn about) cross-stream synchronisation
        with torch.cuda.graph(self.graph, stream=side):
            self._step_body()
        # keep the loss VALUE (static memory of the graph's pool), drop the autograd graph of the captured step: the
        # AccumulateGrad nodes it holds are bound to the capture stream, and eager steps that follow on another
        # stream would be synchronised against it for nothing
        self.loss = self.loss.detach()

    def _step_body(self):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(*self.static[:-1])
        logits = out[0] if isinstance(out, tuple) else out
        self.loss = hipF.binary_cross_entropy_with_logits(logits, self.static[-1], self.running, self.loss_weight)
        hipF.backward_from_loss(self.loss)
        self.optimizer.step()

    def __call__(self, *batch):
        """One train step on `batch` (same shapes as the example).  Returns the loss tensor of the
        replayed step (device scalar; reading it synchronises)."""
        for dst, src in zip(self.static, batch):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss

    def mean_loss_and_reset(self, steps):
        """Mean of the per-step losses since the last reset (one device read)."""
        v = self.running.item() / max(1, steps)
        self.running.zero_()
        return v


# --------------------------------------------------------------------------------------------------------------
# the loop API's way in
# --------------------------------------------------------------------------------------------------------------
def loop_graph_enabled():
    """ECG_HIP_LOOP_GRAPH=1 lets train_one_epoch[_demo] replay captured steps; the default is the eager step.
    Why opt-in (measured on MI355X, round 3, `bench.py` `ref_batch_sizes`): with 33 launches per step the eager loop is
    no longer host-bound even at the reference's batch sizes — B=32 12x1000: 0.591 ms eager vs 0.605 ms replayed,
    B=32 12x5000: 1.276 vs 1.296, B=64 12x5000: 2.110 vs 2.128, B=256 12x1000: 1.640 vs 1.659 — the replay pays for
    the copies into its static input buffers and gains nothing while the GPU is the bottleneck.  It frees the host
    thread (one call per step instead of ~45), which matters only when the input pipeline needs that thread."""
    return os.environ.get("ECG_HIP_LOOP_GRAPH", "0") == "1"


_HOOK_DICTS = ("_forward_hooks", "_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks")


def _has_hooks(model):
    import torch.nn.modules.module as M
    for name in ("_global_forward_hooks", "_global_forward_pre_hooks", "_global_backward_hooks",
                 "_global_backward_pre_hooks"):
        if getattr(M, name, None):
            return True
    for m in model.modules():
        for d in _HOOK_DICTS:
            if getattr(m, d, None):
                return True
    return False


class LoopStepper:
    """Per-(model, optimizer) cache of GraphedTrainStep objects, kept on the optimizer between epochs.

    `train_one_epoch` at the reference's own batch sizes (configs/ecg_baseline.yaml:12 batch 64,
    configs/af_binary.yaml:8 batch 32) is bound by the ~0.7 ms of Python/ctypes enqueue per step, not by the GPU
    (src/training/loop.py:22-36 is the loop being replaced).  Replaying the captured step needs one host call.
    A shape is captured when it shows up for the SECOND time (an epoch of one batch never pays the capture; the ragged
    last batch of an epoch runs eagerly at most once per epoch before it, too, is worth a graph).  Eligible only for:
    FlatAdamW on CUDA, one rank, no module hooks (a hook may read or sync), every trainable parameter owned by the
    optimizer, grad mode on, no kernel timing in progress.  Values are bit-identical to the eager step: the same
    kernels in the same order."""

    MAX_GRAPHS = 4            # distinct batch shapes captured per (model, optimizer): full batch + ragged tails

    def __init__(self, model, optimizer, loss_weight_is_batch):
        self.model, self.optimizer = model, optimizer
        self.loss_weight_is_batch = bool(loss_weight_is_batch)
        self.running = torch.zeros((), dtype=torch.float64, device=optimizer.flat_param.device)
        self.seen, self.graphs = {}, {}

    @staticmethod
    def eligible(model, optimizer):
        from . import _lib
        if not loop_graph_enabled() or not isinstance(optimizer, FlatAdamW):
            return False
        if optimizer.world_size > 1 or optimizer._exchange or not optimizer.flat_param.is_cuda:
            return False
        if _lib._events is not None or not torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
            return False
        owned = {id(p) for p in optimizer._params}
        if any(p.requires_grad and id(p) not in owned for p in model.parameters()):
            return False
        return not _has_hooks(model)

    @classmethod
    def for_loop(cls, model, optimizer, loss_weight_is_batch):
        """The stepper of this (model, optimizer) pair with its loss accumulator zeroed, or None when the loop
        must stay eager."""
        if not cls.eligible(model, optimizer):
            return None
        cache = optimizer.__dict__.setdefault("_ecg_loop_steppers", {})
        key = (id(model), bool(loss_weight_is_batch))
        st = cache.get(key)
        if st is None or st.model is not model:
            st = cache[key] = cls(model, optimizer, loss_weight_is_batch)
        st.running.zero_()
        return st

    def _key(self, batch):
        grp = self.optimizer.param_groups[0]
        return (tuple((tuple(t.shape), t.dtype) for t in batch), hipF.get_conv_precision(),
                float(grp["lr"]), tuple(grp["betas"]), float(grp["eps"]), float(grp["weight_decay"]),
                self.optimizer.flat_param.data_ptr())

    def step(self, batch):
        """Run one train step on `batch` = (*inputs, target) through a captured graph if this shape has one (or has
        earned one); returns False when the caller must run the eager step itself (into `self.running`)."""
        if any(not t.is_cuda for t in batch):
            return False
        key = self._key(batch)
        g = self.graphs.get(key)
        if g is None:
            n = self.seen[key] = self.seen.get(key, 0) + 1
            if n < 2:
                return False
            if len(self.graphs) >= self.MAX_GRAPHS:            # e.g. a scheduler changed lr: drop the oldest capture
                self.graphs.pop(next(iter(self.graphs)))
            w = float(batch[0].shape[0]) if self.loss_weight_is_batch else 1.0
            g = self.graphs[key] = GraphedTrainStep(self.model, self.optimizer, batch, loss_weight=w, running=self.running)
        g(*batch)
        return True
