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
"""
import torch

from . import functional as hipF
from .optim import FlatAdamW


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_batch, warmup=3):
        """example_batch = (*inputs, target) on the GPU; its shapes become the static shapes."""
        if not isinstance(optimizer, FlatAdamW):
            raise TypeError("GraphedTrainStep needs ecg_hip.optim.FlatAdamW (device-side step counter)")
        if optimizer.world_size > 1:
            raise ValueError("GraphedTrainStep does not capture collectives: use it single-GPU")
        self.model, self.optimizer = model, optimizer.make_capturable()
        self.static = [t.clone() for t in example_batch]
        self.running = torch.zeros((), dtype=torch.float64, device=self.static[0].device)
        self.loss = None
        model.train()
        # the warm-up iterations below are REAL optimizer steps on the example batch: snapshot everything they
        # advance (parameters, AdamW moments, step counter, BatchNorm running statistics and counters) and put
        # it back before capture, so that building the graphed step on a loaded checkpoint leaves it untouched
        opt = self.optimizer
        snap = [(t, t.detach().clone()) for t in (opt.flat_param, opt.flat_m, opt.flat_v, opt._step_dev)]
        snap += [(b, b.detach().clone()) for b in model.buffers()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up outside capture (allocator, pack caches)
            for _ in range(warmup):
                self._step_body()
            with torch.no_grad():
                for t, saved in snap:
                    t.copy_(saved)
            self.running.zero_()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # capture on the stream the warm-up ran on: the AccumulateGrad nodes created there keep that stream, and
        # a capture on another one makes autograd insert (and warn about) cross-stream synchronisation
        with torch.cuda.graph(self.graph, stream=side):
            self._step_body()

    def _step_body(self):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(*self.static[:-1])
        logits = out[0] if isinstance(out, tuple) else out
        self.loss = hipF.binary_cross_entropy_with_logits(logits, self.static[-1], self.running, 1.0)
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
