"""FlatAdamW — torch.optim.AdamW semantics (defaults betas=(0.9, 0.999), eps=1e-8, decoupled
weight decay on every parameter: reference scripts/03_train_ecg_baseline.py:130-133) as ONE
launch over one flat fp32 buffer (ecg_adamw_step), instead of 20-28 per-tensor updates.

On construction the parameters are re-pointed as views into a single contiguous buffer
(values unchanged, so state_dicts and checkpoints are unaffected).  `step()` gathers the
per-parameter gradients into a flat gradient with one concatenation, optionally all-reduces
it over a process group (the data-parallel exchange of SURVEY §8e: RCCL all-reduce of 3 MB per
step) and applies the update; averaging by 1/world is folded into the kernel.

With a process group the exchange is split in two buckets so that it hides under backward:
parameters are in forward order, gradients arrive in reverse, so as soon as every gradient of the
"late" bucket (everything after the first `early_params` tensors — blocks 2-3 and the tail, 95 %
of the bytes) has been accumulated, it is gathered and all-reduced asynchronously while the
backward pass of blocks 1 and 0 is still running; step() only exposes the small early bucket.
"""
import torch
import torch.optim.optimizer as _opt_mod

from . import _lib as L


def flatten_tensors_(tensors):
    """Re-point each tensor's storage into one flat buffer (same dtype/device); returns it."""
    tensors = list(tensors)
    if not tensors:
        raise ValueError("nothing to flatten")
    dev, dt = tensors[0].device, tensors[0].dtype
    if any(t.device != dev or t.dtype != dt for t in tensors):
        raise ValueError("all tensors must share device and dtype")
    flat = torch.empty(sum(t.numel() for t in tensors), dtype=dt, device=dev)
    off = 0
    for t in tensors:
        n = t.numel()
        view = flat[off:off + n].view(t.shape)
        view.copy_(t.detach())
        t.data = view
        off += n
    if flat.is_cuda:
        from . import functional as F
        F.parameters_rehomed()           # statements keyed by parameter object stay; their addresses are looked up afresh
    return flat


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 process_group=None, world_size=None, overlap=True, early_params=8, exchange_single_rank=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdamW keeps one flat buffer: pass a single parameter group "
                             "(the reference passes model.parameters())")
        self._params = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        for p in self._params:
            if p.dtype != torch.float32:
                raise L.EcgHipError("FlatAdamW: float32 parameters only")
        self.flat_param = flatten_tensors_(self._params)
        self.flat_m = torch.zeros_like(self.flat_param)
        self.flat_v = torch.zeros_like(self.flat_param)
        # persistent flat gradient; on the GPU the backward kernels write each parameter's gradient straight
        # into its slice (functional.register_grad_sinks), so the per-step gather usually has nothing to do
        self.flat_grad = torch.zeros_like(self.flat_param)
        self._gptrs, self._offs, views, off = [], [0], [], 0
        for p in self._params:
            v = self.flat_grad[off:off + p.numel()]
            views.append(v)
            self._gptrs.append(v.data_ptr())
            off += p.numel()
            self._offs.append(off)
        self._sink_keys = []
        if self.flat_param.is_cuda:
            from . import functional as F
            F.register_grad_sinks(self._params, views)
            self._sink_keys = [p.data_ptr() for p in self._params]
        self._step = 0
        self._step_dev = None            # device-side step counter (graph-capturable mode)
        self.process_group = process_group
        if world_size is None:
            world_size = 1
            if process_group is not None or (torch.distributed.is_available()
                                             and torch.distributed.is_initialized()):
                world_size = torch.distributed.get_world_size(process_group)
        self.world_size = world_size
        # `exchange_single_rank`: run the exchange (hooks, async all-reduces, waits) even when the group has ONE rank —
        # the sum over one rank is the identity, so the step is unchanged bit for bit; it exists so that the RCCL
        # code path can be executed on a one-GPU box (tools/rccl_selftest.py)
        self._exchange = world_size > 1 or (bool(exchange_single_rank) and torch.distributed.is_available()
                                            and torch.distributed.is_initialized())
        # ---- bucketed, backward-overlapped exchange (world > 1 only) -------------------------
        self._n_early = min(max(int(early_params), 0), len(self._params))
        self._split = sum(p.numel() for p in self._params[:self._n_early])      # flat offset of the late bucket
        self._late_pending = 0
        self._late_work = None
        self._early_pending = 0
        self._early_work = None
        self._overlap = bool(overlap) and self._exchange and 0 < self._n_early < len(self._params)
        self._sync = True                 # False inside no_sync(): gradients accumulate locally, nothing is exchanged
        self._overlap_active = self._overlap          # set_overlap(): the hooks stay registered, this switches them
        if self._overlap:
            self._late_total = len(self._params) - self._n_early
            for p in self._params[self._n_early:]:
                p.register_post_accumulate_grad_hook(self._on_late_grad)
            # the early bucket cannot hide under backward (its gradients are the last to arrive), but its
            # all-reduce can at least be issued from the hook of its last gradient instead of waiting for
            # Python to come back from backward() and reach step()
            for p in self._params[:self._n_early]:
                p.register_post_accumulate_grad_hook(self._on_early_grad)
        self._note_overlap()

    def __del__(self):
        try:
            from . import functional as F
            F.unregister_grad_sinks(self._sink_keys)
        except Exception:
            pass
        try:
            if self.flat_param.is_cuda:
                from . import functional as F
                F.declare_backward_collectives(self._params, None, owner=id(self))     # only what this optimizer declared
        except Exception:
            pass

    @staticmethod
    def _flat(p):
        g = p.grad
        return torch.zeros_like(p).view(-1) if g is None else g.reshape(-1)

    def _gather(self, lo, hi):
        """Make the flat gradient hold the gradients of parameters lo..hi-1.  Nothing to do when every .grad
        already IS its slice of the flat buffer (written there by the backward kernels); one concatenation
        when none is; element-wise repair when only some are (a cat must not read what it overwrites)."""
        ps = self._params[lo:hi]
        placed = [p.grad is not None and p.grad.data_ptr() == q for p, q in zip(ps, self._gptrs[lo:hi])]
        if all(placed):
            return
        if not any(placed):
            torch.cat([self._flat(p) for p in ps], out=self.flat_grad[self._offs[lo]:self._offs[hi]])
            return
        for i, (p, ok) in enumerate(zip(ps, placed), start=lo):
            if ok:
                continue
            dst = self.flat_grad[self._offs[i]:self._offs[i + 1]]
            if p.grad is None:
                dst.zero_()
            else:
                dst.copy_(p.grad.reshape(-1))

    def set_overlap(self, on):
        """Switch between the bucketed exchange issued from the backward hooks (True; needs `overlap=True` at
        construction) and ONE all-reduce of the whole flat gradient in step() (False).  Which is faster depends on the
        node: the hooked all-reduces hide under the backward of the early blocks, but RCCL's kernel then shares the
        CUs with conv launches that are sized to fill the chip in exactly one round (measured on one rank: +55 us per
        step hooked, +4 us single).  Call it between steps, on every rank alike; returns the mode now in force."""
        if self._late_work is not None or self._early_work is not None:
            raise RuntimeError("FlatAdamW.set_overlap: an exchange is in flight (call it between steps)")
        self._overlap_active = bool(on) and self._overlap
        self._late_pending = self._early_pending = 0
        self._note_overlap()
        return self._overlap_active

    def _note_overlap(self):
        # Tell the backward kernels, per parameter, whether collectives can run beside them: the hooked exchange issues
        # all-reduces UNDER backward ("busy": the BatchNorm backward of these blocks keeps its two-pass form, whose progress
        # never depends on a communication kernel leaving a CU); the single all-reduce in step() starts after the last
        # backward kernel ("quiet": the one-launch form).  Without an exchange nothing is declared.
        if self.flat_param.is_cuda and self._exchange:
            from . import functional as F
            F.declare_backward_collectives(self._params, bool(self._overlap_active), owner=id(self))

    def calibrate_overlap(self, run_steps, steps=10, warm=3):
        """Pick the faster exchange form ON THIS NODE: `run_steps(n)` must run n complete train steps (forward,
        backward, step()) with this optimizer.  Both forms are timed (wall clock around `steps` steps after `warm`
        untimed ones, device synchronised); the decision is taken on the MAX over ranks, which the all-reduce makes
        identical everywhere, so every rank ends up in the same mode.  Returns {"mode", "ms_per_step": {...}}.
        A no-op ({"mode": "single"}) when there is nothing to choose (one rank, or overlap=False at construction)."""
        import time
        if not self._overlap:
            return {"mode": "single", "ms_per_step": {}}
        dist = torch.distributed
        dev = self.flat_param.device
        res = {}
        for mode in (True, False):
            self.set_overlap(mode)
            run_steps(warm)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            if dev.type == "cuda" and dist.get_backend(self.process_group) == "nccl":
                dist.barrier(group=self.process_group, device_ids=[dev.index])
            else:
                dist.barrier(group=self.process_group)
            t0 = time.perf_counter()
            run_steps(steps)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            t = torch.tensor([(time.perf_counter() - t0) / steps], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.process_group)
            res["overlapped" if mode else "single"] = round(t.item() * 1e3, 4)
        mode = "overlapped" if res["overlapped"] <= res["single"] else "single"
        self.set_overlap(mode == "overlapped")
        return {"mode": mode, "ms_per_step": res}

    def no_sync(self):
        """Context manager for gradient accumulation over several backward passes per step (as
        DistributedDataParallel.no_sync): inside it the hooked exchange is off and gradients only accumulate
        locally; the first backward outside it (or step() itself) exchanges the accumulated gradient once."""
        opt = self

        class _NoSync:
            def __enter__(self):
                self.prev, opt._sync = opt._sync, False

            def __exit__(self, *exc):
                opt._sync = self.prev
                return False

        return _NoSync()

    def _refuse_second_exchange(self):
        raise RuntimeError(
            "FlatAdamW: a second backward pass completed a gradient bucket whose all-reduce from the previous pass "
            "has not been consumed by step(): the bucket already holds the sum over ranks, so adding local "
            "gradients to it and reducing again would give world*g1 + g2.  Wrap all but the last backward of a "
            "step in `optimizer.no_sync()` (gradient accumulation), or call step() between the passes.")

    def _on_late_grad(self, _param):
        """Autograd hook: when the last gradient of the late bucket lands, gather that bucket into
        the flat gradient and start its all-reduce; backward of the early blocks keeps running."""
        if not self._sync or not self._overlap_active:
            return
        if self._late_work is not None:
            self._refuse_second_exchange()
        self._late_pending += 1
        if self._late_pending < self._late_total:
            return
        self._late_pending = 0
        with torch.no_grad():
            self._gather(self._n_early, len(self._params))
            late = self.flat_grad[self._split:]
            self._late_work = torch.distributed.all_reduce(late, group=self.process_group, async_op=True)

    def _on_early_grad(self, _param):
        if not self._sync or not self._overlap_active:
            return
        if self._early_work is not None:
            self._refuse_second_exchange()
        self._early_pending += 1
        if self._early_pending < self._n_early:
            return
        self._early_pending = 0
        with torch.no_grad():
            self._gather(0, self._n_early)
            self._early_work = torch.distributed.all_reduce(self.flat_grad[:self._split], group=self.process_group,
                                                            async_op=True)

    @torch.no_grad()
    def reduce_gradients(self):
        """Flat gradient of this step, summed over ranks; returns (flat, scale) where
        scale = 1/world is applied inside the AdamW kernel.  One all-reduce, or two buckets when the
        late bucket was already launched from the backward hooks."""
        if self._late_work is not None:                       # late bucket is in flight / done
            g = self.flat_grad
            if self._early_work is not None:                  # issued from the hook of its last gradient
                self._early_work.wait()
                self._early_work = None
            else:                                             # some early gradient never arrived (unused parameter)
                self._early_pending = 0
                self._gather(0, self._n_early)
                torch.distributed.all_reduce(g[:self._split], group=self.process_group)
            self._late_work.wait()
            self._late_work = None
            return g, 1.0 / self.world_size
        self._late_pending = 0
        self._early_pending = 0
        if self._early_work is not None:                      # early bucket went out but the late one did not: finish it,
            self._early_work.wait()                           # then fall through to one all-reduce of everything else
            self._early_work = None
            g = self.flat_grad
            self._gather(self._n_early, len(self._params))
            if self._exchange:
                torch.distributed.all_reduce(g[self._split:], group=self.process_group)
            return g, 1.0 / self.world_size
        g = self.flat_grad
        self._gather(0, len(self._params))
        if self._exchange:
            torch.distributed.all_reduce(g, group=self.process_group)
            return g, 1.0 / self.world_size
        return g, 1.0

    def zero_grad(self, set_to_none=True):
        self._forget_handed_sinks()
        return super().zero_grad(set_to_none=set_to_none)

    def _forget_handed_sinks(self):
        """A backward pass that raised leaves functional._sinks_handed non-empty (the autograd engine runs no
        end-of-pass callbacks then) and every later pass would get private gradient tensors instead of the flat-buffer
        sinks, silently.  A new step starts from a clean slate."""
        if self._sink_keys:
            from . import functional as F
            F._sinks_handed.clear()

    @torch.no_grad()
    def step(self, closure=None):
        self._forget_handed_sinks()
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grp = self.param_groups[0]
        g, scale = self.reduce_gradients()
        b1, b2 = grp["betas"]
        if self._step_dev is not None:        # capturable: the kernel reads and bumps the device counter
            L.call("ecg_adamw_step_graph", L.f32(self.flat_param), L.f32(g), L.f32(self.flat_m),
                   L.f32(self.flat_v), g.numel(), L.ptr(self._step_dev), float(grp["lr"]), float(b1),
                   float(b2), float(grp["eps"]), float(grp["weight_decay"]), scale, L.stream())
            return loss
        self._step += 1
        if g.is_cuda:
            L.call("ecg_adamw_step", L.f32(self.flat_param), L.f32(g), L.f32(self.flat_m),
                   L.f32(self.flat_v), g.numel(), self._step, float(grp["lr"]), float(b1), float(b2),
                   float(grp["eps"]), float(grp["weight_decay"]), scale, L.stream())
        else:
            self._step_cpu(g, grp, scale)
        return loss

    def _step_cpu(self, g, grp, scale):
        """CPU parameters (a GPU-less box): the same update with stock torch ops on the flat buffers —
        torch.optim.AdamW's single-tensor formula (decoupled decay, bias-corrected moments)."""
        b1, b2 = grp["betas"]
        lr, eps, wd, t = float(grp["lr"]), float(grp["eps"]), float(grp["weight_decay"]), self._step
        if scale != 1.0:
            g = g * scale
        p, m, v = self.flat_param, self.flat_m, self.flat_v
        p.mul_(1.0 - lr * wd)
        m.lerp_(g, 1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)

    def make_capturable(self):
        """Move the step counter to the device so `step()` can be captured in a HIP graph."""
        if self._step_dev is None:
            self._step_dev = torch.full((1,), self._step, dtype=torch.int32, device=self.flat_param.device)
        return self

    @property
    def steps_taken(self):
        return int(self._step_dev.item()) if self._step_dev is not None else self._step

    # checkpointing of optimizer state (the reference saves none; kept for completeness)
    def state_dict(self):
        return {"step": self.steps_taken, "exp_avg": self.flat_m, "exp_avg_sq": self.flat_v,
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self._step = int(sd["step"])
        if self._step_dev is not None:
            self._step_dev.fill_(self._step)
        self.flat_m.copy_(sd["exp_avg"])
        self.flat_v.copy_(sd["exp_avg_sq"])
        self.param_groups[0].update(sd["param_groups"][0])


# --------------------------------------------------------------------------------------
# Adoption of the stock optimizer the reference's scripts construct (scripts/03_train_ecg_baseline.py:133,
# 04_train_multimodal_prototype.py:158-162, 05_train_af_binary.py:130: `AdamW(model.parameters(), lr=lr, weight_decay=wd)`).
# Stock torch.optim.AdamW steps through ~10 foreach launches over 20-28 tensors and its gradients are 20-28 separate
# tensors: 0.08 ms of host gap per step at B = 256 (round 4: 149.5 k windows/s against 160.7 k with FlatAdamW).
# `adopt_stock_adamw` turns such an optimizer, IN PLACE, into one that steps through the same fused launch as FlatAdamW:
#   * parameters become views of one flat buffer (values unchanged), gradients are written by the backward kernels straight
#     into views of one flat gradient (functional.register_grad_sinks);
#   * `optimizer.state[p]` holds what stock AdamW would hold — `step` (a CPU scalar tensor), `exp_avg`, `exp_avg_sq` — the
#     two moments being views of flat buffers, so `optimizer.state_dict()` is a stock AdamW state_dict: it loads into a fresh
#     `torch.optim.AdamW` (and one saved by stock AdamW loads here) and training continues from it;
#   * the object keeps its identity and stays an instance of torch.optim.AdamW (its class becomes a subclass), param_groups
#     (lr, betas, eps, weight_decay) are read at every step, so LR schedulers built afterwards work unchanged.
# Anything the fused launch does not cover keeps stock behaviour: a step in which some parameter has no gradient (stock
# AdamW skips such a parameter entirely: no decay, no moment update) runs torch's own step on the same views.
# --------------------------------------------------------------------------------------
def _adoptable(opt, allow_cpu=False):
    if type(opt) is not torch.optim.AdamW or "step" in vars(opt):      # (an LR scheduler built EARLIER has bound the stock step)
        return False
    if len(opt.param_groups) != 1:
        return False
    g = opt.param_groups[0]
    if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
        return False
    if not g.get("decoupled_weight_decay", True):
        return False
    if any(torch.is_tensor(v) for v in (g["lr"], g["eps"], g["weight_decay"], *g["betas"])):
        return False
    ps = g["params"]
    if not ps or any((not p.requires_grad) or p.dtype != torch.float32 or p.device != ps[0].device or p.is_sparse for p in ps):
        return False
    if not (ps[0].is_cuda or allow_cpu):
        return False
    st = [opt.state.get(p) for p in ps]
    if any(s for s in st):           # already stepped by stock torch: adopt its state — only a uniform one fits one launch
        if not all(s and set(s) >= {"step", "exp_avg", "exp_avg_sq"} for s in st):
            return False
        if len({float(s["step"]) for s in st}) != 1:
            return False
    return True


class AdoptedAdamW(torch.optim.AdamW):
    """What `adopt_stock_adamw` turns a stock torch.optim.AdamW into (never constructed directly)."""

    def _adopt(self):
        grp = self.param_groups[0]
        ps = list(grp["params"])
        old = [dict(self.state.get(p, {})) for p in ps]
        self._params = ps
        self.flat_param = flatten_tensors_(ps)
        self.flat_m, self.flat_v = torch.zeros_like(self.flat_param), torch.zeros_like(self.flat_param)
        self.flat_grad = torch.zeros_like(self.flat_param)
        self._offs, self._gptrs, self._pptrs, self._steps, gviews, off = [0], [], [], [], [], 0
        fused_steps = bool(grp.get("fused"))
        for p, s in zip(ps, old):
            n = p.numel()
            m, v = self.flat_m[off:off + n].view(p.shape), self.flat_v[off:off + n].view(p.shape)
            if s:
                m.copy_(s["exp_avg"]); v.copy_(s["exp_avg_sq"])
                step = s["step"]
            else:               # as Adam._init_group creates it: a CPU scalar unless the group asked for a fused step
                step = (torch.zeros((), dtype=torch.float32, device=p.device) if fused_steps
                        else torch.tensor(0.0, dtype=torch.get_default_dtype()))
            self.state[p] = {"step": step, "exp_avg": m, "exp_avg_sq": v}
            self._steps.append(step)
            gv = self.flat_grad[off:off + n]
            gviews.append(gv)
            self._gptrs.append(gv.data_ptr())
            self._pptrs.append(p.data_ptr())
            off += n
            self._offs.append(off)
        self._step_count = int(float(self._steps[0]))
        self._uniform = True
        self._sink_keys = []
        if self.flat_param.is_cuda:
            from . import functional as F
            if not any(k in F._grad_sinks for k in self._pptrs):     # (a FlatGradDDP-style owner may hold them already)
                F.register_grad_sinks(ps, gviews)
                self._sink_keys = list(self._pptrs)
                import weakref
                weakref.finalize(self, F.unregister_grad_sinks, list(self._sink_keys))

    _gather = FlatAdamW._gather
    _flat = staticmethod(FlatAdamW._flat)
    _step_cpu = FlatAdamW._step_cpu

    def _rehome_if_moved(self):
        """model.to(...) / p.data = ... after adoption re-points parameters: take them back into a fresh flat buffer
        (values and moments kept).  Returns False when they left the device the moments live on: stock behaviour then."""
        if all(p.data_ptr() == q for p, q in zip(self._params, self._pptrs)):
            return True
        if any(p.device != self.flat_m.device or p.dtype != torch.float32 for p in self._params):
            return False
        if self._sink_keys:
            from . import functional as F
            F.unregister_grad_sinks(self._sink_keys)
        self._adopt()          # (gradients that are views of the old flat gradient stay valid: the views keep it alive)
        return True

    def zero_grad(self, set_to_none=True):
        if self._sink_keys:
            from . import functional as F
            F._sinks_handed.clear()
        return super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def _fused_step(self):
        grp = self.param_groups[0]
        self._gather(0, len(self._params))
        self._step_count += 1
        b1, b2 = grp["betas"]
        if self.flat_param.is_cuda:
            L.call("ecg_adamw_step", L.f32(self.flat_param), L.f32(self.flat_grad), L.f32(self.flat_m), L.f32(self.flat_v),
                   self.flat_grad.numel(), self._step_count, float(grp["lr"]), float(b1), float(b2), float(grp["eps"]),
                   float(grp["weight_decay"]), 1.0, L.stream())
        else:
            self._step = self._step_count
            self._step_cpu(self.flat_grad, grp, 1.0)
        torch._foreach_add_(self._steps, 1)

    def step(self, closure=None):
        if closure is not None or not self._uniform or len(self.param_groups) != 1 \
                or len(self.param_groups[0]["params"]) != len(self._params) or not self._rehome_if_moved() \
                or any(p.grad is None for p in self._params):
            # stock semantics for everything the one launch does not cover; the state tensors are the flat views, so
            # torch's own step updates them in place and the fused form can resume when the step counts are still uniform
            out = _raw_stock_step()(self, closure)
            steps = {float(s["step"]) for s in (self.state.get(p) for p in self._params) if s}
            self._uniform = (len(steps) == 1 and all(self.state.get(p) for p in self._params)
                             and len(self.param_groups) == 1 and len(self.param_groups[0]["params"]) == len(self._params))
            if self._uniform:
                self._step_count = int(steps.pop())
            return out
        if self._sink_keys:
            from . import functional as F
            F._sinks_handed.clear()
        self._fused_step()
        return None

    # step pre / post hooks run as for any torch optimizer (Optimizer.__setstate__ would add this wrapper on its own the
    # first time a state_dict is loaded; it is here from the start so that the hooks never run twice)
    step = torch.optim.Optimizer.profile_hook_step(step)
    step.hooked = True

    def load_state_dict(self, state_dict):
        """A stock AdamW state_dict (or one of this class: the same thing) — loaded by torch, then taken back into the
        flat buffers."""
        super().load_state_dict(state_dict)
        if _adoptable_state(self):
            if self._sink_keys:
                from . import functional as F
                F.unregister_grad_sinks(self._sink_keys)
            self._adopt()
        else:
            self._uniform = False


def _raw_stock_step():
    """torch.optim.AdamW's own step WITHOUT torch's hook-running wrapper (AdoptedAdamW.step carries that wrapper itself)."""
    f = torch.optim.AdamW.step
    return f.__wrapped__ if getattr(f, "hooked", False) else f


def _adoptable_state(opt):
    ps = opt.param_groups[0]["params"] if len(opt.param_groups) == 1 else []
    st = [opt.state.get(p) for p in ps]
    if not ps or not any(st):
        return bool(ps)
    return all(s and set(s) >= {"step", "exp_avg", "exp_avg_sq"} for s in st) and len({float(s["step"]) for s in st}) == 1


def adopt_stock_adamw(optimizer, allow_cpu=False):
    """Turn a stock `torch.optim.AdamW` with stock options over ONE parameter group of fp32 GPU parameters into the fused
    flat form, in place (see above); anything else is returned untouched.  Idempotent and cheap: the reference-API loops
    call it at the top of every epoch (src/training/loop.py, loop_demo.py).  ECG_HIP_ADOPT_ADAMW=0 switches it off."""
    import os
    if isinstance(optimizer, (AdoptedAdamW, FlatAdamW)) or os.environ.get("ECG_HIP_ADOPT_ADAMW", "1") == "0":
        return optimizer
    if not _adoptable(optimizer, allow_cpu):
        return optimizer
    with torch.no_grad():
        optimizer.__class__ = AdoptedAdamW
        optimizer._adopt()
    return optimizer
