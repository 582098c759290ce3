"""Data-parallel training of the ECG models: one process per GPU, each rank a full replica
(3 MB of parameters), the batch dimension sharded, and ONE collective per step — an
all-reduce (sum, then 1/world) of the flat fp32 gradient over RCCL/xGMI
(`torch.distributed` backend "nccl"; "gloo" in the CPU tests).

The reference has no distributed code at all; semantics follow
torch DistributedDataParallel: replicas start from rank 0's parameters and buffers,
gradients are averaged, BatchNorm statistics stay per-rank (no SyncBN in the reference),
and buffers are optionally re-broadcast from rank 0 before each forward.

Two ways to use it:
  * `FlatAdamW(model.parameters(), ..., process_group=pg)` — the exchange is folded into the
    optimizer step (gather -> all_reduce -> fused AdamW with grad_scale=1/world);
  * `FlatGradDDP(model)` + any stock optimizer — gradients are exchanged by a hook that fires
    when the last parameter gradient of the step has been accumulated.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


def init_distributed(backend=None, timeout_s=None):
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract).
    Returns (rank, world, local_rank); a no-op single process when WORLD_SIZE is unset or 1.
    `timeout_s` bounds the rendezvous and every later collective (torch's default is 10 minutes: a rank that died
    before the rendezvous would otherwise hold the others for longer than a bench lease lasts)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ECG_HIP_REHEARSE_ON_ONE_GPU") == "1":
        # development aid for a one-GPU box: every rank shares device 0 and the exchange runs over gloo
        # (RCCL refuses two ranks on one device) — exercises the multi-rank code path, not its speed
        backend, local = "gloo", 0
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_batch(tensors, rank, world):
    """Contiguous equal shards of the batch dimension (global batch = world * per-rank batch)."""
    out = []
    for t in tensors:
        n = t.shape[0]
        if n % world:
            raise ValueError(f"global batch {n} is not divisible by world size {world}")
        per = n // world
        out.append(t[rank * per:(rank + 1) * per])
    return tuple(out)


def broadcast_module_state(module, src=0, group=None):
    """Rank `src`'s parameters and buffers to every rank (as DDP does at construction)."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def allreduce_flat_gradients(params, group=None, world=None):
    """Average .grad of `params` across ranks with a single all-reduce of one flat buffer."""
    params = [p for p in params if p.requires_grad]
    world = world or dist.get_world_size(group)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, group=group)
    flat.mul_(1.0 / world)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return flat


class FlatGradDDP(nn.Module):
    """Wrap a model so that `loss.backward()` leaves rank-averaged gradients in `.grad`."""

    def __init__(self, module, process_group=None, broadcast_buffers=False, exchange_single_rank=False):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.broadcast_buffers = broadcast_buffers
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._params = [p for p in module.parameters() if p.requires_grad]
        self._pending = 0
        # exchange_single_rank: keep the hooks and the all-reduce for a ONE-rank group (identity; lets the RCCL path
        # run on a one-GPU box, tools/rccl_selftest.py)
        self._exchange = self.world > 1 or (bool(exchange_single_rank) and dist.is_initialized())
        if self._exchange:
            broadcast_module_state(module, 0, process_group)
            for p in self._params:
                p.register_post_accumulate_grad_hook(self._on_grad)
            # the one all-reduce is issued from the hook of the LAST gradient, i.e. behind every backward kernel on the
            # stream: nothing communicates while the BatchNorm backward kernels run ("quiet", functional.py)
            if self._params and self._params[0].is_cuda:
                import weakref
                from . import functional as F
                # (an optimizer that re-homes the parameters later and states its own exchange mode overrides this, as it
                # should: a hooked FlatAdamW exchange IS busy under backward; a stock / adopted optimizer says nothing)
                token = ("FlatGradDDP", id(self))
                F.declare_backward_collectives(self._params, False, owner=token)
                # withdraw the statement when the wrapper goes away (by object id + owner token: a later parameter that
                # reuses an id was declared by someone else and keeps its statement)
                weakref.finalize(self, F.declare_backward_collectives, [id(p) for p in self._params], None, token)

    def _on_grad(self, _param):
        self._pending += 1
        if self._pending == len(self._params):       # last gradient of this backward pass
            self._pending = 0
            allreduce_flat_gradients(self._params, self.process_group, self.world)

    def forward(self, *args, **kwargs):
        self._pending = 0
        if self._exchange and self.broadcast_buffers and self.module.training:
            for b in self.module.buffers():
                dist.broadcast(b.data, src=0, group=self.process_group)
        return self.module(*args, **kwargs)
