#!/usr/bin/env python3
"""Execute the RCCL code path of the data-parallel train step on a ONE-GPU box.

A multi-GPU node is not available to the builder (the driver runs the 1/2/4/8 curve at round end), and RCCL refuses two
ranks on one device, so the multi-rank tests exchange over gloo.  What CAN run here is a one-rank RCCL group: it
initialises the communicator (`ncclCommInitRank`), launches real all-reduce / broadcast / barrier calls on RCCL's
stream, and returns the same `Work` handles, stream-ordering and allocator behaviour the N-rank run will see — the sum
over one rank is the identity, so every result can be checked bit for bit against the non-distributed step:

  * `FlatAdamW(exchange_single_rank=True)`: hooked, bucketed, async all-reduces issued from backward + waits in step();
  * `FlatGradDDP(exchange_single_rank=True)` + the stock `torch.optim.AdamW` (what scripts/03 constructs);
  * `FlatAdamW.no_sync()` accumulation;
and the per-call cost of the exchange (issue → wait on the compute stream) is measured for the flat gradient
(719 397 floats) — the fixed RCCL cost a step pays before any link traffic.  Prints one JSON line.

    python tools/rccl_selftest.py [--steps 30]
"""
import argparse
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ptbxl-multimodal_amd"))
# The BatchNorm backward takes its one-launch form unless collectives can run under backward (ecg_hip.functional:
# declare_backward_collectives): the hooked exchange declares its parameters "busy" (two passes), the single all-reduce and
# FlatGradDDP declare them "quiet" (one launch).  The two forms associate their partial sums differently, so each
# bit-for-bit comparison below runs its reference step (no exchange) in the form the exchanged step takes.

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--port", type=int, default=29631)
    ap.add_argument("--sweep", action="store_true", help="also time other early-bucket sizes")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ecg_hip import _lib, ddp
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_cnn import ECGCNN
    from src.training.loop import train_one_epoch
    from src.utils.seed import set_seed

    if not torch.cuda.is_available():
        raise SystemExit("rccl_selftest: needs a GPU")
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.port))
    os.environ.pop("ECG_HIP_REHEARSE_ON_ONE_GPU", None)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _lib.load()
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "torch": torch.__version__,
           "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())}

    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    out["ranks_seen_by_allreduce"] = int(ones.item())

    B, steps = args.batch, args.steps
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 12, 1000, generator=g).to(dev)
    y = (torch.rand(B, 5, generator=g) < 0.3).float().to(dev)

    def fresh():
        set_seed(42)
        return ECGCNN(num_labels=5).to(dev)

    def flat_state(model):
        return torch.cat([t.detach().reshape(-1).float() for t in list(model.parameters()) + list(model.buffers())])

    class Batches(list):                      # what the loop needs of a DataLoader: iteration and len(.dataset)
        @property
        def dataset(self):
            return range(sum(b[0].shape[0] for b in self))

    def run(model, opt, wrapped=None, n=3):
        losses = [train_one_epoch(wrapped or model, Batches([(x, y)]), opt, dev) for _ in range(n)]
        torch.cuda.synchronize()
        return losses

    import ecg_hip.functional as hipF

    def one_launch(model):            # the form ConvBlockFn.backward will pick for this model's blocks (per call, per parameter)
        return bool(hipF.bn_backward_one_launch_allowed(model.backbone[1].net[0].weight.data_ptr()))

    # references without an exchange, in both forms of the BatchNorm backward
    ref = fresh()
    out["bn_backward_one_launch"] = {"undeclared_single_rank": one_launch(ref)}
    ref_losses = run(ref, FlatAdamW(ref.parameters(), lr=1e-3, weight_decay=1e-4))
    hipF.set_bn_backward_one_launch(False)
    ref2p = fresh()
    ref2p_losses = run(ref2p, FlatAdamW(ref2p.parameters(), lr=1e-3, weight_decay=1e-4))
    hipF.set_bn_backward_one_launch(True)

    # 1. FlatAdamW: hooked two-bucket exchange over RCCL vs no exchange at all (two-pass BatchNorm backward in both)
    m1 = fresh()
    ddp.broadcast_module_state(m1, 0)
    o1 = FlatAdamW(m1.parameters(), lr=1e-3, weight_decay=1e-4, exchange_single_rank=True)
    out["bn_backward_one_launch"]["hooked_exchange"] = one_launch(m1)
    l1 = run(m1, o1)
    out["flat_adamw_hooked_exchange"] = {
        "overlap_hooks_active": bool(o1._overlap),
        "bit_identical_to_unexchanged_step": bool(torch.equal(flat_state(ref2p), flat_state(m1))) and l1 == ref2p_losses}

    # 2. the exchange in one piece (overlap off): nothing communicates under backward, the one-launch form stays on
    m2 = fresh()
    o2 = FlatAdamW(m2.parameters(), lr=1e-3, weight_decay=1e-4, exchange_single_rank=True, overlap=False)
    out["bn_backward_one_launch"]["single_allreduce"] = one_launch(m2)
    l2 = run(m2, o2)
    out["flat_adamw_single_allreduce"] = {
        "bit_identical_to_unexchanged_step": bool(torch.equal(flat_state(ref), flat_state(m2))) and l2 == ref_losses}
    o1.set_overlap(False)
    out["bn_backward_one_launch"]["hooked_then_set_overlap_false"] = one_launch(m1)
    o1.set_overlap(True)

    # 3. stock AdamW behind FlatGradDDP (scripts/03's optimizer)
    ref3 = fresh()
    ref3_losses = run(ref3, torch.optim.AdamW(ref3.parameters(), lr=1e-3, weight_decay=1e-4))
    m3 = fresh()
    w3 = ddp.FlatGradDDP(m3, exchange_single_rank=True)
    out["bn_backward_one_launch"]["flat_grad_ddp"] = one_launch(m3)
    l3 = run(m3, torch.optim.AdamW(m3.parameters(), lr=1e-3, weight_decay=1e-4), wrapped=w3)
    out["flat_grad_ddp_stock_adamw"] = {
        "bit_identical_to_unexchanged_step": bool(torch.equal(flat_state(ref3), flat_state(m3))) and l3 == ref3_losses}

    # 4. accumulation: two backward passes, the first under no_sync() (a hooked optimizer: two-pass form on both sides)

    def two_pass(model, opt, exchanged):
        opt.zero_grad(set_to_none=True)
        half = B // 2
        ctx = opt.no_sync() if exchanged else contextlib.nullcontext()
        with ctx:
            hipF.binary_cross_entropy_with_logits(model(x[:half]), y[:half]).backward()
        hipF.binary_cross_entropy_with_logits(model(x[half:]), y[half:]).backward()
        opt.step()
        torch.cuda.synchronize()

    ra, ma = fresh(), fresh()
    hipF.set_bn_backward_one_launch(False)
    two_pass(ra, FlatAdamW(ra.parameters(), lr=1e-3), False)
    hipF.set_bn_backward_one_launch(True)
    two_pass(ma, FlatAdamW(ma.parameters(), lr=1e-3, exchange_single_rank=True), True)
    out["no_sync_accumulation"] = {"bit_identical_to_unexchanged_step": bool(torch.equal(flat_state(ra), flat_state(ma)))}

    # 5. what one exchange costs on the compute stream (no link traffic at one rank: the fixed RCCL cost)
    flat = o1.flat_grad
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for _ in range(5):
        dist.all_reduce(flat, async_op=True).wait()
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        w_late = dist.all_reduce(flat[o1._split:], async_op=True)
        w_early = dist.all_reduce(flat[:o1._split], async_op=True)
        w_early.wait()
        w_late.wait()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    out["allreduce_two_buckets_ms"] = {"median": round(ms[len(ms) // 2], 4), "min": round(ms[0], 4),
                                       "max": round(ms[-1], 4), "floats": int(flat.numel()), "n": steps}

    # 6. step time with and without the exchange (fresh models, same batch; one epoch of `steps` steps, one sync)
    def timed(exchange, **kw):
        model = fresh()
        opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, exchange_single_rank=exchange, **kw)
        train_one_epoch(model, Batches([(x, y)] * 15), opt, dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(3):
            a.record()
            train_one_epoch(model, Batches([(x, y)] * steps), opt, dev)
            b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b) / steps
            best = t if best is None else min(best, t)
        return round(best, 4)

    out["step_ms"] = {"no_exchange": timed(False), "one_rank_rccl_exchange": timed(True),
                      "one_rank_rccl_single_allreduce": timed(True, overlap=False)}
    if args.sweep:
        out["step_ms"].update({f"early_params_{n}": timed(True, early_params=n) for n in (2, 4, 12, 16)})
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()
    ok = (out["ranks_seen_by_allreduce"] == 1 and out["flat_adamw_hooked_exchange"]["overlap_hooks_active"]
          and all(out[k]["bit_identical_to_unexchanged_step"] for k in
                  ("flat_adamw_hooked_exchange", "flat_adamw_single_allreduce", "flat_grad_ddp_stock_adamw",
                   "no_sync_accumulation")))
    out["ok"] = bool(ok)
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
