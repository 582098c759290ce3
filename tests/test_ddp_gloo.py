"""Multi-process (gloo, CPU) tests of the data-parallel path: one all-reduce of the flat
gradient per step, per-rank BatchNorm statistics, rank-0 broadcast of the initial replica.
The model here is the CPU oracle (stock torch) — the DDP helpers are model-agnostic; the HIP
kernels are covered by the -m gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import check_put, golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from ecg_hip import ddp
    r, w, _ = ddp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    return ddp


def _worker_g5(rank, world, port, name, out_dir):
    """G5: 8 shards of 4 windows, per-shard BN, averaged gradients, AdamW step on rank-0 buffers."""
    ddp = _setup(rank, world, port)
    from oracle import ref_models as R
    ctor, C, demo, lr = {"cnn5": (lambda: R.RefECGCNN(num_labels=5), 5, False, 1.5e-3),
                         "mm": (lambda: R.RefECGMultimodal(), 5, True, 1e-4)}[name]
    R.seed_all(42 + rank)                 # deliberately different replicas: rank 0 must win
    model = ctor().train()
    wrapped = ddp.FlatGradDDP(model)      # broadcasts rank 0's parameters + buffers
    R.seed_all(42)
    ref0 = ctor()
    if rank != 0:
        for a, b in zip(model.state_dict().values(), ref0.state_dict().values()):
            assert torch.equal(a, b), "replica was not initialised from rank 0"
    batch = R.synthetic_batch(32, 1000, C, demo=demo)
    shard = ddp.shard_batch(batch, rank, world)
    assert shard[0].shape[0] == 4
    opt = R.make_adamw(model, lr, 1e-4)
    opt.zero_grad()
    out = wrapped(*shard[:-1])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, shard[-1])
    loss.backward()                       # hook: one all-reduce of the flat gradient, then 1/world
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    opt.step()
    if rank == 0:
        torch.save({"grads": grads, "sd": model.state_dict()}, os.path.join(out_dir, f"{name}.pt"))
    # every rank holds the same averaged gradient
    flat = torch.cat([g.reshape(-1) for g in grads.values()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    for g in gathered:
        assert torch.equal(g, gathered[0])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["cnn5", "mm"])
def test_g5_ddp_semantics_world8(tmp_path, name):
    world, port = 8, _free_port()
    mp.spawn(_worker_g5, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(tmp_path, f"{name}.pt"))
    g = golden("g5_ddp")
    for k, v in got["grads"].items():
        check_put(g, f"{name}_avg_grad_{k}", v, atol=2e-6)
    lr = float(g[f"{name}_lr"])
    for k, v in got["sd"].items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(g[f"{name}_post_sd_{k}"]) == 1
        else:
            # Adam normalisation amplifies ~1e-7 summation-order noise of near-zero gradients
            check_put(g, f"{name}_post_sd_{k}", v, atol=2.02 * lr if ".net.0.bias" in k else 2e-5)


def _worker_flat(rank, world, port, out_dir):
    """FlatAdamW.reduce_gradients (the optimizer-integrated exchange) + helpers, world 2."""
    ddp = _setup(rank, world, port)
    from ecg_hip.optim import FlatAdamW, flatten_tensors_
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    before = [p.detach().clone() for p in lin.parameters()]
    opt = FlatAdamW(lin.parameters(), lr=1e-3, weight_decay=1e-4, overlap=False)   # single bucket
    assert opt.world_size == 2 and opt.flat_param.numel() == sum(p.numel() for p in lin.parameters())
    for p, b in zip(lin.parameters(), before):
        assert torch.equal(p, b) and p.data_ptr() >= opt.flat_param.data_ptr()   # values kept, storage shared
    x = torch.full((4, 7), float(rank + 1))
    lin(x).sum().backward()
    local = torch.cat([p.grad.reshape(-1) for p in lin.parameters()]).clone()
    flat, scale = opt.reduce_gradients()
    assert scale == 0.5
    both = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    assert torch.allclose(flat, both[0] + both[1])                     # SUM; 1/world is applied in the kernel
    # the update on CPU parameters is stock torch math: one step == torch.optim.AdamW on the averaged gradient
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ref.load_state_dict({k: v.clone() for k, v in lin.state_dict().items()})
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=1e-4)
    off = 0
    for q in ref.parameters():
        q.grad = (0.5 * (both[0] + both[1]))[off:off + q.numel()].view_as(q).clone()
        off += q.numel()
    ropt.step()
    opt.step()
    for a, b in zip(lin.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=0, atol=1e-7)
    # hook-based wrapper: uneven shards are rejected, equal shards are contiguous
    with pytest.raises(ValueError):
        ddp.shard_batch((torch.zeros(5, 2),), rank, world)
    a, = ddp.shard_batch((torch.arange(8).view(8, 1),), rank, world)
    assert a.flatten().tolist() == list(range(rank * 4, rank * 4 + 4))
    dist.barrier()
    dist.destroy_process_group()


def _worker_overlap(rank, world, port, out_dir):
    """Bucketed exchange launched from backward hooks == single flat all-reduce (world 2)."""
    _setup(rank, world, port)
    from ecg_hip.optim import FlatAdamW
    from oracle import ref_models as R
    R.seed_all(42)
    model = R.RefECGMultimodal().train()
    opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)            # overlap on, early bucket = 8 tensors
    assert opt._overlap and opt._n_early == 8
    assert opt._split == sum(p.numel() for p in list(model.parameters())[:8])
    batch = R.synthetic_batch(8, 256, 5, gen_seed=100 + rank, demo=True)
    for it in range(2):                                                        # two steps: state resets correctly
        opt.zero_grad()
        loss = torch.nn.functional.binary_cross_entropy_with_logits(model(*batch[:-1]), batch[-1])
        loss.backward()
        assert opt._late_work is not None                                      # launched during backward
        local = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
        flat, scale = opt.reduce_gradients()
        assert scale == 0.5 and opt._late_work is None
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        assert torch.allclose(flat, both[0] + both[1], rtol=0, atol=0)
    # set_overlap(False): the hooks stay quiet, step() exchanges everything in one collective — same summed gradient;
    # and back again (what bench.py's per-node calibration does between steps)
    for mode in (False, True):
        assert opt.set_overlap(mode) is mode
        opt.zero_grad()
        torch.nn.functional.binary_cross_entropy_with_logits(model(*batch[:-1]), batch[-1]).backward()
        assert (opt._late_work is not None) is mode
        local = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
        flat, scale = opt.reduce_gradients()
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        assert scale == 0.5 and torch.equal(flat, both[0] + both[1])
    opt.zero_grad()
    torch.nn.functional.binary_cross_entropy_with_logits(model(*batch[:-1]), batch[-1]).backward()
    try:
        opt.set_overlap(False)                       # an exchange is in flight
        raise AssertionError("set_overlap accepted a switch with an all-reduce in flight")
    except RuntimeError as e:
        assert "in flight" in str(e)
    opt.reduce_gradients()
    # calibrate_overlap: both forms timed on this "node", the same decision on every rank
    def run_steps(n):
        for _ in range(n):
            opt.zero_grad()
            torch.nn.functional.binary_cross_entropy_with_logits(model(*batch[:-1]), batch[-1]).backward()
            opt.step()
    cal = opt.calibrate_overlap(run_steps, steps=2, warm=1)
    assert cal["mode"] in ("overlapped", "single") and set(cal["ms_per_step"]) == {"overlapped", "single"}
    assert opt._overlap_active is (cal["mode"] == "overlapped")
    flags = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(flags, torch.tensor([1.0 if cal["mode"] == "overlapped" else 0.0]))
    assert flags[0].item() == flags[1].item()
    # overlap=False gives the same flat gradient through one collective
    opt2 = FlatAdamW(model.parameters(), lr=1e-4, overlap=False)
    assert not opt2._overlap and opt2.set_overlap(True) is False
    assert opt2.calibrate_overlap(run_steps)["mode"] == "single"
    dist.barrier()
    dist.destroy_process_group()


def _worker_product_g5(rank, world, port, name, out_dir):
    """G5 through the PRODUCT modules (src.models on CPU tensors = stock torch layers) and the product
    optimizer: FlatAdamW(process_group) with the bucketed exchange issued from backward hooks."""
    ddp = _setup(rank, world, port)
    from ecg_hip.optim import FlatAdamW
    from oracle import ref_models as R
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    from src.training.loop import train_one_epoch
    from src.training.loop_demo import train_one_epoch_demo
    from src.utils.seed import set_seed
    ctor, C, demo, lr = {"cnn5": (lambda: ECGCNN(num_labels=5), 5, False, 1.5e-3),
                         "mm": (lambda: ECGMultimodal(), 5, True, 1e-4)}[name]
    set_seed(42 + rank)                   # deliberately different replicas: rank 0 must win
    model = ctor().train()
    ddp.broadcast_module_state(model, 0)
    opt = FlatAdamW(model.parameters(), lr=lr, weight_decay=1e-4)
    assert opt.world_size == world and opt._overlap
    batch = R.synthetic_batch(32, 1000, C, demo=demo)
    shard = ddp.shard_batch(batch, rank, world)

    class OneBatch:
        dataset = range(shard[0].shape[0])

        def __iter__(self):
            return iter([shard])

    (train_one_epoch_demo if demo else train_one_epoch)(model, OneBatch(), opt, "cpu")
    if rank == 0:
        torch.save({"sd": model.state_dict(), "flat_grad": opt.flat_grad.clone() / world}, os.path.join(out_dir, f"p_{name}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["cnn5", "mm"])
def test_g5_through_product_modules_and_flat_optimizer_world8(tmp_path, name):
    world, port = 8, _free_port()
    mp.spawn(_worker_product_g5, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(tmp_path, f"p_{name}.pt"))
    g = golden("g5_ddp")
    lr = float(g[f"{name}_lr"])
    for k, v in got["sd"].items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(g[f"{name}_post_sd_{k}"]) == 1
        else:
            check_put(g, f"{name}_post_sd_{k}", v, atol=2.02 * lr if ".net.0.bias" in k else 2e-5)


def _worker_accum(rank, world, port, out_dir):
    """Gradient accumulation: two backward passes per step.  Without no_sync() the second pass would add local
    gradients to an already rank-summed bucket — FlatAdamW refuses; with no_sync() the accumulated gradient is
    exchanged once."""
    _setup(rank, world, port)
    from ecg_hip.optim import FlatAdamW
    torch.manual_seed(0)
    lin = torch.nn.Sequential(*[torch.nn.Linear(6, 6) for _ in range(6)])          # 12 tensors: early bucket = 8
    opt = FlatAdamW(lin.parameters(), lr=1e-3, weight_decay=0.0)
    assert opt._overlap
    xs = [torch.full((3, 6), float(rank + 1 + i)) for i in range(2)]
    lin(xs[0]).sum().backward()
    with pytest.raises(RuntimeError, match="no_sync"):
        lin(xs[1]).sum().backward()
    opt.reduce_gradients()                                  # drain the in-flight work of the first pass
    opt.zero_grad()
    with opt.no_sync():
        lin(xs[0]).sum().backward()
        assert opt._late_work is None and opt._early_work is None
    lin(xs[1]).sum().backward()                             # exchanges g(x0) + g(x1)
    local = []
    for i in range(2):
        ref = torch.nn.Sequential(*[torch.nn.Linear(6, 6) for _ in range(6)])
        ref.load_state_dict(lin.state_dict())
        ref(xs[i]).sum().backward()
        local.append(torch.cat([p.grad.reshape(-1) for p in ref.parameters()]))
    mine = local[0] + local[1]
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    flat, scale = opt.reduce_gradients()
    assert scale == 0.5 and torch.allclose(flat, both[0] + both[1], rtol=1e-6, atol=1e-6)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulation_needs_no_sync_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker_accum, args=(world, port, str(tmp_path)), nprocs=world, join=True)


def test_bucketed_overlap_exchange_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker_overlap, args=(world, port, str(tmp_path)), nprocs=world, join=True)


def test_flat_optimizer_exchange_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker_flat, args=(world, port, str(tmp_path)), nprocs=world, join=True)


def test_single_process_is_a_noop():
    os.environ.pop("WORLD_SIZE", None)
    from ecg_hip import ddp
    assert ddp.init_distributed("gloo") == (0, 1, 0)
    m = torch.nn.Linear(3, 2)
    w = ddp.FlatGradDDP(m)
    assert w.world == 1 and w(torch.zeros(1, 3)).shape == (1, 2)
