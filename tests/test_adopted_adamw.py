"""The stock optimizer of the reference's scripts (scripts/03_train_ecg_baseline.py:133, 04:158-162, 05:130:
`AdamW(model.parameters(), lr=lr, weight_decay=wd)`) adopted into the fused flat step (ecg_hip.optim.adopt_stock_adamw):
same object, stock state_dict, stock semantics wherever the one launch does not apply."""
import copy

import numpy as np
import pytest
import torch

DEV = "cuda"
gpu = pytest.mark.gpu


def _twin_models():
    from src.models.ecg_cnn import ECGCNN
    from src.utils.seed import set_seed
    set_seed(42)
    a = ECGCNN(num_labels=5)
    b = copy.deepcopy(a)
    return a, b


def _fake_grads(models, seed):
    g = torch.Generator().manual_seed(seed)
    grads = [torch.randn(p.shape, generator=g) * 1e-2 for p in models[0].parameters()]
    for m in models:
        for p, gr in zip(m.parameters(), grads):
            p.grad = gr.clone().to(p.device)


def _assert_same(ma, mb, exact=True, atol=0.0):
    for (k, p), q in zip(ma.named_parameters(), mb.parameters()):
        if exact:
            assert torch.equal(p.detach().cpu(), q.detach().cpu()), k
        else:
            assert (p.detach().cpu() - q.detach().cpu()).abs().max().item() <= atol, k


def test_adopted_adamw_is_the_same_object_with_a_stock_state_dict_and_stock_arithmetic():
    from ecg_hip.optim import AdoptedAdamW, adopt_stock_adamw
    ma, mb = _twin_models()
    oa = torch.optim.AdamW(ma.parameters(), lr=1e-3, weight_decay=1e-4)
    ob = torch.optim.AdamW(mb.parameters(), lr=1e-3, weight_decay=1e-4)
    assert adopt_stock_adamw(oa) is oa and type(oa) is torch.optim.AdamW          # CPU parameters: untouched by default
    got = adopt_stock_adamw(oa, allow_cpu=True)
    assert got is oa and isinstance(oa, torch.optim.AdamW) and isinstance(oa, AdoptedAdamW)
    assert adopt_stock_adamw(oa, allow_cpu=True) is oa                             # idempotent
    ps = list(ma.parameters())
    base = oa.flat_param.data_ptr()
    assert ps[0].data_ptr() == base and all(base <= p.data_ptr() < base + 4 * oa.flat_param.numel() for p in ps)
    for s in range(3):
        _fake_grads((ma, mb), s)
        oa.step(), ob.step()
    _assert_same(ma, mb)                      # the flat CPU step is torch's single-tensor formula, op for op
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"] == sb["param_groups"] and sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert sa["state"][k].keys() == sb["state"][k].keys() == {"step", "exp_avg", "exp_avg_sq"}
        assert float(sa["state"][k]["step"]) == 3.0 and sa["state"][k]["step"].device.type == "cpu"
        assert torch.equal(sa["state"][k]["exp_avg"], sb["state"][k]["exp_avg"])
        assert torch.equal(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"])


def test_checkpoint_of_an_adopted_optimizer_continues_in_stock_torch_and_back():
    """adopted -> state_dict -> a FRESH stock torch.optim.AdamW continues bit-compatibly; and a stock checkpoint loads into an
    adopted optimizer, which then keeps stepping through the flat buffers."""
    from ecg_hip.optim import adopt_stock_adamw
    ma, mb = _twin_models()
    oa = adopt_stock_adamw(torch.optim.AdamW(ma.parameters(), lr=2e-3, weight_decay=1e-2), allow_cpu=True)
    for s in range(2):
        _fake_grads((ma,), s)
        oa.step()
    ckpt = copy.deepcopy({"model": ma.state_dict(), "opt": oa.state_dict()})
    mb.load_state_dict(ckpt["model"])
    ob = torch.optim.AdamW(mb.parameters(), lr=1.0)                  # options come from the checkpoint
    ob.load_state_dict(ckpt["opt"])
    assert type(ob) is torch.optim.AdamW and ob.param_groups[0]["lr"] == 2e-3
    for s in range(2, 5):
        _fake_grads((ma, mb), s)
        oa.step(), ob.step()
    _assert_same(ma, mb)
    # ... and back: the stock optimizer's checkpoint into a third, adopted one
    mc, _ = _twin_models()
    oc = adopt_stock_adamw(torch.optim.AdamW(mc.parameters(), lr=1.0), allow_cpu=True)
    mc.load_state_dict(mb.state_dict())
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))
    assert oc._uniform and oc._step_count == 5
    assert oc.state[next(iter(mc.parameters()))]["exp_avg"].data_ptr() == oc.flat_m.data_ptr()      # back in the flat buffers
    for s in range(5, 7):
        _fake_grads((mb, mc), s)
        ob.step(), oc.step()
    _assert_same(mb, mc)


def test_adopted_adamw_keeps_stock_semantics_where_the_one_launch_does_not_apply():
    from ecg_hip.optim import AdoptedAdamW, adopt_stock_adamw
    ma, mb = _twin_models()
    oa = adopt_stock_adamw(torch.optim.AdamW(ma.parameters(), lr=1e-3, weight_decay=1e-2), allow_cpu=True)
    ob = torch.optim.AdamW(mb.parameters(), lr=1e-3, weight_decay=1e-2)
    seen = []
    oa.register_step_post_hook(lambda opt, args, kwargs: seen.append(1))
    _fake_grads((ma, mb), 0)
    oa.step(), ob.step()
    assert seen == [1]                                                 # step hooks run
    # a parameter without a gradient: stock AdamW skips it entirely (no decay, no moment update, no step count)
    _fake_grads((ma, mb), 1)
    for m in (ma, mb):
        m.head.weight.grad = None
    oa.step(), ob.step()
    _assert_same(ma, mb)
    assert not oa._uniform and float(oa.state[ma.head.weight]["step"]) == 1.0 and float(oa.state[ma.head.bias]["step"]) == 2.0
    _fake_grads((ma, mb), 2)
    oa.step(), ob.step()                                               # step counts differ per tensor now: torch's own step
    _assert_same(ma, mb)
    # an LR scheduler built AFTER adoption drives param_groups as ever; one built BEFORE has bound the stock step: not adopted
    mc, md = _twin_models()
    oc = adopt_stock_adamw(torch.optim.AdamW(mc.parameters(), lr=1e-3), allow_cpu=True)
    od = torch.optim.AdamW(md.parameters(), lr=1e-3)
    sc, sd = (torch.optim.lr_scheduler.StepLR(o, step_size=1, gamma=0.5) for o in (oc, od))
    for s in range(3):
        _fake_grads((mc, md), s)
        oc.step(), od.step(), sc.step(), sd.step()
    assert oc.param_groups[0]["lr"] == od.param_groups[0]["lr"] == 1.25e-4
    _assert_same(mc, md)
    assert not isinstance(adopt_stock_adamw(od, allow_cpu=True), AdoptedAdamW)
    # options the fused launch does not implement, several groups, another optimizer class: left alone
    me, _ = _twin_models()
    for o in (torch.optim.AdamW(me.parameters(), amsgrad=True), torch.optim.AdamW(me.parameters(), maximize=True),
              torch.optim.AdamW([{"params": me.backbone.parameters()}, {"params": me.head.parameters(), "lr": 1e-2}]),
              torch.optim.Adam(me.parameters()), torch.optim.SGD(me.parameters(), lr=0.1)):
        assert adopt_stock_adamw(o, allow_cpu=True) is o and not isinstance(o, AdoptedAdamW)
    # parameters re-pointed after adoption (model.to(...), p.data = ...): taken back into a fresh flat buffer at the next step
    mf, mg = _twin_models()
    of = adopt_stock_adamw(torch.optim.AdamW(mf.parameters(), lr=1e-3), allow_cpu=True)
    og = torch.optim.AdamW(mg.parameters(), lr=1e-3)
    _fake_grads((mf, mg), 0)
    of.step(), og.step()
    mf.proj.weight.data = mf.proj.weight.data.clone()
    _fake_grads((mf, mg), 1)
    of.step(), og.step()
    _assert_same(mf, mg)
    assert mf.proj.weight.data_ptr() != 0 and of._uniform and of._step_count == 2
    base = of.flat_param.data_ptr()
    assert base <= mf.proj.weight.data_ptr() < base + 4 * of.flat_param.numel()


def test_model_checkpoint_after_adoption_is_an_ordinary_checkpoint():
    """scripts/03:167, 04:207, 05:158 save `model.state_dict()` after training: with the parameters re-homed as views of one flat
    buffer the file must stay an ordinary checkpoint — no larger than the tensors it holds (torch.save stores the shared
    storage once), loadable with strict=True into a fresh model, bit for bit."""
    import io
    from ecg_hip.optim import adopt_stock_adamw
    from src.models.ecg_cnn import ECGCNN
    ma, _ = _twin_models()
    oa = adopt_stock_adamw(torch.optim.AdamW(ma.parameters(), lr=1e-3), allow_cpu=True)
    _fake_grads((ma,), 0)
    oa.step()
    buf = io.BytesIO()
    torch.save({"model_state": ma.state_dict(), "classes": ["a"]}, buf)
    nbytes = sum(t.numel() * t.element_size() for t in ma.state_dict().values())
    assert buf.tell() < 1.1 * nbytes + 65536, (buf.tell(), nbytes)
    buf.seek(0)
    ck = torch.load(buf, map_location="cpu")
    mb = ECGCNN(num_labels=5)
    mb.load_state_dict(ck["model_state"], strict=True)
    for (k, a), b in zip(ma.state_dict().items(), mb.state_dict().values()):
        assert torch.equal(a, b), k
    # ... and a copy of the model (copy.deepcopy, as the tests and EMA helpers do) does not alias the optimizer's buffers
    mc = copy.deepcopy(ma)
    _fake_grads((ma,), 1)
    oa.step()
    assert not torch.equal(next(iter(mc.parameters())), next(iter(ma.parameters())))


@gpu
def test_adopted_stock_adamw_through_the_loops_equals_flat_adamw_and_checkpoints_into_stock_torch():
    """On the GPU the loops adopt the scripts' optimizer: same kernel, same flat layout as FlatAdamW -> the same bits; its
    checkpoint loads into a fresh stock torch.optim.AdamW on the CPU and the next updates agree to fp32 rounding."""
    from ecg_hip.optim import AdoptedAdamW, FlatAdamW
    from oracle import ref_models as R
    from src.models.ecg_cnn import ECGCNN
    from src.training.loop import train_one_epoch
    from src.utils.seed import set_seed

    class DS(torch.utils.data.Dataset):
        def __init__(self, *t):
            self.t = t

        def __len__(self):
            return self.t[0].shape[0]

        def __getitem__(self, i):
            return tuple(a[i] for a in self.t)

    batch = R.synthetic_batch(16, 1000, 5, demo=False)
    loader = torch.utils.data.DataLoader(DS(*batch), batch_size=8, shuffle=False)
    models, opts = [], []
    for Opt in (torch.optim.AdamW, FlatAdamW):
        set_seed(42)
        m = ECGCNN(num_labels=5).to(DEV)
        o = Opt(m.parameters(), lr=1e-3, weight_decay=1e-4)
        losses = [train_one_epoch(m, loader, o, DEV) for _ in range(2)]
        models.append(m), opts.append((o, losses))
    assert isinstance(opts[0][0], AdoptedAdamW) and isinstance(opts[0][0], torch.optim.AdamW)
    assert opts[0][1] == opts[1][1]
    _assert_same(models[0], models[1])
    for (k, a), b in zip(models[0].state_dict().items(), models[1].state_dict().values()):
        assert torch.equal(a, b), k
    # gradients of the last step sit in the adopted optimizer's flat gradient (written there by the backward kernels)
    o = opts[0][0]
    lo, hi = o.flat_grad.data_ptr(), o.flat_grad.data_ptr() + 4 * o.flat_grad.numel()
    assert all(lo <= p.grad.data_ptr() < hi for p in models[0].parameters())
    sd = o.state_dict()
    assert all(float(s["step"]) == 4.0 and s["step"].device.type == "cpu" for s in sd["state"].values())
    # checkpoint -> stock torch on the CPU
    set_seed(42)
    mc = ECGCNN(num_labels=5)
    mc.load_state_dict({k: v.cpu() for k, v in models[0].state_dict().items()})
    oc = torch.optim.AdamW(mc.parameters(), lr=1.0)
    oc.load_state_dict(copy.deepcopy(sd))
    assert type(oc) is torch.optim.AdamW
    for p, q in zip(models[0].parameters(), mc.parameters()):
        q.grad = p.grad.detach().cpu().clone()
    before = [p.detach().cpu().clone() for p in models[0].parameters()]
    o.step(), oc.step()
    for (k, p), q, b in zip(models[0].named_parameters(), mc.parameters(), before):
        moved = (p.detach().cpu() - b).abs().max().item()
        assert moved > 0 and (p.detach().cpu() - q.detach()).abs().max().item() <= 1e-6 * max(1.0, q.abs().max().item()) + 2e-7, k
