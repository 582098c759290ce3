"""GPU parity of the whole drop-in path (src.models + src.training on the HIP kernels) against
the golden fixtures captured from the reference and against the CPU oracle.
Bar (north_star): logits within 1e-4 in fp32; integer outputs (y_pred, num_batches_tracked) exact."""
import copy

import numpy as np
import pytest
import torch

from util import check_put, golden, paired, sd_from_npz
from oracle import ref_models as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ctors():
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    return {"cnn5": (lambda: ECGCNN(num_labels=5), lambda: R.RefECGCNN(num_labels=5), 5, False),
            "cnn1": (lambda: ECGCNN(num_labels=1), lambda: R.RefECGCNN(num_labels=1), 1, False),
            "mm": (lambda: ECGMultimodal(), lambda: R.RefECGMultimodal(), 5, True)}


class _DS(torch.utils.data.Dataset):
    def __init__(self, *t):
        self.t = t

    def __len__(self):
        return self.t[0].shape[0]

    def __getitem__(self, i):
        return tuple(a[i] for a in self.t)


def test_g3_eval_known_answer():
    """Committed checkpoints x demo windows (12x5000): logits vs the reference CPU re-run (1e-4),
    committed CSV probabilities (5e-4) and thresholded predictions (exact away from 0.5)."""
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    g = golden("g3_eval_known_answer")
    x, demo = torch.from_numpy(g["ecg"]).to(DEV), torch.from_numpy(g["demo"]).to(DEV)
    for name, ctor in [("baseline", lambda: ECGCNN(num_labels=5)), ("multimodal", lambda: ECGMultimodal()),
                       ("af", lambda: ECGCNN(num_labels=1))]:
        m = ctor()
        m.load_state_dict(sd_from_npz(golden("g3_ckpt_" + name)), strict=True)
        m.to(DEV).eval()
        with torch.no_grad():
            logits = m(x, demo) if name == "multimodal" else m(x)
        np.testing.assert_allclose(logits.cpu().numpy(), g[name + "_logits"], atol=1e-4, err_msg=name)
        from ecg_hip import functional as hipF
        prob = hipF.sigmoid(logits).cpu().numpy()
        np.testing.assert_allclose(prob, g[name + "_csv_prob"], atol=5e-4)
        far = np.abs(g[name + "_csv_prob"] - 0.5) > 5e-4
        assert np.array_equal((prob >= 0.5).astype(np.int64)[far], g[name + "_csv_pred"][far])
        # BN counters are untouched by eval forwards (int64, exact)
        key = [k for k in m.state_dict() if k.endswith("num_batches_tracked")][0]
        assert int(m.state_dict()[key].item()) == int(golden("g3_ckpt_" + name)[key])


@pytest.mark.parametrize("B", [4, 32])
@pytest.mark.parametrize("name", ["cnn5", "cnn1", "mm"])
@pytest.mark.parametrize("optim", ["torch", "flat"])
def test_g4_train_steps_through_reference_loop_api(name, B, optim):
    """3 AdamW steps through src.training.{loop,loop_demo} exactly as scripts/03-05 drive them."""
    from ecg_hip.optim import FlatAdamW
    from src.training.loop import train_one_epoch
    from src.training.loop_demo import train_one_epoch_demo
    from src.utils.seed import set_seed
    g = golden("g4_train_step")
    ctor, _, C, demo = _ctors()[name]
    p = f"{name}_B{B}_"
    lr = float(g[p + "lr"])
    set_seed(42)
    model = ctor().to(DEV)
    batch = R.synthetic_batch(B, 1000, C, demo=demo)
    if B == 4 and name == "cnn5":
        np.testing.assert_array_equal(batch[0].numpy(), g["x_B4"])
    loader = torch.utils.data.DataLoader(_DS(*batch), batch_size=B, shuffle=False)
    Opt = FlatAdamW if optim == "flat" else torch.optim.AdamW
    opt = Opt(model.parameters(), lr=lr, weight_decay=1e-4)
    fn = train_one_epoch_demo if demo else train_one_epoch
    # first forward, train mode, on a copy so BN buffers of `model` are not advanced
    m2 = copy.deepcopy(model).train()
    with torch.no_grad():
        out = m2(*[t.to(DEV) for t in batch[:-1]])
    np.testing.assert_allclose(out.cpu().numpy(), g[p + "logits0"], atol=1e-4)
    for step in (1, 2, 3):
        ep = fn(model, loader, opt, DEV)
        # step 1 is computed from identical parameters; later steps inherit the Adam sensitivity below
        assert abs(ep - float(g[p + f"epoch_loss{step}"])) < (5e-6 if step == 1 else 2e-4), (step, ep)
        if step in (1, 3):
            q = p + f"s{step}_"
            sd = model.state_dict()
            for k, v in sd.items():
                if k.endswith("num_batches_tracked"):
                    assert int(v.item()) == int(g[q + "sd_" + k]) == step     # int64, exact
                    continue
                got, want = paired(g, q + "sd_" + k, v)
                diff = np.abs(got - want)
                if "running_" in k and step == 1:
                    assert diff.max() <= 1e-5, (k, diff.max())        # buffers after one forward: tight
                    continue
                # AdamW normalises every gradient by its own magnitude (first step = lr*g/(|g|+eps)),
                # so an element whose gradient is at rounding-noise level (conv biases under
                # train-mode BN: ~1e-9; weights with |g| ~ eps = 1e-8) lands anywhere within +-lr per
                # step of the reference, and those few elements then perturb later steps.  State
                # parity is therefore: hard bound 2*lr*step on EVERY element, and >= 97 % of the
                # elements within 0.3*lr*step; gradient parity (below) carries the tight check.
                assert diff.max() <= 2.02 * lr * step + 1e-6, (k, step, diff.max())
                if ".net.0.bias" not in k and "running_" not in k:
                    bad = int((diff > 0.3 * lr * step).sum())
                    assert bad <= max(2, 0.03 * diff.size), (k, step, bad, diff.size)
            if step == 1:      # gradients of the first step (computed from identical parameters)
                for k, prm in model.named_parameters():
                    tol = 1e-6 if ".net.0.bias" in k else 1e-4
                    check_put(g, q + "grad_" + k, prm.grad, atol=tol)


def test_g6_gradcam_hooks_and_demo_gradient():
    """Hooks on the last Conv1d (forward + full backward) see the same tensors as in the
    reference; CAM and d logit / d x_demo match (reference scripts/00_demo_inference.py:36-43,
    scripts/12:78-97)."""
    import torch.nn.functional as F
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    g = golden("g6_hooks")
    ga = golden("g3_eval_known_answer")
    x = torch.from_numpy(ga["ecg"][:1]).to(DEV)
    for name, ctor in [("baseline", lambda: ECGCNN(num_labels=5)), ("multimodal", lambda: ECGMultimodal())]:
        m = ctor()
        m.load_state_dict(sd_from_npz(golden("g3_ckpt_" + name)))
        m.to(DEV).eval()
        convs = [c for c in m.modules() if isinstance(c, torch.nn.Conv1d)]
        last = convs[-1]
        store = {}
        h1 = last.register_forward_hook(lambda mod, i, o: store.__setitem__("act", o.detach().clone()))
        h2 = last.register_full_backward_hook(lambda mod, gi, go: store.__setitem__("grad", go[0].detach().clone()))
        demo = torch.from_numpy(ga["demo"][:1]).to(DEV).requires_grad_(True)
        m.zero_grad()
        logits = m(x, demo) if name == "multimodal" else m(x)
        logits[:, 0].sum().backward()
        h1.remove(), h2.remove()
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g[name + "_logits"], atol=1e-4)
        A, G = store["act"][0], store["grad"][0]
        check_put(g, name + "_act", A, atol=1e-4)
        check_put(g, name + "_grad", G, atol=1e-6)
        cam = torch.relu((G.mean(dim=1, keepdim=True) * A).sum(dim=0))
        np.testing.assert_allclose(cam.cpu().numpy(), g[name + "_cam"], atol=1e-5)
        up = F.interpolate(cam[None, None], size=5000, mode="linear", align_corners=False)[0, 0]
        np.testing.assert_allclose(up.cpu().numpy(), g[name + "_cam_up"], atol=1e-5)
        if name == "multimodal":
            np.testing.assert_allclose(demo.grad.cpu().numpy(), g["multimodal_dlogit_ddemo"], atol=1e-5)
        # the deprecated NON-full hook, registered exactly as src/interpretability/grad_cam_1d.py:30-36 does
        # (forward hook stores output.detach(), register_backward_hook stores grad_output[0].detach())
        legacy = {}
        m_l = ctor()                     # a fresh module: torch refuses to mix full and non-full hooks on one
        m_l.load_state_dict(sd_from_npz(golden("g3_ckpt_" + name)))
        m_l.to(DEV).eval()
        last_l = [c for c in m_l.modules() if isinstance(c, torch.nn.Conv1d)][-1]
        last_l.register_forward_hook(lambda mod, i, o: legacy.__setitem__("act", o.detach()))
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # torch deprecates the non-full hook; the reference uses it
            last_l.register_backward_hook(lambda mod, gi, go: legacy.__setitem__("grad", go[0].detach()))
            lg = m_l(x, demo.detach()) if name == "multimodal" else m_l(x)
            lg[:, 0].sum().backward()
        check_put(g, name + "_act", legacy["act"][0], atol=1e-4)
        check_put(g, name + "_grad", legacy["grad"][0], atol=1e-6)
        assert torch.equal(legacy["act"], store["act"]) and torch.equal(legacy["grad"], store["grad"])
        # and the fused path (no hooks) gives the same logits
        with torch.no_grad():
            l2 = m(x, demo.detach()) if name == "multimodal" else m(x)
        np.testing.assert_allclose(l2.cpu().numpy(), logits.detach().cpu().numpy(), atol=2e-5)


@pytest.mark.parametrize("name,B,T", [("cnn5", 256, 1000), ("mm", 256, 1000), ("cnn1", 64, 5000), ("cnn1", 256, 5000)])
def test_full_size_train_step_vs_cpu_oracle(name, B, T):
    """BASELINE.json sizes (B=256, 12x1000 = configs[1]/[2]; the AF-binary 12x5000 window of configs[4] at B=64 and
    at its full B=256): one fp32 train step vs the stock-torch CPU restatement — logits 1e-4, thresholded
    predictions exact, first-step gradients, BN buffers and counters."""
    from ecg_hip.optim import FlatAdamW
    from src.utils.seed import set_seed
    from ecg_hip import functional as hipF
    ctor, rctor, C, demo = _ctors()[name]
    set_seed(42)
    model = ctor().to(DEV).train()
    R.seed_all(42)
    ref = rctor().train()
    batch = R.synthetic_batch(B, T, C, demo=demo)
    opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
    ropt = R.make_adamw(ref, 1e-4, 1e-4)
    opt.zero_grad()
    logits = model(*[t.to(DEV) for t in batch[:-1]])
    loss = hipF.binary_cross_entropy_with_logits(logits, batch[-1].to(DEV))
    loss.backward()
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    opt.step()
    rlogits, rloss = R.train_step(ref, ropt, batch)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), rlogits.numpy(), atol=1e-4)
    assert abs(loss.item() - rloss) < 1e-5
    assert np.array_equal((logits.detach().cpu() >= 0).numpy(), (rlogits >= 0).numpy())     # y_pred, exact
    for k, prm in ref.named_parameters():
        tol = 1e-6 if ".net.0.bias" in k else 5e-5
        np.testing.assert_allclose(grads[k].numpy(), prm.grad.numpy(), atol=tol, err_msg=k)
    for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        if k.endswith("num_batches_tracked"):
            assert int(a.item()) == int(b.item()) == 1
        elif "running_" in k:
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1e-5, err_msg=k)


@pytest.mark.parametrize("name,optim,B,steps", [("cnn5", "flat", 32, 8), ("cnn5", "torch", 32, 8), ("mm", "flat", 32, 8),
                                                ("cnn5", "flat", 256, 3), ("mm", "flat", 256, 3)])
def test_trajectory_error_vs_float64_no_worse_than_the_cpu_fp32_path(name, optim, B, steps):
    """Multi-step trajectories cannot be pinned element by element (AdamW normalises noise-level gradients: two
    correct fp32 runs differ by up to lr per step on those elements — the bound test_g4 uses).  What CAN be pinned:
    against the SAME model trained in float64 (the exact trajectory, CPU oracle in double), the HIP path must not
    drift more than the reference's own CPU fp32 run does.  Eight AdamW steps at B=32, and three at the HEADLINE size
    (B=256, 12x1000: BASELINE.json configs[1] / configs[2]), lr 1e-3."""
    from ecg_hip.optim import FlatAdamW
    from src.utils.seed import set_seed
    from ecg_hip import functional as hipF
    ctor, rctor, C, demo = _ctors()[name]
    lr = 1e-3
    batch = R.synthetic_batch(B, 1000, C, demo=demo)
    set_seed(42)
    model = ctor().to(DEV).train()
    R.seed_all(42)
    ref32 = rctor().train()
    ref64 = copy.deepcopy(ref32).double()
    batch64 = tuple(t.double() for t in batch)
    opt = (FlatAdamW if optim == "flat" else torch.optim.AdamW)(model.parameters(), lr=lr, weight_decay=1e-4)
    o32, o64 = R.make_adamw(ref32, lr, 1e-4), R.make_adamw(ref64, lr, 1e-4)
    dbatch = [t.to(DEV) for t in batch]
    l_hip, l32, l64 = [], [], []
    for _ in range(steps):
        opt.zero_grad()
        loss = hipF.binary_cross_entropy_with_logits(model(*dbatch[:-1]), dbatch[-1])
        loss.backward()
        opt.step()
        l_hip.append(loss.item())
        l32.append(R.train_step(ref32, o32, batch)[1])
        l64.append(R.train_step(ref64, o64, batch64)[1])
    l_hip, l32, l64 = np.array(l_hip), np.array(l32), np.array(l64)
    # Same bars at every batch size.  (Round 3 had scaled them by sqrt(B/32) for the headline batch: the weight-gradient
    # kernels then accumulated one k-ordered fp32 fma chain of 64 x stages terms per slab, 500 ... 1 900 terms at B=256.
    # The kernels now sum each 64-term stage from zero and add it to a second accumulator — conv1d_mfma.hip, "TWO-LEVEL
    # ACCUMULATION" — and the unscaled bars hold; tools/wgrad_error.py prints the per-layer attribution.)
    chain = 1.0
    floor = 2e-6
    assert np.abs(l_hip - l64).max() <= max(3.0 * np.abs(l32 - l64).max(), floor), (l_hip - l64, l32 - l64)
    sd, sd32, sd64 = model.state_dict(), ref32.state_dict(), ref64.state_dict()
    tot_hip = tot_ref = n = 0.0
    for (k, a), b32, b64 in zip(sd.items(), sd32.values(), sd64.values()):
        if k.endswith("num_batches_tracked"):
            assert int(a.item()) == int(b64.item()) == steps
            continue
        e_hip = (a.detach().cpu().double() - b64).abs().numpy().ravel()
        e_ref = (b32.double() - b64).abs().numpy().ravel()
        tot_hip, tot_ref, n = tot_hip + e_hip.sum(), tot_ref + e_ref.sum(), n + e_hip.size
        # per tensor: the same hard bound for both runs, and the HIP run's mean drift within 2x of the CPU fp32 run's
        # (floor: a hundredth of one step, for tensors that both runs track almost exactly)
        assert e_hip.max() <= 2.02 * lr * steps + 1e-6, (k, e_hip.max())
        assert e_hip.mean() <= 2.0 * chain * e_ref.mean() + 0.01 * lr, (k, e_hip.mean(), e_ref.mean())
        # share of elements that drifted by more than a third of a step per step
        far_hip, far_ref = (e_hip > 0.3 * lr * steps).mean(), (e_ref > 0.3 * lr * steps).mean()
        assert far_hip <= far_ref + max(0.02 * chain, 2.0 / e_hip.size), (k, far_hip, far_ref)
    assert tot_hip / n <= 1.5 * chain * tot_ref / n + 1e-7, (tot_hip / n, tot_ref / n)


def _assert_grads_match_a_cpu_run(grads, ref32, ref64, tol_of, what=""):
    """Parameter gradients against the reference model run on the CPU.  An fp32 run of the SAME network can take a ReLU /
    max-pool decision differently from the exact (float64) forward pass when two candidates are within rounding of each
    other; one such flip moves a dY element by one position and changes a channel's gradients by ~1/sqrt(terms) — 1e-4 ...
    1e-2 at small batches, far above any rounding of the sums.  tools/wgrad_error.py itemises them
    (profiles/r05_accuracy_attribution_b7_t999.txt): at B=7, T=999 exactly one decision differs from float64 — block 2,
    (n, c, j) = (4, 34, 110), candidates 1.05e-7 apart under a live gradient of 1.8e-5 — and it is flipped by the CPU fp32
    run and by the HIP path's UNFUSED leaves (a hooked model; round 4's table counted those and called them "HIP"), NOT by
    the fused ConvBlock launches this test and the train step run: those take the float64 run's decisions (0 flips, gradients
    5e-7 rel-RMS from float64 where the CPU fp32 run sits at 7e-3).  Both CPU runs are legitimate outcomes of the reference,
    so every tensor must match AT LEAST ONE of them within the tolerance — a HIP-only flip still fails."""
    for k, p32 in ref32.named_parameters():
        g, g32, g64 = grads[k], p32.grad.numpy(), dict(ref64.named_parameters())[k].grad.numpy()
        tol = tol_of(k, g32)
        e32, e64 = np.abs(g - g32).max(), np.abs(g - g64.astype(np.float32)).max()
        assert min(e32, e64) <= tol, f"{k} {what}: |hip - cpu fp32| = {e32:.3e}, |hip - cpu float64| = {e64:.3e}, tol {tol:.1e}"


@pytest.mark.parametrize("B,T", [(1, 1000), (3, 16), (5, 17), (2, 5000), (7, 999)])
def test_ragged_batches_and_lengths_vs_cpu_oracle(B, T):
    """Last batch is ragged (drop_last unset), Grad-CAM uses B=1, any T >= 16 must work
    (SURVEY §8b); T=5000 is the real window length."""
    from src.utils.seed import set_seed
    from ecg_hip import functional as hipF
    ctor, rctor, C, demo = _ctors()["mm"]
    set_seed(1)
    model = ctor().to(DEV)
    R.seed_all(1)
    ref = rctor()
    ref64 = copy.deepcopy(ref).double()
    x, xd, y = R.synthetic_batch(B, T, C, gen_seed=B * 100 + T, demo=True)
    for train in ([True, False] if B > 1 else [False]):
        model.train(train), ref.train(train), ref64.train(train)
        model.zero_grad(), ref.zero_grad(), ref64.zero_grad()
        logits = model(x.to(DEV), xd.to(DEV))
        rlogits = ref(x, xd)
        np.testing.assert_allclose(logits.detach().cpu().numpy(), rlogits.detach().numpy(), atol=1e-4)
        hipF.binary_cross_entropy_with_logits(logits, y.to(DEV)).backward()
        torch.nn.functional.binary_cross_entropy_with_logits(rlogits, y).backward()
        torch.nn.functional.binary_cross_entropy_with_logits(ref64(x.double(), xd.double()), y.double()).backward()
        # absolute 1e-4 at unit scale; eval-mode BN with fresh running stats leaves activations
        # (hence gradients) un-normalised, so scale the bound with the gradient magnitude
        _assert_grads_match_a_cpu_run(
            {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}, ref, ref64,
            lambda k, g: 1e-6 if ".net.0.bias" in k and train else 1e-4 * max(1.0, float(np.abs(g).max())),
            what=f"train={train}")


def test_eval_loop_api_and_device_side_loss():
    """eval_one_epoch / eval_one_epoch_demo: metrics dict + BCE, ragged last batch."""
    from src.models.ecg_cnn import ECGCNN
    from src.training.loop import eval_one_epoch
    from src.training.metrics import compute_metrics
    g = golden("g3_eval_known_answer")
    m = ECGCNN(num_labels=5)
    ck = sd_from_npz(golden("g3_ckpt_baseline"))
    m.load_state_dict(ck)
    m.to(DEV)
    ref = R.RefECGCNN(num_labels=5)
    ref.load_state_dict(ck)
    ref.eval()
    x = torch.from_numpy(g["ecg"])[:, :, :2000]
    x = torch.cat([x, x.flip(-1)])[:5]
    y = (torch.arange(25).reshape(5, 5) % 3 == 0).float()
    loader = torch.utils.data.DataLoader(_DS(x, y), batch_size=2, shuffle=False)   # batches 2,2,1
    out = eval_one_epoch(m, loader, DEV)
    with torch.no_grad():
        rl = ref(x)
    rloss = torch.nn.functional.binary_cross_entropy_with_logits(rl, y).item()
    # sample-weighted mean of per-batch means == global mean for this loss
    assert abs(out["bce_loss"] - rloss) < 1e-5
    refm = compute_metrics(y.numpy(), torch.sigmoid(rl).numpy())
    for k in ("auroc_macro", "auprc_macro", "f1_macro"):
        assert abs(out[k] - refm[k]) < 1e-6 or (np.isnan(out[k]) and np.isnan(refm[k]))


def test_inference_takes_the_one_launch_conv_blocks():
    """model.eval() under torch.no_grad() (the reference's eval loops, src/training/loop.py:47-48): blocks 0-2
    run as ONE launch each (conv + running-stat BN + ReLU + pool); with grad enabled the same weights take
    the train-capable sequence; both agree with the stock-torch restatement."""
    from ecg_hip import _lib
    from src.models.ecg_multimodal import ECGMultimodal
    from src.utils.seed import set_seed
    set_seed(42)
    m = ECGMultimodal().to(DEV).eval()
    R.seed_all(42)
    ref = R.RefECGMultimodal().eval()
    with torch.no_grad():                       # give the running statistics some non-trivial values
        for mod, rmod in zip(m.modules(), ref.modules()):
            if isinstance(mod, torch.nn.BatchNorm1d):
                rm, rv = torch.randn(mod.num_features) * 0.2, torch.rand(mod.num_features) + 0.5
                mod.running_mean.copy_(rm), mod.running_var.copy_(rv)
                rmod.running_mean.copy_(rm), rmod.running_var.copy_(rv)
    x, xd, _ = R.synthetic_batch(6, 1000, 5, demo=True)
    with _lib.kernel_timing() as kt, torch.no_grad():
        out = m(x.to(DEV), xd.to(DEV))
    names = [k[0] for k in kt.result]
    assert names.count("ecg_conv1d_bn_relu_pool_eval_fwd") == 3, names
    assert names.count("ecg_conv1d_bn_relu_pool_gap_eval_fwd") == 1, names      # last block: the average pool too
    assert "ecg_bn_finalize" not in names and "ecg_conv1d_fwd" not in names
    with _lib.kernel_timing() as kt2:
        out_g = m(x.to(DEV), xd.to(DEV))        # grad mode on: parameters need gradients
    assert not any("eval_fwd" in k[0] for k in kt2.result)
    with torch.no_grad():
        want = ref(x, xd)
    assert (out.cpu() - want).abs().max() <= 1e-4
    assert (out_g.detach().cpu() - want).abs().max() <= 1e-4
    assert (out - out_g).abs().max() <= 5e-5


def test_gradients_land_in_the_flat_buffer_and_still_accumulate():
    """With FlatAdamW the backward kernels write every parameter gradient straight into its slice of the flat
    gradient (no per-step gather); a second backward WITHOUT zero_grad must still accumulate correctly
    (the sink is only used while .grad is None), and zero_grad(set_to_none=False) must not double anything."""
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_multimodal import ECGMultimodal
    from src.training.loop_demo import bce_loss_fn
    from src.utils.seed import set_seed
    set_seed(42)
    m = ECGMultimodal().to(DEV).train()
    opt = FlatAdamW(m.parameters(), lr=1e-4, weight_decay=1e-4)
    x, xd, y = (t.to(DEV) for t in R.synthetic_batch(8, 1000, 5, demo=True))
    opt.zero_grad()
    bce_loss_fn(m(x, xd), y).backward()
    lo, hi = opt.flat_grad.data_ptr(), opt.flat_grad.data_ptr() + 4 * opt.flat_grad.numel()
    for n, p in m.named_parameters():
        assert lo <= p.grad.data_ptr() < hi, f"{n}: gradient was not written into the flat buffer"
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    flat1 = opt.flat_grad.clone()
    assert torch.equal(flat1, torch.cat([g1[n].reshape(-1) for n, _ in m.named_parameters()]))
    # same batch again, no zero_grad: autograd adds -> exactly twice (BN running stats do not enter the gradient)
    bce_loss_fn(m(x, xd), y).backward()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[n], rtol=1e-6, atol=1e-12), n
    # zero-filled (not None) gradients: the kernels must not write under autograd's feet
    opt.zero_grad(set_to_none=False)
    bce_loss_fn(m(x, xd), y).backward()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, g1[n], rtol=1e-6, atol=1e-12), n
    # a step from this state equals a step from freshly computed gradients
    before = opt.flat_param.clone()
    opt.step()
    assert not torch.equal(before, opt.flat_param)
    # dropping the optimizer unregisters its sinks: gradients become ordinary tensors again.  (The flat buffer itself is
    # kept alive here: once freed, the caching allocator may hand its address range to one of the new gradients.)
    keep_flat = opt.flat_grad
    del opt
    import gc
    gc.collect()
    for p in m.parameters():
        p.grad = None
    bce_loss_fn(m(x, xd), y).backward()
    assert all(not (lo <= p.grad.data_ptr() < hi) for p in m.parameters())
    del keep_flat


def test_hooked_block_trains_like_the_fused_path_with_flat_optimizer():
    """A forward hook on an inner module switches that block to the unfused leaves, whose gradients are ordinary
    tensors while the other blocks write into FlatAdamW's flat buffer: the optimizer must repair the mix
    (no gather may read what it overwrites) and land on the same parameters as the all-fused step."""
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_cnn import ECGCNN
    from src.training.loop import train_one_epoch
    from src.utils.seed import set_seed
    x, y = R.synthetic_batch(8, 1000, 5)
    states = []
    for hooked in (False, True):
        set_seed(42)
        m = ECGCNN(num_labels=5).to(DEV)
        seen = []
        if hooked:
            m.backbone[2].net[0].register_forward_hook(lambda mod, inp, out: seen.append(tuple(out.shape)))
        opt = FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
        for _ in range(2):
            train_one_epoch(m, torch.utils.data.DataLoader(_DS(x, y), batch_size=8), opt, DEV)
        assert (len(seen) == 2 and seen[0] == (8, 128, 250)) if hooked else not seen
        states.append({k: v.detach().cpu() for k, v in m.state_dict().items()})
    for k in states[0]:
        if k.endswith("num_batches_tracked"):
            assert int(states[0][k]) == int(states[1][k]) == 2
        else:
            # two AdamW steps at lr 1e-3: noise-level gradients may move by +-lr per step on either path
            assert (states[0][k] - states[1][k]).abs().max() <= 2.02 * 1e-3 * 2, k
            assert ((states[0][k] - states[1][k]).abs() > 0.3 * 1e-3 * 2).float().mean() <= 0.03 or states[0][k].numel() < 70, k


def test_legacy_concat_fusion_model_vs_stock_torch():
    """§8(f)-4: the reconstructed concat-fusion model on the HIP leaves against the same module
    tree built from stock torch layers (no reference output exists for it)."""
    import torch.nn as nn
    from src.models.ecg_demo_concat import ECGDemoConcat
    from src.utils.seed import set_seed
    set_seed(3)
    m = ECGDemoConcat(dropout=0.0).to(DEV).train()

    class Ref(nn.Module):
        def __init__(self):
            super().__init__()
            self.ecg_encoder = R.RefECGCNN(num_labels=5)
            self.demo_encoder = nn.Module()
            self.demo_encoder.net = nn.Sequential(nn.Linear(5, 32), nn.ReLU(), nn.Linear(32, 64), nn.ReLU())
            self.classifier = nn.Sequential(nn.Linear(320, 256), nn.ReLU(), nn.Dropout(0.0), nn.Linear(256, 5))

        def forward(self, x, xd):
            _, z = self.ecg_encoder(x, return_features=True)
            return self.classifier(torch.cat([z, self.demo_encoder.net(xd)], dim=1))

    ref = Ref().train()
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()}, strict=True)
    x, xd, y = R.synthetic_batch(6, 500, 5, demo=True)
    logits = m(x.to(DEV), xd.to(DEV))
    rl = ref(x, xd)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), rl.detach().numpy(), atol=1e-4)
    from ecg_hip import functional as hipF
    hipF.binary_cross_entropy_with_logits(logits, y.to(DEV)).backward()
    torch.nn.functional.binary_cross_entropy_with_logits(rl, y).backward()
    for (k, a), (_, b) in zip(m.named_parameters(), ref.named_parameters()):
        if b.grad is None:      # ecg_encoder.head is unused by the concat model
            assert a.grad is None or float(a.grad.abs().max()) == 0.0, k
            continue
        tol = 1e-6 if ".net.0.bias" in k and "backbone" in k else 1e-4
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), atol=tol, err_msg=k)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_short_training_run_reduces_loss(precision):
    """End-to-end sanity through the loop API: 25 AdamW steps on a fixed synthetic batch must fit it — in fp32 and in
    the opt-in bf16 mode (bf16 conv operands, bf16 storage of y / p / dp between the kernels), whose loss curve must
    stay within 5 % of the fp32 one at every epoch."""
    from ecg_hip import functional as hipF
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_multimodal import ECGMultimodal
    from src.training.loop_demo import eval_one_epoch_demo, train_one_epoch_demo
    from src.utils.seed import set_seed
    if precision == "bf16":
        ref = _short_run_losses("fp32")
        got = _short_run_losses("bf16")
        assert got[-1] < 0.6 * got[0], got
        assert all(abs(a - b) <= 0.05 * b for a, b in zip(got, ref)), (got, ref)
        return
    set_seed(0)
    model = ECGMultimodal().to(DEV)
    batch = tuple(t.to(DEV) for t in R.synthetic_batch(64, 1000, 5, demo=True))
    opt = FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)

    class Loader:
        dataset = range(64 * 5)

        def __iter__(self):
            return iter([batch] * 5)

    losses = [train_one_epoch_demo(model, Loader(), opt, DEV) for _ in range(5)]
    assert losses[-1] < 0.6 * losses[0], losses
    out = eval_one_epoch_demo(model, Loader(), DEV)
    assert set(out) == {"auroc_macro", "auprc_macro", "f1_macro", "bce_loss"} and np.isfinite(out["bce_loss"])


def _short_run_losses(precision):
    from ecg_hip import functional as hipF
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_multimodal import ECGMultimodal
    from src.training.loop_demo import train_one_epoch_demo
    from src.utils.seed import set_seed
    set_seed(0)
    model = ECGMultimodal().to(DEV)
    batch = tuple(t.to(DEV) for t in R.synthetic_batch(64, 1000, 5, demo=True))
    opt = FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)

    class Loader:
        dataset = range(64 * 5)

        def __iter__(self):
            return iter([batch] * 5)

    with hipF.conv_precision(precision):
        return [train_one_epoch_demo(model, Loader(), opt, DEV) for _ in range(5)]


@pytest.mark.parametrize("B,T", [(8, 5000), (19, 1000), (256, 5000)])
def test_bf16_mixed_precision_train_step_config5(B, T):
    """BASELINE.json config 5: ECGCNN(num_labels=1) (AF binary, configs/af_binary.yaml shape), long windows, bf16 conv
    operands — (256, 5000) is the configuration at its STATED size, end to end.  No fp32-level parity is claimed: the step
    must track the fp32 CPU oracle at bf16 accuracy (the reference has no mixed precision to compare with).  B=19: a
    batch that no tile or split of the row kernels divides.

    Bars = about 1.3x the largest value measured over these cases (round 4, tools/bf16_grad_error.py; the measured
    value of every tensor is in the assertion message):
      logits 1.2e-3 -> 3e-3, loss 4.2e-5 -> 2e-4;
      tensors BEHIND the last bf16 conv (block-3 BatchNorm, proj, head): rel <= 3.7e-3, 1-cos <= 6.8e-6 -> 1e-2 / 2e-5;
      tensors upstream of it: rel 0.064 ... 0.188, 1-cos 0.002 ... 0.0174 -> 0.25 / 0.025.  These are NOT rounding of sums:
      a bf16 y (eps 2^-8) flips ~0.3 % of the ReLU / pooling decisions, each flip moves one dY element, and the relative
      error of a gradient grows like sqrt(2 x fraction flipped) ~ 0.08 per block (0.07 / 0.11 / 0.14 / 0.157 from block 3
      down at the full size) — the price of bf16 activations, the same with fp32 storage of y (0.093 ... 0.14)."""
    from ecg_hip import functional as hipF
    from src.models.ecg_cnn import ECGCNN
    from src.utils.seed import set_seed
    set_seed(42)
    model = ECGCNN(num_labels=1).to(DEV).train()
    R.seed_all(42)
    ref = R.RefECGCNN(num_labels=1).train()
    x, y = R.synthetic_batch(B, T, 1)
    with hipF.conv_precision("bf16"):
        logits = model(x.to(DEV))
        loss = hipF.binary_cross_entropy_with_logits(logits, y.to(DEV))
        loss.backward()
    assert hipF.get_conv_precision() == "fp32"
    rl = ref(x)
    rloss = torch.nn.functional.binary_cross_entropy_with_logits(rl, y)
    rloss.backward()
    dlogit = float((logits.detach().cpu() - rl.detach()).abs().max())
    assert 1e-6 < dlogit < 3e-3, dlogit                 # bf16 accuracy — and it really took the bf16 path
    assert abs(loss.item() - rloss.item()) < 2e-4, (loss.item(), rloss.item())
    measured = {}
    for (k, a), (_, b) in zip(model.named_parameters(), ref.named_parameters()):
        if ".net.0.bias" in k:
            continue
        g, r = a.grad.cpu().numpy().ravel().astype(np.float64), b.grad.numpy().ravel().astype(np.float64)
        rel = np.linalg.norm(g - r) / (np.linalg.norm(r) + 1e-12)
        omc = 1.0 - float(g @ r) / (np.linalg.norm(g) * np.linalg.norm(r) + 1e-20)
        measured[k] = (round(float(rel), 5), float(f"{omc:.3g}"))
    for k, (rel, omc) in measured.items():
        behind = k.startswith(("backbone.3.net.1", "proj", "head"))
        rel_bar, omc_bar = (1e-2, 2e-5) if behind else (0.25, 0.025)
        assert rel <= rel_bar and omc <= omc_bar, f"{k}: rel {rel} (bar {rel_bar}), 1-cos {omc} (bar {omc_bar}); all: {measured}"


@pytest.mark.parametrize("case", ["input_grad", "frozen1", "frozen2", "frozen3", "unaligned", "eval"])
def test_bf16_mode_runs_the_fp32_kernels_wherever_the_mixed_form_does_not_apply(case, monkeypatch):
    """The mixed-precision step has ONE form (bf16 rows between the kernels of a training block); a block that does not fit
    it — a BatchNorm frozen (eval-mode statistics) under gradients, a model that is evaluating — runs the exact fp32 kernels,
    and the tensors handed between a bf16 block and an fp32 block are fp32 (a bf16 block casts an fp32 input it cannot read
    in place: an input gradient is wanted, or the tensor is not 16-byte aligned).  Checked per block on the entry points each
    ConvBlock launches, and on the result."""
    from ecg_hip import _lib, functional as hipF
    from src.models.ecg_cnn import ECGCNN
    from src.utils.seed import set_seed
    x0, y = R.synthetic_batch(6, 1000, 5)
    set_seed(3)
    model = ECGCNN(num_labels=5).to(DEV).train()
    if case.startswith("frozen"):
        model.backbone[int(case[-1])].net[1].eval()
    if case == "eval":
        model.eval()
    x = x0.to(DEV)
    if case == "unaligned":             # a view 4 bytes into a larger buffer: contiguous, not 16-byte aligned
        buf = torch.empty(x.numel() + 1, device=DEV)
        buf[1:].copy_(x.reshape(-1))
        x = buf[1:].view_as(x)
        assert x.data_ptr() % 16 != 0 and x.is_contiguous()
    x = x.requires_grad_(case == "input_grad")
    names, raw = [], _lib.call

    def spy(name, *args):
        names.append(name)
        raw(name, *args)
    monkeypatch.setattr(_lib, "call", spy)
    monkeypatch.setattr(hipF, "_call", spy)
    with hipF.conv_precision("bf16"):
        if case == "eval":
            with torch.no_grad():
                out = model(x)
        else:
            out = model(x)
            hipF.binary_cross_entropy_with_logits(out, y.to(DEV)).backward()
    monkeypatch.undo()
    fwd = [n for n in names if n in ("ecg_conv1d_fwd", "ecg_conv1d_fwd_bf16_yh", "ecg_conv1d_bn_relu_pool_eval_fwd",
                                     "ecg_conv1d_bn_relu_pool_gap_eval_fwd")]
    F32, H = "ecg_conv1d_fwd", "ecg_conv1d_fwd_bf16_yh"
    want = {"input_grad": [F32, H, H, H],         # 12 input channels have no bf16 input-gradient kernel: block 0 runs fp32,
                                                  # block 1 casts the fp32 activation to bf16 rows and widens its dx again
            "frozen1": [H, F32, H, H], "frozen2": [H, H, F32, H], "frozen3": [H, H, H, F32],
            "unaligned": [H, H, H, H],
            "eval": ["ecg_conv1d_bn_relu_pool_eval_fwd"] * 3 + ["ecg_conv1d_bn_relu_pool_gap_eval_fwd"]}[case]
    assert fwd == want, (case, fwd)
    if case == "eval":
        set_seed(3)
        ref = ECGCNN(num_labels=5).to(DEV).eval()
        with torch.no_grad():
            assert torch.equal(out, ref(x0.to(DEV)))        # inference in bf16 mode IS the fp32 one-launch path
        return
    # every weight gradient came from the kernel family of its block's form: one bf16 form, one fp32 form
    wg = [n for n in names if n in ("ecg_conv1d_bwd_weight_bias_ld", "ecg_conv1d_bwd_weight_bias_bf16_ncl")]
    assert wg == [{F32: "ecg_conv1d_bwd_weight_bias_ld", H: "ecg_conv1d_bwd_weight_bias_bf16_ncl"}[f] for f in reversed(want)]
    # ... and the step tracks the fp32 run at bf16 accuracy
    set_seed(3)
    ref = ECGCNN(num_labels=5).to(DEV).train()
    if case.startswith("frozen"):
        ref.backbone[int(case[-1])].net[1].eval()
    xr = x0.to(DEV).requires_grad_(case == "input_grad")
    hipF.binary_cross_entropy_with_logits(ref(xr), y.to(DEV)).backward()
    g = torch.cat([p.grad.reshape(-1) for k, p in model.named_parameters() if ".net.0.bias" not in k])
    gr = torch.cat([p.grad.reshape(-1) for k, p in ref.named_parameters() if ".net.0.bias" not in k])
    assert torch.nn.functional.cosine_similarity(g, gr, dim=0).item() > 0.99
    if case == "input_grad":
        assert x.grad.dtype == torch.float32 and x.grad.shape == x0.shape
        assert torch.nn.functional.cosine_similarity(x.grad.reshape(-1), xr.grad.reshape(-1), dim=0).item() > 0.98
    if case == "frozen3":       # nothing bf16 sits between the loss and block 3: its parameter gradients are the fp32 path's
        for k in ("backbone.3.net.1.weight", "proj.weight", "head.weight"):
            a, b = dict(model.named_parameters())[k].grad, dict(ref.named_parameters())[k].grad
            assert float((a - b).norm() / b.norm()) < 2e-2, k


def test_graphed_train_step_matches_eager():
    """A captured hipGraph of the whole step (fwd + BCE + bwd + FlatAdamW with a device-side step
    counter) replays to the same parameters as the eager loop."""
    from ecg_hip.graph import GraphedTrainStep
    from ecg_hip.optim import FlatAdamW
    from ecg_hip import functional as hipF
    from src.models.ecg_multimodal import ECGMultimodal
    from src.utils.seed import set_seed
    batch = tuple(t.to(DEV) for t in R.synthetic_batch(16, 1000, 5, demo=True))
    set_seed(7)
    eager = ECGMultimodal().to(DEV).train()
    eopt = FlatAdamW(eager.parameters(), lr=1e-3, weight_decay=1e-4)
    set_seed(7)
    graphed = ECGMultimodal().to(DEV).train()
    gopt = FlatAdamW(graphed.parameters(), lr=1e-3, weight_decay=1e-4)
    sd0 = {k: v.clone() for k, v in graphed.state_dict().items()}
    step = GraphedTrainStep(graphed, gopt, batch, warmup=2)
    # the constructor's warm-up iterations are real steps, but it puts back everything they advanced
    for k, v in graphed.state_dict().items():
        assert torch.equal(v, sd0[k]), k
    assert gopt.steps_taken == 0 and float(gopt.flat_m.abs().max()) == 0.0 and float(gopt.flat_v.abs().max()) == 0.0
    losses = []
    for _ in range(4):
        eopt.zero_grad()
        loss = hipF.binary_cross_entropy_with_logits(eager(*batch[:-1]), batch[-1])
        loss.backward()
        eopt.step()
        losses.append(loss.item())
        gl = step(*batch)
        assert abs(gl.item() - losses[-1]) < 1e-6
    assert gopt.steps_taken == 4 and eopt.steps_taken == 4
    assert abs(step.mean_loss_and_reset(4) - float(np.mean(losses))) < 1e-6
    for (k, a), (_, b) in zip(graphed.state_dict().items(), eager.state_dict().items()):
        if k.endswith("num_batches_tracked"):
            assert int(a.item()) == int(b.item()) == 4
        else:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-6, err_msg=k)


def test_parameter_used_twice_in_one_backward_matches_stock_torch():
    """Model called twice before backward() (shared weights): each parameter gets TWO gradients in one pass.
    Only the first may land in the flat-buffer sink; autograd must end with dw1 + dw2 (not 2*dw2)."""
    from ecg_hip import functional as hipF
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_multimodal import ECGMultimodal
    from src.utils.seed import set_seed
    set_seed(42)
    model = ECGMultimodal().to(DEV).train()
    R.seed_all(42)
    ref = R.RefECGMultimodal().train()
    opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)      # registers the gradient sinks
    b1 = R.synthetic_batch(6, 500, 5, demo=True)
    b2 = R.synthetic_batch(6, 500, 5, gen_seed=77, demo=True)
    for rep in range(2):                     # second repetition: the handed-out set was cleared by the engine callback
        opt.zero_grad()
        loss = (hipF.binary_cross_entropy_with_logits(model(b1[0].to(DEV), b1[1].to(DEV)), b1[2].to(DEV))
                + hipF.binary_cross_entropy_with_logits(model(b2[0].to(DEV), b2[1].to(DEV)), b2[2].to(DEV)))
        loss.backward()
        ref.zero_grad()
        rl = (torch.nn.functional.binary_cross_entropy_with_logits(ref(b1[0], b1[1]), b1[2])
              + torch.nn.functional.binary_cross_entropy_with_logits(ref(b2[0], b2[1]), b2[2]))
        rl.backward()
        assert abs(loss.item() - rl.item()) < 1e-5
        for (k, a), (_, b) in zip(model.named_parameters(), ref.named_parameters()):
            tol = 1e-6 if ".net.0.bias" in k else 1e-4
            np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), atol=tol, err_msg=f"{k} rep {rep}")
        # undo the BN running-stat drift between the two models for the second repetition
        ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})


@pytest.mark.parametrize("name", ["cnn5", "mm"])
def test_train_step_is_bitwise_reproducible(name):
    """Two identical B=256 train steps from the same state: bit-identical flat gradient, BN buffers and
    parameters (fixed-order weight-gradient slabs, ordered statistics partials, no atomics)."""
    from ecg_hip import functional as hipF
    from ecg_hip.optim import FlatAdamW
    from src.utils.seed import set_seed
    ctor, _, C, demo = _ctors()[name]
    batch = tuple(t.to(DEV) for t in R.synthetic_batch(256, 1000, C, demo=demo))
    outs = []
    for rep in range(2):
        set_seed(42)
        model = ctor().to(DEV).train()
        opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
        for _ in range(2):
            opt.zero_grad()
            loss = hipF.binary_cross_entropy_with_logits(model(*batch[:-1]), batch[-1])
            loss.backward()
            opt.step()
        outs.append((opt.flat_grad.clone(), loss.detach().clone(), {k: v.clone() for k, v in model.state_dict().items()}))
        del model, opt
    assert torch.equal(outs[0][0], outs[1][0]), "flat gradient differs between identical runs"
    assert torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k


@pytest.mark.parametrize("name", ["cnn5", "mm"])
def test_loop_api_replays_a_captured_step_and_matches_the_eager_loop(name, monkeypatch):
    """train_one_epoch[_demo] with a FlatAdamW replay the step as one hipGraph per batch shape once a shape has been
    seen twice (the reference's batch sizes are host-bound eagerly: configs/ecg_baseline.yaml:12, src/training/
    loop.py:22-36).  Same kernels in the same order: parameters, buffers, counters and epoch losses must be
    BIT-identical to the eager loop (ECG_HIP_LOOP_GRAPH=0), ragged last batch included."""
    from ecg_hip.graph import LoopStepper
    from ecg_hip.optim import FlatAdamW
    from src.training.loop import train_one_epoch
    from src.training.loop_demo import train_one_epoch_demo
    from src.utils.seed import set_seed
    ctor, _, C, demo = _ctors()[name]
    data = R.synthetic_batch(8 * 5 + 3, 500, C, demo=demo)               # 5 full batches of 8 and a ragged one of 3
    loader = torch.utils.data.DataLoader(_DS(*data), batch_size=8, shuffle=False)
    fn = train_one_epoch_demo if demo else train_one_epoch
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ECG_HIP_LOOP_GRAPH", mode)
        set_seed(11)
        model = ctor().to(DEV)
        opt = FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
        losses = [fn(model, loader, opt, DEV) for _ in range(3)]
        st = getattr(opt, "_ecg_loop_steppers", {})
        runs[mode] = (losses, {k: v.clone() for k, v in model.state_dict().items()}, opt.steps_taken, st)
    assert runs["0"][3] == {}                                             # opt-out: nothing was captured
    (stepper,) = runs["1"][3].values()
    assert len(stepper.graphs) == 2                                       # the full shape (epoch 1) and the ragged one (epoch 2)
    assert runs["0"][2] == runs["1"][2] == 18
    assert runs["0"][0] == runs["1"][0], (runs["0"][0], runs["1"][0])     # epoch losses: same double sums
    for k, v in runs["0"][1].items():
        assert torch.equal(v, runs["1"][1][k]), k


def test_loop_api_stays_eager_when_a_graph_would_change_behaviour(monkeypatch):
    from ecg_hip.graph import LoopStepper
    monkeypatch.delenv("ECG_HIP_LOOP_GRAPH", raising=False)
    from ecg_hip.optim import FlatAdamW as _F
    from src.models.ecg_cnn import ECGCNN as _M
    _m = _M(num_labels=5).to(DEV)
    assert LoopStepper.for_loop(_m, _F(_m.parameters(), lr=1e-3), True) is None          # opt-in: off by default
    monkeypatch.setenv("ECG_HIP_LOOP_GRAPH", "1")
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_cnn import ECGCNN
    model = ECGCNN(num_labels=5).to(DEV)
    flat = FlatAdamW(model.parameters(), lr=1e-3)
    assert LoopStepper.for_loop(model, flat, True) is not None
    assert LoopStepper.for_loop(model, torch.optim.AdamW(model.parameters()), True) is None      # stock optimizer
    h = model.backbone[-1].net[0].register_forward_hook(lambda m, i, o: None)                      # Grad-CAM style hook
    assert LoopStepper.for_loop(model, flat, True) is None
    h.remove()
    assert LoopStepper.for_loop(model, flat, True) is not None
    with torch.no_grad():
        assert LoopStepper.for_loop(model, flat, True) is None
    extra = torch.nn.Linear(3, 3).to(DEV)                                                          # a parameter the optimizer does not own
    model.extra = extra
    assert LoopStepper.for_loop(model, flat, True) is None
