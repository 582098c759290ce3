"""Pins the CPU oracle (oracle/) against the golden vectors captured from the
reference import (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from util import check_put, golden, sd_from_npz
from oracle import ref_models as R

G1_TOL = 2e-5   # |y| ~ 1, fp32 reference vs double-accumulating oracle


def test_g1_conv_ops(oracle):
    g = golden("g1_conv_ops")
    for c in range(int(g["ncases"])):
        p = f"c{c}_"
        blk = int(g[p + "blk"])
        w, b = g[f"w_blk{blk}"], g[f"b_blk{blk}"]
        x, dy = g[p + "x"], g[p + "dy"]
        y = oracle.conv1d_fwd(x, w, b, 7)
        np.testing.assert_allclose(y, g[p + "y"], atol=G1_TOL)
        dx = oracle.conv1d_bwd_data(dy, w, x.shape[2], 7)
        np.testing.assert_allclose(dx, g[p + "dx"], atol=5e-5)
        dw, db = oracle.conv1d_bwd_weight(dy, x, 15, 7)
        check_put(g, p + "dw", dw, atol=5e-5)
        np.testing.assert_allclose(db, g[p + "db"], atol=5e-5)


def test_g2_block_train(oracle):
    g1 = golden("g1_conv_ops")
    g = golden("g2_block_train")
    for bi in range(int(g["nblocks"])):
        p = f"b{bi}_"
        w, b = g1[f"w_blk{bi}"], g1[f"b_blk{bi}"]
        gamma, beta = g[p + "gamma"], g[p + "beta"]
        C = w.shape[0]
        rm, rv = np.zeros(C, np.float32), np.ones(C, np.float32)
        nbt = np.zeros((), np.int64)
        for step in (1, 2):
            q = p + f"s{step}_"
            x, dp = g[q + "x"], g[q + "dp"]
            y = oracle.conv1d_fwd(x, w, b, 7)
            np.testing.assert_allclose(y, g[q + "conv_out"], atol=G1_TOL)
            mean, invstd = oracle.bn_stats(y, rm, rv, nbt)
            pooled = oracle.bn_relu_pool_fwd(y, gamma, beta, mean, invstd)
            np.testing.assert_allclose(pooled, g[q + "pooled"], atol=3e-5)
            np.testing.assert_allclose(rm, g[q + "running_mean"], atol=1e-6)
            np.testing.assert_allclose(rv, g[q + "running_var"], atol=1e-6)
            assert int(nbt) == int(g[q + "nbt"]) == step          # integer: exact
            dy, dgam, dbet = oracle.bn_relu_pool_bwd(y, dp, gamma, beta, mean, invstd)
            np.testing.assert_allclose(dgam, g[q + "dgamma"], atol=2e-4)
            np.testing.assert_allclose(dbet, g[q + "dbeta"], atol=2e-4)
            dw, db = oracle.conv1d_bwd_weight(dy, x, 15, 7)
            check_put(g, q + "dw", dw, atol=3e-4)
            # conv bias grad is mathematically zero under train-mode BN: absolute tolerance only
            np.testing.assert_allclose(db, g[q + "db"], atol=1e-4)
            dx = oracle.conv1d_bwd_data(dy, w, x.shape[2], 7)
            np.testing.assert_allclose(dx, g[q + "dx"], atol=3e-4)


def _models():
    return {"baseline": lambda: R.RefECGCNN(num_labels=5),
            "multimodal": lambda: R.RefECGMultimodal(),
            "af": lambda: R.RefECGCNN(num_labels=1)}


def test_g3_eval_known_answer_torch_restatement():
    g = golden("g3_eval_known_answer")
    x = torch.from_numpy(g["ecg"])
    demo = torch.from_numpy(g["demo"])
    for name, ctor in _models().items():
        m = ctor()
        m.load_state_dict(sd_from_npz(golden("g3_ckpt_" + name)), strict=True)
        m.eval()
        with torch.no_grad():
            logits = m(x, demo) if name == "multimodal" else m(x)
        np.testing.assert_allclose(logits.numpy(), g[name + "_logits"], atol=1e-5)
        prob = torch.sigmoid(logits).numpy()
        # committed CSV rows: coarse cross-check: 5e-4 on prob, y_pred exact away from 0.5
        np.testing.assert_allclose(prob, g[name + "_csv_prob"], atol=5e-4)
        pred = (prob >= 0.5).astype(np.int64)
        far = np.abs(prob - 0.5) > 5e-4
        assert np.array_equal(pred[far], g[name + "_csv_pred"][far])


def test_g3_eval_known_answer_c_oracle(oracle):
    g = golden("g3_eval_known_answer")
    x = g["ecg"][:1]
    for name in ("baseline", "multimodal"):
        z = golden("g3_ckpt_" + name)
        sd = {k: np.array(z[k]) for k in z.files if not k.startswith("meta_")}
        if name == "multimodal":
            logits, _ = oracle.multimodal_forward(sd, x, g["demo"][:1], train=False)
        else:
            logits, _, _ = oracle.ecgcnn_forward(sd, x, train=False)
        np.testing.assert_allclose(logits, g[name + "_logits"][:1], atol=1e-4)


CFGS = {"cnn5": (lambda: R.RefECGCNN(num_labels=5), 5, False),
        "cnn1": (lambda: R.RefECGCNN(num_labels=1), 1, False),
        "mm": (lambda: R.RefECGMultimodal(), 5, True)}


@pytest.mark.parametrize("B", [4, 32])
@pytest.mark.parametrize("name", list(CFGS))
def test_g4_train_step_torch_restatement(name, B):
    g = golden("g4_train_step")
    ctor, C, demo = CFGS[name]
    p = f"{name}_B{B}_"
    R.seed_all(42)
    m = ctor()
    cs = torch.stack([v.double().sum() for v in m.state_dict().values()]).numpy()
    np.testing.assert_allclose(cs, g[p + "init_checksum"], rtol=0, atol=0)   # same init RNG stream
    batch = R.synthetic_batch(B, 1000, C, demo=demo)
    xs = np.array([batch[0].double().sum().item(), batch[0].double().abs().sum().item()])
    np.testing.assert_allclose(xs, g[p + "x_checksum"], rtol=0, atol=0)
    assert np.array_equal(batch[-1].numpy(), g[p + "y"])
    opt = R.make_adamw(m, float(g[p + "lr"]), 1e-4)
    m.train()
    for step in (1, 2, 3):
        logits, loss = R.train_step(m, opt, batch)
        if step == 1:
            np.testing.assert_allclose(logits.numpy(), g[p + "logits0"], atol=1e-5)
            assert abs(loss - float(g[p + "loss0"])) < 1e-6
        assert abs(loss - float(g[p + f"epoch_loss{step}"])) < 1e-5
        if step in (1, 3):
            q = p + f"s{step}_"
            for k, v in m.state_dict().items():
                check_put(g, q + "sd_" + k, v, atol=2e-5)
            for k, prm in m.named_parameters():
                check_put(g, q + "grad_" + k, prm.grad, atol=2e-5)


@pytest.mark.parametrize("name", ["cnn5", "mm"])
def test_g4_first_step_c_oracle(oracle, name):
    """The hand-written backward decomposition (the one the HIP path uses) against the
    reference's autograd, B=4."""
    g = golden("g4_train_step")
    ctor, C, demo = CFGS[name]
    p = f"{name}_B4_"
    R.seed_all(42)
    m = ctor()
    sd = {k: v.numpy().copy() for k, v in m.state_dict().items()}
    batch = [t.numpy() for t in R.synthetic_batch(4, 1000, C, demo=demo)]
    if demo:
        logits, loss, grads, _ = oracle.multimodal_loss_and_grads(sd, *batch)
    else:
        logits, loss, grads = oracle.ecgcnn_loss_and_grads(sd, *batch)
    np.testing.assert_allclose(logits, g[p + "logits0"], atol=1e-5)
    assert abs(loss - float(g[p + "loss0"])) < 1e-6
    for k, v in grads.items():
        check_put(g, p + "s1_grad_" + k, v, atol=2e-5)
    for k in sd:
        if "running" in k or "num_batches" in k:
            check_put(g, p + "s1_sd_" + k, sd[k], atol=1e-6)


def test_adamw_oracle_matches_torch(oracle):
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(1000).astype(np.float32)
    t = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([t], lr=1.5e-3, weight_decay=1e-4)
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 4):
        gr = rng.standard_normal(1000).astype(np.float32)
        t.grad = torch.from_numpy(gr.copy())
        opt.step()
        oracle.adamw(p, gr, m, v, step, 1.5e-3, wd=1e-4)
        np.testing.assert_allclose(p, t.detach().numpy(), atol=1e-7, rtol=1e-6)
