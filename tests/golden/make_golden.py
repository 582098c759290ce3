#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, CPU torch).  The reference's
source never travels: only inputs and expected outputs are written, as .npz data.

    python tests/golden/make_golden.py            # writes tests/golden/g*.npz

Fixture list follows SURVEY.md §8(c):
  g1_conv_ops        per-op Conv1d fwd / dgrad / wgrad / bias-grad, 4 block geometries + edge L
  g2_block_train     ConvBlock in train mode: conv-out, pooled, BN buffers after 1 and 2 steps, grads
  g3_eval_*          committed checkpoints x demo samples -> logits (CPU re-run) + committed CSV rows
  g4_train_step      set_seed(42) models, 1 and 3 AdamW steps through the reference's own loops
  g5_ddp             8 shards of 4 with per-shard BN, averaged grads, post-step params
  g6_hooks           Grad-CAM hook tensors at the last Conv1d + d logit / d x_demo
Large tensors are stored as a strided subsample (`sub`) plus their L2 norm.
"""
import csv
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("ECG_REFERENCE_ROOT", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from src.models.ecg_cnn import ConvBlock, ECGCNN            # noqa: E402  (reference)
from src.models.ecg_multimodal import ECGMultimodal          # noqa: E402
from src.training.loop import train_one_epoch                # noqa: E402
from src.training.loop_demo import train_one_epoch_demo      # noqa: E402
from src.utils.seed import set_seed                          # noqa: E402

torch.set_num_threads(8)
META = dict(torch_version=torch.__version__, threads=torch.get_num_threads(),
            reference_pin="torch==2.8.0 (requirements.txt:54)")

SUB_TARGET = 2048


def sub(t):
    """Deterministic strided subsample used for large tensors (also in tests/util.py)."""
    a = t.detach().cpu().numpy().reshape(-1) if torch.is_tensor(t) else np.asarray(t).reshape(-1)
    stride = max(1, a.size // SUB_TARGET)
    return a[::stride].copy()


def put(d, name, t):
    a = np.array(t.detach().cpu().numpy() if torch.is_tensor(t) else t, copy=True)  # snapshot, not a view
    if a.size <= 4 * SUB_TARGET:
        d[name] = a
    else:
        d[name + "__sub"] = sub(a)
        d[name + "__norm"] = np.float64(np.linalg.norm(a.astype(np.float64)))


def synthetic(B, T, C, demo, gen_seed=1234):
    g = torch.Generator().manual_seed(gen_seed)
    x = torch.randn(B, 12, T, generator=g)
    y = (torch.rand(B, C, generator=g) < 0.3).float()
    if demo:
        return x, torch.rand(B, 5, generator=g), y
    return x, y


def save(name, d):
    d = dict(d)
    for k, v in META.items():
        d["meta_" + k] = np.array(str(v))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KB  ({len(d)} arrays)")


# ------------------------------------------------------------------ G1
def g1():
    set_seed(42)
    model = ECGCNN(num_labels=5)
    d = {}
    g = torch.Generator().manual_seed(7)
    # (block index, L): the four block geometries at small L, plus odd L and L<K / L==16 edges
    cases = [(0, 100), (1, 61), (2, 33), (3, 16), (0, 7), (1, 15), (3, 1)]
    for ci, (blk, L) in enumerate(cases):
        conv = model.backbone[blk].net[0]
        x = torch.randn(2, conv.in_channels, L, generator=g, requires_grad=True)
        y = conv(x)
        dy = torch.randn(y.shape, generator=g)
        conv.zero_grad()
        y.backward(dy)
        p = f"c{ci}_"
        d[p + "blk"], d[p + "L"] = np.int64(blk), np.int64(L)
        d[p + "x"] = x.detach().numpy()
        if ci < 4:      # weights of the four block geometries are stored once (cases 0-3); edge cases reuse them
            d[f"w_blk{blk}"], d[f"b_blk{blk}"] = conv.weight.detach().numpy(), conv.bias.detach().numpy()
        d[p + "dy"], d[p + "y"] = dy.numpy(), y.detach().numpy()
        d[p + "dx"], d[p + "db"] = x.grad.numpy(), conv.bias.grad.numpy().copy()
        put(d, p + "dw", conv.weight.grad)
    d["ncases"] = np.int64(len(cases))
    save("g1_conv_ops", d)


# ------------------------------------------------------------------ G2
def g2():
    d = {}
    g = torch.Generator().manual_seed(11)
    import copy
    set_seed(42)
    model = ECGCNN(num_labels=5)      # conv weights == g1_conv_ops w_blk{i}/b_blk{i}
    for ci, (cin, cout, L) in enumerate([(12, 32, 50), (32, 64, 33), (64, 128, 16), (128, 256, 9)]):
        blk = copy.deepcopy(model.backbone[ci])
        assert isinstance(blk, ConvBlock) and blk.net[0].in_channels == cin
        with torch.no_grad():  # non-trivial affine so dgamma/dbeta paths are exercised
            blk.net[1].weight.copy_(1.0 + 0.1 * torch.randn(cout, generator=g))
            blk.net[1].bias.copy_(0.1 * torch.randn(cout, generator=g))
        blk.train()
        p = f"b{ci}_"
        d[p + "gamma"], d[p + "beta"] = blk.net[1].weight.detach().numpy().copy(), blk.net[1].bias.detach().numpy().copy()
        for step in (1, 2):
            x = torch.randn(3, cin, L, generator=g, requires_grad=True)
            conv_out = {}
            h = blk.net[0].register_forward_hook(lambda m, i, o: conv_out.__setitem__("y", o.detach().clone()))
            out = blk(x)
            h.remove()
            dp = torch.randn(out.shape, generator=g)
            blk.zero_grad()
            out.backward(dp)
            q = p + f"s{step}_"
            d[q + "x"], d[q + "dp"] = x.detach().numpy(), dp.numpy()
            d[q + "conv_out"], d[q + "pooled"] = conv_out["y"].numpy(), out.detach().numpy()
            d[q + "running_mean"] = blk.net[1].running_mean.numpy().copy()
            d[q + "running_var"] = blk.net[1].running_var.numpy().copy()
            d[q + "nbt"] = blk.net[1].num_batches_tracked.numpy().copy()
            d[q + "dx"] = x.grad.numpy()
            put(d, q + "dw", blk.net[0].weight.grad)
            d[q + "db"] = blk.net[0].bias.grad.numpy().copy()
            d[q + "dgamma"], d[q + "dbeta"] = blk.net[1].weight.grad.numpy().copy(), blk.net[1].bias.grad.numpy().copy()
    d["nblocks"] = np.int64(4)
    save("g2_block_train", d)


# ------------------------------------------------------------------ G3
DEMO_IDS = (0, 3, 4)


def _csv_rows(path, rows):
    with open(path) as f:
        r = list(csv.reader(f))
    hdr, body = r[0], r[1:]
    prob_cols = [i for i, h in enumerate(hdr) if h.startswith("y_prob")]
    pred_cols = [i for i, h in enumerate(hdr) if h.startswith("y_pred")]
    prob = np.array([[float(body[i][c]) for c in prob_cols] for i in rows], np.float32)
    pred = np.array([[int(body[i][c]) for c in pred_cols] for i in rows], np.int64)
    return prob, pred


def _load_ckpt(path):
    ck = torch.load(path, map_location="cpu")
    return ck["model_state"] if "model_state" in ck else ck


def g3():
    meta = list(csv.DictReader(open(os.path.join(REF, "data/demo/meta.csv"))))
    singles = [m for m in meta if m["modality"] == "single"]
    mms = [m for m in meta if m["modality"] == "multimodal"]
    rows = [int(singles[i]["index_in_split"]) for i in DEMO_IDS]
    ecg = np.stack([np.load(os.path.join(REF, "data/demo", singles[i]["file"]))["ecg"] for i in DEMO_IDS])
    mm = [np.load(os.path.join(REF, "data/demo", mms[i]["file"])) for i in DEMO_IDS]
    ecg_mm = np.stack([m["ecg"] for m in mm])
    demo = np.stack([m["demo"] for m in mm])
    assert np.array_equal(ecg, ecg_mm), "single and multimodal demo ECGs differ"
    assert rows == [int(mms[i]["index_in_split"]) for i in DEMO_IDS]
    d = {"ecg": ecg, "demo": demo, "rows": np.array(rows, np.int64)}
    x = torch.from_numpy(ecg)

    specs = [
        ("baseline", "outputs/ecg_baseline/ckpts/ecg_baseline_best.pth",
         "outputs/ecg_baseline/preds/ecg_baseline_test_preds.csv", lambda: ECGCNN(num_labels=5)),
        ("multimodal", "outputs/ecg_multimodal/ckpts/ecg_multimodal_best.pth",
         "outputs/ecg_multimodal/preds/ecg_multimodal_test_preds.csv", lambda: ECGMultimodal()),
        ("af", "outputs/af_binary/ckpts/af_binary_best.pth",
         "outputs/af_binary/preds/af_binary_test_preds.csv", lambda: ECGCNN(num_labels=1)),
    ]
    for name, ck, cs, ctor in specs:
        sd = _load_ckpt(os.path.join(REF, ck))
        model = ctor()
        model.load_state_dict(sd, strict=True)
        model.eval()
        with torch.no_grad():
            logits = model(x, torch.from_numpy(demo)) if name == "multimodal" else model(x)
        prob, pred = _csv_rows(os.path.join(REF, cs), rows)
        d[name + "_logits"] = logits.numpy()
        d[name + "_csv_prob"], d[name + "_csv_pred"] = prob, pred
        err = np.abs(torch.sigmoid(logits).numpy() - prob).max()
        print(f"  {name}: max|sigmoid(logits) - csv prob| = {err:.2e}")
        # checkpoint weights as plain arrays (data, not pickle)
        save("g3_ckpt_" + name, {k: v.numpy() for k, v in sd.items()})
    save("g3_eval_known_answer", d)


# ------------------------------------------------------------------ G4 / G5
class _DS(torch.utils.data.Dataset):
    def __init__(self, *t):
        self.t = t

    def __len__(self):
        return self.t[0].shape[0]

    def __getitem__(self, i):
        return tuple(a[i] for a in self.t)


def _snapshot(d, prefix, model, with_grads):
    for k, v in model.state_dict().items():
        put(d, prefix + "sd_" + k, v)
    if with_grads:
        for k, p in model.named_parameters():
            put(d, prefix + "grad_" + k, p.grad)


def g4():
    d = {}
    cfgs = [("cnn5", lambda: ECGCNN(num_labels=5), 5, False, 1.5e-3),
            ("cnn1", lambda: ECGCNN(num_labels=1), 1, False, 1e-3),
            ("mm", lambda: ECGMultimodal(), 5, True, 1e-4)]
    for B in (4, 32):
        for name, ctor, C, demo, lr in cfgs:
            p = f"{name}_B{B}_"
            set_seed(42)
            model = ctor()
            put(d, p + "init_checksum", torch.stack([v.double().sum() for v in model.state_dict().values()]))
            batch = synthetic(B, 1000, C, demo)
            put(d, p + "x_checksum", torch.stack([batch[0].double().sum(), batch[0].double().abs().sum()]))
            if B == 4 and name == "cnn5":
                d["x_B4"] = batch[0].numpy()
            d[p + "y"] = batch[-1].numpy()
            if demo:
                d[p + "x_demo"] = batch[1].numpy()
            opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-4)
            loader = torch.utils.data.DataLoader(_DS(*batch), batch_size=B, shuffle=False)
            fn = train_one_epoch_demo if demo else train_one_epoch
            # logits of the first forward (train mode, before any update)
            model.train()
            with torch.no_grad():
                import copy
                m2 = copy.deepcopy(model)
                m2.train()
                out = m2(*batch[:-1])
                d[p + "logits0"] = (out[0] if isinstance(out, tuple) else out).numpy()
                d[p + "loss0"] = np.float64(F.binary_cross_entropy_with_logits(torch.from_numpy(d[p + "logits0"]), batch[-1]).item())
            for step in (1, 2, 3):
                ep_loss = fn(model, loader, opt, "cpu")
                d[p + f"epoch_loss{step}"] = np.float64(ep_loss)
                if step in (1, 3):
                    _snapshot(d, p + f"s{step}_", model, with_grads=True)
            d[p + "lr"] = np.float64(lr)
    save("g4_train_step", d)


def g5():
    d = {}
    for name, ctor, C, demo, lr in [("cnn5", lambda: ECGCNN(num_labels=5), 5, False, 1.5e-3),
                                   ("mm", lambda: ECGMultimodal(), 5, True, 1e-4)]:
        p = name + "_"
        set_seed(42)
        model = ctor()
        batch = synthetic(32, 1000, C, demo)
        model.train()
        world = 8
        acc = None
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        rank0_buffers = None
        for r in range(world):
            model.load_state_dict(sd0)          # every rank starts from identical replica
            model.zero_grad()
            shard = tuple(t[r * 4:(r + 1) * 4] for t in batch)
            out = model(*shard[:-1])
            loss = F.binary_cross_entropy_with_logits(out, shard[-1])
            loss.backward()
            g = [q.grad.clone() for q in model.parameters()]
            acc = g if acc is None else [a + b for a, b in zip(acc, g)]
            if r == 0:
                rank0_buffers = {k: v.clone() for k, v in model.state_dict().items()}
        # rank-0 buffers are what DistributedDataParallel(broadcast_buffers=True) keeps
        model.load_state_dict(rank0_buffers)
        for q, a in zip(model.parameters(), acc):
            q.grad = a / world
        for k, q in model.named_parameters():
            put(d, p + "avg_grad_" + k, q.grad)
        opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-4)
        opt.step()
        for k, v in model.state_dict().items():
            put(d, p + "post_sd_" + k, v)
        d[p + "lr"] = np.float64(lr)
    save("g5_ddp", d)


# ------------------------------------------------------------------ G6
def g6():
    d = {}
    ga = np.load(os.path.join(OUT, "g3_eval_known_answer.npz"))
    x = torch.from_numpy(ga["ecg"][:1])
    demo = torch.from_numpy(ga["demo"][:1]).requires_grad_(True)
    for name, ck, ctor in [("baseline", "outputs/ecg_baseline/ckpts/ecg_baseline_best.pth", lambda: ECGCNN(num_labels=5)),
                           ("multimodal", "outputs/ecg_multimodal/ckpts/ecg_multimodal_best.pth", lambda: ECGMultimodal())]:
        model = ctor()
        model.load_state_dict(_load_ckpt(os.path.join(REF, ck)))
        model.eval()
        last = (model.backbone if name == "baseline" else model.ecg_backbone.backbone)[-1].net[0]
        store = {}
        h1 = last.register_forward_hook(lambda m, i, o: store.__setitem__("act", o.detach().clone()))
        h2 = last.register_full_backward_hook(lambda m, gi, go: store.__setitem__("grad", go[0].detach().clone()))
        model.zero_grad()
        logits = model(x, demo) if name == "multimodal" else model(x)
        logits[:, 0].sum().backward()
        h1.remove(), h2.remove()
        A, G = store["act"][0], store["grad"][0]            # (256, 625)
        cam = torch.relu((G.mean(dim=1, keepdim=True) * A).sum(dim=0))   # (625,)
        cam_up = F.interpolate(cam[None, None], size=5000, mode="linear", align_corners=False)[0, 0]
        put(d, name + "_act", A)
        put(d, name + "_grad", G)
        d[name + "_cam"], d[name + "_cam_up"] = cam.numpy(), cam_up.numpy()
        d[name + "_logits"] = logits.detach().numpy()
        if name == "multimodal":
            d["multimodal_dlogit_ddemo"] = demo.grad.numpy().copy()
    save("g6_hooks", d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6"]
    for w in which:
        print("==", w)
        globals()[w]()
