"""The drop-in modules on a GPU-less box: every reference script picks
`"cuda" if torch.cuda.is_available() else "cpu"` (scripts/03_train_ecg_baseline.py:120,
scripts/04_train_multimodal_prototype.py:137), so CPU tensors must run — through the stock torch layers the
ecg_hip.nn leaves inherit from (never through oracle/).  Golden fixture G4 (captured from the imported
reference) pins the numbers: this is "scripts/03-05 run unchanged" without a GPU."""
import numpy as np
import pytest
import torch

from util import check_put, golden, paired
from oracle import ref_models as R


def _ctors():
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    return {"cnn5": (lambda: ECGCNN(num_labels=5), 5, False), "cnn1": (lambda: ECGCNN(num_labels=1), 1, False),
            "mm": (lambda: ECGMultimodal(), 5, True)}


class _DS(torch.utils.data.Dataset):
    def __init__(self, *t):
        self.t = t

    def __len__(self):
        return self.t[0].shape[0]

    def __getitem__(self, i):
        return tuple(a[i] for a in self.t)


@pytest.mark.parametrize("name", ["cnn5", "cnn1", "mm"])
@pytest.mark.parametrize("optim", ["torch", "flat"])
def test_g4_train_steps_on_cpu_through_the_loop_api(name, optim):
    from ecg_hip.optim import FlatAdamW
    from src.training.loop import train_one_epoch
    from src.training.loop_demo import train_one_epoch_demo
    from src.utils.seed import set_seed
    g = golden("g4_train_step")
    ctor, C, demo = _ctors()[name]
    B, p = 4, f"{name}_B4_"
    lr = float(g[p + "lr"])
    set_seed(42)
    model = ctor()
    batch = R.synthetic_batch(B, 1000, C, demo=demo)
    loader = torch.utils.data.DataLoader(_DS(*batch), batch_size=B, shuffle=False)
    Opt = FlatAdamW if optim == "flat" else torch.optim.AdamW
    opt = Opt(model.parameters(), lr=lr, weight_decay=1e-4)
    fn = train_one_epoch_demo if demo else train_one_epoch
    for step in (1, 2, 3):
        ep = fn(model, loader, opt, "cpu")
        assert abs(ep - float(g[p + f"epoch_loss{step}"])) < (2e-6 if step == 1 else 1e-4), (step, ep)
        if step == 1:
            for k, prm in model.named_parameters():
                check_put(g, p + "s1_grad_" + k, prm.grad, atol=1e-6 if ".net.0.bias" in k else 2e-5)
        if step in (1, 3):
            for k, v in model.state_dict().items():
                q = p + f"s{step}_sd_" + k
                if k.endswith("num_batches_tracked"):
                    assert int(v.item()) == int(g[q]) == step
                    continue
                got, want = paired(g, q, v)
                # same ATen kernels as the reference: only thread-count summation noise, amplified by Adam on
                # noise-level gradients (conv biases under train-mode BN)
                assert np.abs(got - want).max() <= (2.02 * lr * step if ".net.0.bias" in k else 5e-5), k


def test_eval_loop_and_hooks_on_cpu():
    """eval_one_epoch + the Grad-CAM hook pattern (scripts/00_demo_inference.py:36-43) on CPU tensors."""
    from src.models.ecg_cnn import ECGCNN
    from src.training.loop import eval_one_epoch
    from src.utils.seed import set_seed
    set_seed(3)
    m = ECGCNN(num_labels=5)
    x, y = R.synthetic_batch(6, 256, 5)
    out = eval_one_epoch(m, torch.utils.data.DataLoader(_DS(x, y), batch_size=4), "cpu")
    assert set(out) == {"auroc_macro", "auprc_macro", "f1_macro", "bce_loss"} and np.isfinite(out["bce_loss"])
    last = [c for c in m.modules() if isinstance(c, torch.nn.Conv1d)][-1]
    seen = {}
    last.register_forward_hook(lambda mod, i, o: seen.__setitem__("act", o.detach()))
    last.register_full_backward_hook(lambda mod, gi, go: seen.__setitem__("grad", go[0].detach()))
    m.eval()
    m(x[:1])[:, 0].sum().backward()
    assert seen["act"].shape == seen["grad"].shape == (1, 256, 256 // 8)


def test_multimodal_cpu_matches_stock_restatement():
    from src.models.ecg_multimodal import ECGMultimodal
    from src.utils.seed import set_seed
    set_seed(42)
    m = ECGMultimodal().eval()
    R.seed_all(42)
    ref = R.RefECGMultimodal().eval()
    x, xd, _ = R.synthetic_batch(3, 500, 5, demo=True)
    with torch.no_grad():
        np.testing.assert_allclose(m(x, xd).numpy(), ref(x, xd).numpy(), atol=1e-6)
