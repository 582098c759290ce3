"""CPU-side checks of the drop-in boundary: module tree, state_dict keys/dtypes,
seeded-init equivalence with the reference (via golden checksums), strict checkpoint loads."""
import numpy as np
import pytest
import torch

from util import golden, sd_from_npz
from oracle import ref_models as R


def _product():
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    return {"cnn5": lambda: ECGCNN(num_labels=5), "cnn1": lambda: ECGCNN(num_labels=1),
            "mm": lambda: ECGMultimodal()}


@pytest.mark.parametrize("name", ["cnn5", "cnn1", "mm"])
def test_seeded_init_matches_reference_checksums(name):
    from src.utils.seed import set_seed
    g = golden("g4_train_step")
    set_seed(42)
    m = _product()[name]()
    cs = torch.stack([v.double().sum() for v in m.state_dict().values()]).numpy()
    np.testing.assert_array_equal(cs, g[f"{name}_B4_init_checksum"])


def test_state_dict_contract_and_strict_ckpt_load():
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    for name, ctor, n in [("baseline", lambda: ECGCNN(num_labels=5), 32),
                          ("af", lambda: ECGCNN(num_labels=1), 32),
                          ("multimodal", lambda: ECGMultimodal(), 38)]:
        m = ctor()
        ck = sd_from_npz(golden("g3_ckpt_" + name))
        sd = m.state_dict()
        assert len(sd) == n and list(sd) == list(ck)
        for k in sd:
            assert sd[k].shape == ck[k].shape and sd[k].dtype == ck[k].dtype, k
        assert sd["backbone.0.net.1.num_batches_tracked" if name != "multimodal"
                  else "ecg_backbone.backbone.0.net.1.num_batches_tracked"].dtype == torch.int64
        m.load_state_dict(ck, strict=True)


def test_module_tree_addressing_used_by_gradcam_scripts():
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    m = ECGCNN()
    assert m.head.out_features == 3          # reference default num_labels=3
    last = m.backbone[-1].net[0]
    assert isinstance(last, torch.nn.Conv1d) and last.out_channels == 256 and last.kernel_size == (15,)
    assert isinstance(m.backbone[0].net[1], torch.nn.BatchNorm1d)
    assert isinstance(m.backbone[0].net[2], torch.nn.ReLU) and isinstance(m.backbone[0].net[3], torch.nn.MaxPool1d)
    found = [c for c in m.modules() if isinstance(c, torch.nn.Conv1d)]
    assert found[-1] is last
    mm = ECGMultimodal(ecg_feat_dim=128, demo_hidden_dim=32, some_unused_key=1)
    assert mm.ecg_backbone.proj.out_features == 128 and mm.film_gen.out_features == 256
    assert mm.demo_encoder.mlp[0].out_features == 64 and mm.demo_encoder.mlp[2].out_features == 32
    assert isinstance(mm.ecg_backbone.backbone[-1].net[0], torch.nn.Conv1d)
    # baseline checkpoint warm-start of the backbone alone (scripts/04:149-156)
    base = sd_from_npz(golden("g3_ckpt_baseline"))
    res = ECGMultimodal().ecg_backbone.load_state_dict(base, strict=False)
    assert sorted(res.unexpected_keys) == ["head.bias", "head.weight"] and not res.missing_keys


def test_hook_registration_switches_block_to_leaf_path():
    from ecg_hip import nn as hipnn
    from src.models.ecg_cnn import ECGCNN
    m = ECGCNN()
    blk = m.backbone[-1]
    assert not hipnn.has_hooks(blk.net, *blk.net)
    h = blk.net[0].register_forward_hook(lambda *a: None)
    assert hipnn.has_hooks(blk.net, *blk.net)
    h.remove()
    assert not hipnn.has_hooks(blk.net, *blk.net)
    h = blk.net[0].register_full_backward_hook(lambda *a: None)
    assert hipnn.has_hooks(blk.net, *blk.net)
    h.remove()


def test_metrics_and_seed_helpers():
    from src.training.metrics import compute_metrics
    from src.utils.seed import set_seed
    y = np.array([[1, 0], [0, 1], [1, 1], [0, 0]])
    p = np.array([[0.9, 0.2], [0.1, 0.8], [0.7, 0.6], [0.3, 0.4]])
    out = compute_metrics(y, p)
    assert out["auroc_macro"] == 1.0 and out["auprc_macro"] == 1.0 and out["f1_macro"] == 1.0
    out = compute_metrics(np.zeros((3, 2)), p[:3])          # single-class column -> NaN, not a raise
    assert np.isnan(out["auroc_macro"]) and out["f1_macro"] == 0.0
    set_seed(7); a = torch.rand(3)
    set_seed(7); b = torch.rand(3)
    assert torch.equal(a, b) and torch.backends.cudnn.deterministic and not torch.backends.cudnn.benchmark


def test_models_survive_deepcopy_and_pickle_with_pack_cache():
    """The per-model weight-pack cache holds raw device pointers; copying or pickling a model
    must drop it instead of failing or aliasing (torch.save(model) / copy.deepcopy(model))."""
    import copy
    import pickle
    from ecg_hip.functional import WeightPacker
    from src.models.ecg_multimodal import ECGMultimodal
    m = ECGMultimodal()
    m._packer._key = ("stale",)
    m._packer._tables = object()          # stands in for the ctypes pointer tables
    m2 = copy.deepcopy(m)
    assert isinstance(m2._packer, WeightPacker) and m2._packer._key is None
    m3 = pickle.loads(pickle.dumps(m))
    assert m3._packer._key is None
    assert list(m3.state_dict()) == list(m.state_dict())


def test_legacy_concat_model_matches_committed_checkpoint_layout():
    """§8(f)-4: ECGDemoConcat reproduces key names / shapes / dtypes of outputs/ecg_demo/ckpts/
    ecg_demo_best.pth (captured in g7_concat_ckpt_shapes.npz), i.e. it would load strict=True."""
    from src.models.ecg_demo_concat import ECGDemoConcat
    g = golden("g7_concat_ckpt_shapes")
    sd = ECGDemoConcat().state_dict()
    assert list(sd) == [str(k) for k in g["keys"]]
    for (k, v), shp, dt in zip(sd.items(), g["shapes"], g["dtypes"]):
        assert str(tuple(v.shape)) == str(shp) and str(v.dtype) == str(dt), k
