"""MI355X-native counterpart of the reference's `src/models/ecg_multimodal.py`.

ECGBackbone (:19-41), DemoEncoder (:44-59) and the FiLM-conditioned ECGMultimodal (:67-99)
with identical constructor signatures, attribute names and state_dict keys.  The FiLM fusion
`(1 + tanh(gamma)) * z + beta` is one HIP launch each way (ecg_film_fwd / ecg_film_bwd) and
the two ReLUs of the demographic MLP are folded into their Linear launches.
"""
import torch
import torch.nn as nn

from ecg_hip import functional as hipF
from ecg_hip import nn as hipnn
from src.models.ecg_cnn import (BACKBONE_WIDTHS, ConvBlock, backbone_features, fully_fusable,  # noqa: F401
                                fused_forward, make_backbone)


class ECGBackbone(nn.Module):
    """[B, in_leads, T] -> [B, feat_dim]: the ECGCNN trunk without the classification head."""

    def __init__(self, in_leads: int = 12, feat_dim: int = 256):
        super().__init__()
        self.backbone = make_backbone(in_leads)
        self.gap = hipnn.HipAdaptiveAvgPool1d(1)
        self.proj = hipnn.HipLinear(BACKBONE_WIDTHS[-1], feat_dim)

    def features(self, x: torch.Tensor) -> torch.Tensor:
        """Globally pooled backbone activation [B, 256] (the input of `proj`)."""
        return backbone_features(self.backbone, self.gap, x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.proj(self.features(x))


class DemoEncoder(nn.Module):
    """MLP over [age_norm, sex_id, height_norm, weight_norm, pacemaker]; first layer is 64 wide."""

    def __init__(self, demo_dim: int = 5, hidden_dim: int = 64):
        super().__init__()
        fc1, fc2 = hipnn.HipLinear(demo_dim, 64), hipnn.HipLinear(64, hidden_dim)
        fc1.fuse_relu = fc2.fuse_relu = True     # ReLU runs inside the Linear launch
        self.mlp = nn.Sequential(fc1, hipnn.HipFusedReLU(inplace=True), fc2, hipnn.HipFusedReLU(inplace=True))

    def forward(self, x_demo: torch.Tensor) -> torch.Tensor:
        return self.mlp(x_demo)


class ECGMultimodal(nn.Module):
    """ECG embedding modulated feature-wise (FiLM) by the demographic embedding."""

    def __init__(self, in_leads: int = 12, feat_dim: int = 256, demo_dim: int = 5,
                 num_labels: int = 5, demo_hidden_dim: int = 64, ecg_feat_dim: int = None, **kwargs):
        super().__init__()
        if ecg_feat_dim is not None:          # config key `ecg_feat_dim` overrides feat_dim
            feat_dim = ecg_feat_dim
        self.ecg_backbone = ECGBackbone(in_leads=in_leads, feat_dim=feat_dim)
        self.demo_encoder = DemoEncoder(demo_dim=demo_dim, hidden_dim=demo_hidden_dim)
        self.film_gen = hipnn.HipLinear(demo_hidden_dim, 2 * feat_dim)
        self.head = hipnn.HipLinear(feat_dim, num_labels)
        self._packer = hipF.WeightPacker()

    def forward(self, x_ecg: torch.Tensor, x_demo: torch.Tensor) -> torch.Tensor:
        enc, bb = self.demo_encoder, self.ecg_backbone
        if x_ecg.is_cuda and x_ecg.dim() == 3 and fully_fusable(bb.backbone, bb.gap, bb, bb.proj, enc, enc.mlp, *enc.mlp,
                                              self.film_gen, self.head):
            # 4 fused blocks + fused tail, one weight-repack launch for the whole model
            logits, _ = fused_forward(self._packer, bb.backbone, bb.gap, x_ecg, x_demo, bb.proj, self.head,
                                      enc.mlp[0], enc.mlp[2], self.film_gen)
            return logits
        if x_ecg.is_cuda and not hipnn.has_hooks(bb, bb.proj, enc, enc.mlp, *enc.mlp, self.film_gen, self.head):
            # proj + demographic MLP + film_gen + FiLM + head: one fused launch
            logits, _ = hipF.tail(bb.features(x_ecg), x_demo, bb.proj, self.head, enc.mlp[0], enc.mlp[2],
                                  self.film_gen)
            return logits
        z = bb(x_ecg)
        film = self.film_gen(enc(x_demo))      # [B, 2F]: gamma-raw | beta
        return self.head(hipF.film(z, film))
