"""Legacy late-fusion-by-concatenation model (the literal "concat" of BASELINE.json.north_star).

Its source file is NOT in the reference tree; only its weights survive in
`outputs/ecg_demo/ckpts/ecg_demo_best.pth`.  The module tree below is
reconstructed from that checkpoint's keys and shapes:

    ecg_encoder      = ECGCNN(num_labels=5)              (z = proj output, [B, 256])
    demo_encoder.net = Linear(5, 32) -> ReLU -> Linear(32, 64) [-> ReLU]
    classifier       = Linear(320, 256) -> ReLU -> Dropout -> Linear(256, 5)   on cat([z, d])

so the checkpoint loads with strict=True.  Parity is pinned by shapes only: no committed
prediction file matches this checkpoint (the `ecg_demo` CSV was produced by the FiLM model), and
whether a final ReLU followed `demo_encoder.net.2` cannot be recovered (`demo_relu` flag).
The concatenation is a torch.cat of two [B, <=256] tensors; everything else runs on the HIP leaves.
"""
import torch
import torch.nn as nn

from ecg_hip import nn as hipnn
from src.models.ecg_cnn import ECGCNN


class _DemoNet(nn.Module):
    def __init__(self, demo_dim, demo_relu):
        super().__init__()
        fc1, fc2 = hipnn.HipLinear(demo_dim, 32), hipnn.HipLinear(32, 64)
        fc1.fuse_relu = True
        fc2.fuse_relu = bool(demo_relu)
        layers = [fc1, hipnn.HipFusedReLU(inplace=True), fc2]
        if demo_relu:
            layers.append(hipnn.HipFusedReLU(inplace=True))
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return self.net(x)


class ECGDemoConcat(nn.Module):
    def __init__(self, in_leads: int = 12, feat_dim: int = 256, demo_dim: int = 5, num_labels: int = 5,
                 dropout: float = 0.3, demo_relu: bool = True):
        super().__init__()
        self.ecg_encoder = ECGCNN(in_leads=in_leads, feat_dim=feat_dim, num_labels=num_labels)
        self.demo_encoder = _DemoNet(demo_dim, demo_relu)
        hidden = hipnn.HipLinear(feat_dim + 64, 256)
        hidden.fuse_relu = True
        self.classifier = nn.Sequential(hidden, hipnn.HipFusedReLU(inplace=True), nn.Dropout(dropout),
                                        hipnn.HipLinear(256, num_labels))

    def forward(self, x_ecg: torch.Tensor, x_demo: torch.Tensor) -> torch.Tensor:
        _, z = self.ecg_encoder(x_ecg, return_features=True)
        fused = torch.cat([z, self.demo_encoder(x_demo)], dim=1)        # late fusion by concatenation
        return self.classifier(fused)
