"""MI355X-native counterpart of the reference's `src/models/ecg_cnn.py`.

Same classes, constructor arguments, attribute names, module tree and state_dict keys as
the reference (ConvBlock: src/models/ecg_cnn.py:10-20, ECGCNN: :32-68), so checkpoints load
with strict=True and `scripts/03,05,06,08,11,13` run unchanged with this package first on
PYTHONPATH.  The leaves are `ecg_hip.nn` modules (subclasses of the stock torch.nn layers)
and a ConvBlock normally runs as ONE fused autograd node — conv + BN-statistics epilogue,
finalize, BN-apply+ReLU+MaxPool — instead of four ATen dispatches.  When somebody has
hooked an inner module (Grad-CAM on `backbone[-1].net[0]`), the block falls back to calling
the four HIP leaves one by one so hooks observe the same tensors as in the reference.
"""
import torch
import torch.nn as nn

from ecg_hip import functional as hipF
from ecg_hip import nn as hipnn

BACKBONE_WIDTHS = (32, 64, 128, 256)


class ConvBlock(nn.Module):
    """Conv1d(k, padding=k//2) -> BatchNorm1d -> ReLU -> MaxPool1d(p); `.net` is the
    indexable [conv, bn, relu, pool] sequence the Grad-CAM scripts address."""

    def __init__(self, in_ch: int, out_ch: int, k: int = 15, p: int = 2):
        super().__init__()
        self.net = nn.Sequential(
            hipnn.HipConv1d(in_ch, out_ch, kernel_size=k, padding=k // 2),
            hipnn.HipBatchNorm1d(out_ch),
            hipnn.HipReLU(inplace=True),
            hipnn.HipMaxPool1d(kernel_size=p),
        )
        self._fusable = (p == 2)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        conv, bn, relu, pool = self.net[0], self.net[1], self.net[2], self.net[3]
        # CPU tensors: the stock torch layers the leaves inherit from (no GPU on this box)
        if x.is_cuda and self._fusable and not hipnn.has_hooks(self.net, conv, bn, relu, pool):
            return hipF.conv_block(x, conv, bn)
        return self.net(x)


def make_backbone(in_leads: int) -> nn.Sequential:
    widths = (in_leads,) + BACKBONE_WIDTHS
    return nn.Sequential(*(ConvBlock(a, b) for a, b in zip(widths, widths[1:])))


def fully_fusable(backbone: nn.Sequential, gap: nn.Module, *tail_modules) -> bool:
    """True when nobody hooks any module of the path and every block is a stock ConvBlock: the whole
    forward then runs as fused launches with ONE weight-repack launch up front."""
    blocks = list(backbone)
    mods = [backbone, gap, *tail_modules]
    for blk in blocks:
        if not (isinstance(blk, ConvBlock) and blk._fusable):
            return False
        mods += [blk, blk.net, *blk.net]
    return not hipnn.has_hooks(*mods)


def fused_forward(packer, backbone, gap, x, x_demo, proj, head, mlp0=None, mlp2=None, film_gen=None):
    """(logits, z): 4 fused ConvBlocks (the last one with the global average pool folded in) and
    the fused tail, with all weight repacking done by one grouped launch."""
    blocks = list(backbone)
    linears = [proj] if x_demo is None else [proj, film_gen]
    packs, transposed = packer.pack([b.net[0] for b in blocks], linears, torch.is_grad_enabled())
    carry = None           # mixed precision only: the true row length of a bf16 activation handed from block to block
    for i, blk in enumerate(blocks):
        last = i == len(blocks) - 1
        x, carry = hipF.conv_block_chain(x, blk.net[0], blk.net[1], gap=last, packed=packs[i], carry=carry,
                                      next_conv=None if last else blocks[i + 1].net[0],
                                      next_bn=None if last else blocks[i + 1].net[1])
    return hipF.tail(x, x_demo, proj, head, mlp0, mlp2, film_gen, transposed=transposed)


def backbone_features(backbone: nn.Sequential, gap: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """`gap(backbone(x)).squeeze(-1)` -> [B, C].  When nobody hooks the containers, the last
    ConvBlock runs with the global average pool folded into its BN+ReLU+pool kernel, so the
    last pooled activation is never written to HBM."""
    blocks = list(backbone)
    last = blocks[-1]
    if (x.is_cuda and isinstance(last, ConvBlock) and last._fusable and x.dim() == 3
            and not hipnn.has_hooks(backbone, gap, *blocks, last.net, *last.net)):
        for blk in blocks[:-1]:
            x = blk(x)
        return hipF.conv_block(x, last.net[0], last.net[1], gap=True)
    return gap(backbone(x)).squeeze(-1)


class ECGCNN(nn.Module):
    """12-lead ECG classifier: 4 ConvBlocks -> global average pool -> proj -> head.

    Args mirror the reference: in_leads (12), feat_dim (latent size), num_labels.
    """

    def __init__(self, in_leads: int = 12, feat_dim: int = 256, num_labels: int = 3):
        super().__init__()
        self.backbone = make_backbone(in_leads)
        self.gap = hipnn.HipAdaptiveAvgPool1d(1)
        self.proj = hipnn.HipLinear(BACKBONE_WIDTHS[-1], feat_dim)
        self.head = hipnn.HipLinear(feat_dim, num_labels)
        self._packer = hipF.WeightPacker()

    def forward(self, x: torch.Tensor, return_features: bool = False):
        """x: [B, in_leads, T] -> logits [B, num_labels] (or (logits, z) if return_features)."""
        if x.is_cuda and x.dim() == 3 and fully_fusable(self.backbone, self.gap, self.proj, self.head):
            logits, z = fused_forward(self._packer, self.backbone, self.gap, x, None, self.proj, self.head)
            return (logits, z) if return_features else logits
        pooled = backbone_features(self.backbone, self.gap, x)
        if pooled.is_cuda and not hipnn.has_hooks(self.proj, self.head):
            logits, z = hipF.tail(pooled, None, self.proj, self.head)       # one fused launch
        else:
            z = self.proj(pooled)
            logits = self.head(z)
        return (logits, z) if return_features else logits
