"""torch.nn-compatible leaf modules backed by libecg_hip.so.

Every class subclasses the stock torch.nn layer the reference constructs, so
  * parameter/buffer names, shapes, dtypes and init RNG order are identical (state_dicts of
    the reference load with strict=True; `set_seed(42)` gives the same initial weights),
  * `isinstance(m, torch.nn.Conv1d)` discovery and module hooks used by the Grad-CAM scripts
    keep working (reference scripts/00_demo_inference.py:64-71, 36-37),
but `forward` dispatches to the HIP kernels whenever the input lives on a HIP device.

CPU tensors take the stock torch layer each class inherits from (`super().forward`): every
reference script selects `"cuda" if torch.cuda.is_available() else "cpu"`
(scripts/03_train_ecg_baseline.py:120), so on a GPU-less box the drop-in modules behave exactly
like the reference's.  That is stock ATen, not `oracle/`; a CUDA tensor with a missing
libecg_hip.so still fails loudly (ecg_hip/_lib.py).
"""
import torch
import torch.nn as nn

from . import functional as F_
from ._lib import EcgHipError


def _check_conv(m):
    if (m.stride != (1,) or m.dilation != (1,) or m.groups != 1 or m.padding_mode != "zeros"
            or isinstance(m.padding, str)):
        raise EcgHipError("HipConv1d supports stride=1, dilation=1, groups=1, zero integer padding "
                          "(the reference's Conv1d(k, padding=k//2), src/models/ecg_cnn.py:13)")


class HipConv1d(nn.Conv1d):
    def forward(self, x):
        if not x.is_cuda:
            return super().forward(x)
        _check_conv(self)
        return F_.Conv1dFn.apply(x, self.weight, self.bias, self.padding[0])


class HipBatchNorm1d(nn.BatchNorm1d):
    def forward(self, x):
        if not x.is_cuda:
            return super().forward(x)
        if not self.affine:
            raise EcgHipError("HipBatchNorm1d requires affine=True")
        if x.dim() != 3:
            raise EcgHipError(f"HipBatchNorm1d expects [N, C, L], got {tuple(x.shape)}")
        return F_.BatchNormFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var,
                                    self.num_batches_tracked, self.training, self.momentum, self.eps)


class HipReLU(nn.ReLU):
    """ReLU(inplace=True) in the reference; the HIP leaf writes a new tensor (same values)."""

    def forward(self, x):
        if not x.is_cuda:
            return super().forward(x)
        return F_.ReLUFn.apply(x)


class HipMaxPool1d(nn.MaxPool1d):
    def forward(self, x):
        if not x.is_cuda:
            return super().forward(x)
        k = self.kernel_size if isinstance(self.kernel_size, int) else self.kernel_size[0]
        s = self.stride if isinstance(self.stride, int) else self.stride[0]
        if k != 2 or s != 2 or self.padding not in (0, (0,)) or self.dilation not in (1, (1,)) \
                or self.ceil_mode or self.return_indices:
            raise EcgHipError("HipMaxPool1d supports MaxPool1d(2) only (src/models/ecg_cnn.py:16)")
        return F_.MaxPool2Fn.apply(x)


class HipAdaptiveAvgPool1d(nn.AdaptiveAvgPool1d):
    def forward(self, x):
        if not x.is_cuda:
            return super().forward(x)
        if self.output_size not in (1, (1,)):
            raise EcgHipError("HipAdaptiveAvgPool1d supports output_size=1 only (src/models/ecg_cnn.py:46)")
        return F_.GapFn.apply(x)


class HipLinear(nn.Linear):
    """nn.Linear; `fuse_relu=True` folds a following ReLU into the same launch."""

    fuse_relu = False

    def forward(self, x):
        if not x.is_cuda:
            y = super().forward(x)
            return torch.relu(y) if self.fuse_relu else y
        return F_.LinearFn.apply(x, self.weight, self.bias, self.fuse_relu)


class HipFusedReLU(nn.ReLU):
    """Placeholder that keeps the reference's Sequential indices (mlp.1, mlp.3) when the ReLU
    has been folded into the preceding HipLinear launch."""

    def forward(self, x):
        return x


def has_hooks(*modules):
    """True if any of the modules (or torch's global module hooks) would observe a call."""
    import torch.nn.modules.module as M
    if (M._global_forward_hooks or M._global_forward_pre_hooks or M._global_backward_hooks
            or getattr(M, "_global_backward_pre_hooks", None)):
        return True
    for m in modules:
        if (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks
                or getattr(m, "_backward_pre_hooks", None)):
            return True
    return False
