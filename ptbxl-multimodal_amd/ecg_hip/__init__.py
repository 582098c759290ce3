"""ecg_hip — Python host side of the MI355X-native ECG 1D-CNN path.

`ecg_hip.nn` holds torch.nn-compatible leaf modules whose forward/backward run the HIP
kernels of libecg_hip.so (include/ecg_hip.h) through `ecg_hip.functional`;
`ecg_hip.optim.FlatAdamW` and `ecg_hip.ddp.FlatGradDDP` are the optimizer and the
one-collective-per-step data-parallel wrapper.  CUDA(HIP) tensors always take the HIP kernels
(a missing shared library raises `EcgHipError`, never a silent fallback); CPU tensors take the
stock torch layers the modules inherit from, as the reference does on a GPU-less box.
"""
from ._lib import EcgHipError, LIB_PATH, load  # noqa: F401

__all__ = ["EcgHipError", "LIB_PATH", "load"]
