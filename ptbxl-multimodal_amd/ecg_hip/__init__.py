"""ecg_hip — Python host side of the MI355X-native ECG 1D-CNN path.

`ecg_hip.nn` holds torch.nn-compatible leaf modules whose forward/backward run the HIP
kernels of libecg_hip.so (include/ecg_hip.h) through `ecg_hip.functional`;
`ecg_hip.optim.FlatAdamW` and `ecg_hip.ddp.FlatGradDDP` are the optimizer and the
one-collective-per-step data-parallel wrapper.  There is no CPU fallback anywhere in this
package: CPU tensors or a missing shared library raise `EcgHipError`.
"""
from ._lib import EcgHipError, LIB_PATH, load  # noqa: F401

__all__ = ["EcgHipError", "LIB_PATH", "load"]
