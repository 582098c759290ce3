"""CPU-side checks of the C-ABI boundary: libecg_hip.so loads, exports every symbol that
include/ecg_hip.h declares, and validates arguments on the host before any launch."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from ecg_hip import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib.load()


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "ecg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ecg_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from ecg_hip import _lib
    declared = _header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ecg_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"


def test_version_and_error_channel(lib):
    assert lib.ecg_version() == 100
    rc = lib.ecg_conv1d_fwd(None, None, None, None, None, 1, 12, 32, 100, 40, 7, None)
    assert rc == 1      # ECG_EINVAL: kernel size outside [1,31]; rejected before any launch
    assert b"kernel size" in lib.ecg_last_error()
    rc = lib.ecg_conv1d_fwd(None, None, None, None, None, 1, 12, 32, 100, 15, 7, None)
    assert rc == 1 and b"null pointer" in lib.ecg_last_error()
    rc = lib.ecg_conv1d_fwd(None, None, None, None, None, 1, 12, 32, 3, 15, 2, None)
    assert rc == 1 and b"empty output" in lib.ecg_last_error()
    rc = lib.ecg_adamw_step(None, None, None, None, 16, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, None)
    assert rc == 1


def test_size_helpers_are_pure(lib):
    assert lib.ecg_conv1d_fwd_stat_partials(256, 12, 32, 1000, 15, 7) > 0
    assert lib.ecg_conv1d_bwd_weight_ws_floats(256, 12, 32, 1000, 15, 7) >= 32 * 12 * 15 + 32
    assert lib.ecg_bn_relu_pool_bwd_ws_floats(256, 32, 1000) >= 32 * 2
    assert lib.ecg_bn_stat_partials_count(4, 32, 100) >= 1


def test_kernel_wrappers_refuse_cpu_tensors_but_modules_route_them_to_stock_torch():
    """The ABI wrappers never see a CPU tensor: the nn modules send those to the stock torch layer they inherit
    from (tests/test_cpu_path.py checks the numbers); a CPU tensor handed to a Function directly still raises."""
    import torch
    from ecg_hip import EcgHipError, functional as hipF
    from src.models.ecg_cnn import ECGCNN
    assert ECGCNN(num_labels=5)(torch.randn(2, 12, 64)).shape == (2, 5)
    with pytest.raises(EcgHipError, match="CPU tensor"):
        hipF.ReLUFn.apply(torch.zeros(4))
    with pytest.raises(EcgHipError, match="CPU tensor"):
        hipF.BceWithLogitsFn.apply(torch.zeros(2, 5), torch.zeros(2, 5), None, 1.0)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ecg_hip import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.EcgHipError, match="missing"):
        _lib.load()


def test_one_launch_batchnorm_backward_is_decided_per_call_from_the_parameters(monkeypatch):
    """No process-wide switch in the library any more: ConvBlockFn.backward picks the one-launch or the two-pass entry
    point per call.  Parameters declared "busy" (hooks issue all-reduces under backward) take two passes, "quiet" ones the
    one-launch form; undeclared parameters are quiet in a single-rank process and busy as soon as more than one rank
    exists (stock DistributedDataParallel users get the safe form without asking)."""
    import torch
    from ecg_hip import functional as F
    monkeypatch.setattr(F, "_bn_one_launch", True)
    monkeypatch.setattr(F, "_backward_collectives", {})
    monkeypatch.setattr(F, "_collectives_by_ptr", None)
    a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(3))
    ka, kb = a.data_ptr(), b.data_ptr()
    assert F.bn_backward_one_launch_allowed(ka) and F.bn_backward_one_launch_allowed(kb)
    F.declare_backward_collectives([a], True)
    assert not F.bn_backward_one_launch_allowed(ka) and F.bn_backward_one_launch_allowed(kb)      # per parameter, not global
    F.declare_backward_collectives([a], False)
    assert F.bn_backward_one_launch_allowed(ka)
    monkeypatch.setattr(F, "_multi_rank", lambda: True)
    assert F.bn_backward_one_launch_allowed(ka) and not F.bn_backward_one_launch_allowed(kb)      # undeclared + multi-rank
    F.declare_backward_collectives([a], None)
    assert not F.bn_backward_one_launch_allowed(ka)
    monkeypatch.setattr(F, "_multi_rank", lambda: False)
    assert F.set_bn_backward_one_launch(False) is True
    assert not F.bn_backward_one_launch_allowed(ka) and not F.bn_backward_one_launch_allowed(kb)
    assert F.set_bn_backward_one_launch(True) is False
    # statements follow the parameter OBJECT: an optimizer that re-homes the storage afterwards (FlatAdamW, the adopted stock
    # AdamW) does not leave a wrapper's statement on a dead address, and an owner only withdraws what it said itself
    from ecg_hip.optim import flatten_tensors_
    monkeypatch.setattr(F, "_multi_rank", lambda: True)
    F.declare_backward_collectives([a, b], False, owner="wrapper")
    old = (a.data_ptr(), b.data_ptr())
    flat = flatten_tensors_([a, b])
    F.parameters_rehomed()               # (flatten_tensors_ calls it for device tensors)
    assert a.data_ptr() == flat.data_ptr() and a.data_ptr() != old[0]
    assert F.bn_backward_one_launch_allowed(a.data_ptr()) and F.bn_backward_one_launch_allowed(b.data_ptr())
    assert not F.bn_backward_one_launch_allowed(old[1])           # the abandoned address is undeclared again
    F.declare_backward_collectives([a, b], None, owner="optimizer")   # somebody else's withdrawal: no effect
    assert F.bn_backward_one_launch_allowed(a.data_ptr())
    F.declare_backward_collectives([id(a), id(b)], None, "wrapper")   # the finalizer's form: ids + the owner's token
    assert not F.bn_backward_one_launch_allowed(a.data_ptr())


def test_library_reads_no_environment_and_allocates_nothing():
    """include/ecg_hip.h: "never allocates", "no global mutable state", "reads no environment variable" — checked on the
    dynamic symbol table of the built library (what it would have to import to break the promise)."""
    import subprocess
    from ecg_hip import _lib as L
    if not os.path.exists(L.LIB_PATH):
        pytest.skip("library not built")
    syms = subprocess.run(["nm", "-D", "--undefined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for banned in ("getenv", "secure_getenv", "hipMalloc", "hipFree", "hipMemset", "hipMallocAsync", "hipHostMalloc",
                   "pthread_mutex_lock"):
        assert not any(line.split()[-1].split("@")[0] == banned for line in syms.splitlines() if line.strip()), banned
