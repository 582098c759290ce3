"""ctypes binding of libecg_hip.so (include/ecg_hip.h).

There is no CPU fallback at this level: if the shared library is missing, or a tensor handed to a
kernel wrapper is not a contiguous float32 CUDA tensor, the call raises.  (CPU tensors are routed to
stock torch one level up, in ecg_hip/nn.py and the model classes — never to oracle/.)  Every wrapper
launches on torch's current HIP stream and never synchronises.
"""
import ctypes
import functools
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ECG_HIP_LIB: load another build of the same ABI instead (the diagnostic `make STAMP=1` library of the profiling tools)
LIB_PATH = os.environ.get("ECG_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libecg_hip.so")

_vp, _i, _f, _sz, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_longlong

# name -> (restype, argtypes); mirrors include/ecg_hip.h one to one
SIGNATURES = {
    "ecg_version": (_i, []),
    "ecg_last_error": (ctypes.c_char_p, []),
    "ecg_check_device": (_i, []),
    "ecg_conv1d_pack_weights": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "ecg_pack_weights_grouped": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "ecg_bn_relu_pool_bwd_one_launch_splits": (_i, [_i, _i, _i, _i]),
    "ecg_bn_relu_pool_bwd_one_launch_counter_uints": (_sz, [_i, _i, _i, _i]),
    "ecg_bn_relu_pool_bwd_one_launch": (_i, [_vp] * 7 + [_i] + [_vp] * 3 + [_i] * 6 + [_vp]),
    "ecg_pack_weights_grouped_mixed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "ecg_conv1d_fwd_stat_partials": (_i, [_i, _i, _i, _i, _i, _i]),
    "ecg_conv1d_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ecg_conv1d_bwd_data": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ecg_conv1d_bwd_weight_ws_floats": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ecg_conv1d_bwd_weight_bias": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ecg_conv1d_dy_row_stride": (_i, [_i] * 7),
    "ecg_conv1d_multiplies_per_output_pair": (_i, [_i] * 5),
    "ecg_conv1d_bwd_data_ld": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ecg_conv1d_bwd_weight_bias_ld": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ecg_conv1d_bf16_supported": (_i, [_i, _i, _i, _i]),
    "ecg_conv1d_bf16_packed_elems": (_sz, [_i, _i, _i]),
    "ecg_conv1d_pack_weights_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "ecg_conv1d_fwd_bf16_yh": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp] + [_i] * 6 + [_vp]),
    "ecg_conv1d_fwd_bf16_yh_stat_partials": (_i, [_i] * 9),
    "ecg_conv1d_bf16_ring_tile": (_i, [_i] * 8),
    "ecg_bn_stats_relu_pool_fwd_h": (_i, [_vp, _i, _ll, _vp, _vp, _vp, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ecg_bn_relu_pool_bwd_h": (_i, [_vp, _i, _vp, _i, _i] + [_vp] * 5 + [_i] + [_vp] * 3 + [_i] * 4 + [_vp]),
    "ecg_conv1d_bwd_data_bf16hh": (_i, [_vp, _i, _vp, _vp, _i] + [_i] * 6 + [_vp]),
    "ecg_conv1d_bf16_tk_supported": (_i, [_i] * 4),
    "ecg_conv1d_bf16_tk_dy_stride": (_i, [_i]),
    "ecg_conv1d_bwd_weight_bf16_ncl_ws_floats": (_sz, [_i] * 6),
    "ecg_conv1d_bwd_weight_bias_bf16_ncl": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp] + [_i] * 6 + [_vp]),
    "ecg_bn_stat_partials_count": (_i, [_i, _i, _i]),
    "ecg_bn_stat_partials": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ecg_bn_finalize": (_i, [_vp, _i, _ll, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _vp]),
    "ecg_bn_invstd": (_i, [_vp, _vp, _i, _f, _vp]),
    "ecg_bn_relu_pool_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ecg_bn_stats_relu_pool_fwd": (_i, [_vp, _i, _ll, _vp, _vp, _vp, _f, _f] + [_vp] * 6 + [_i] * 4 + [_vp]),
    "ecg_bn_stats_relu_pool_gap_fwd_yh": (_i, [_vp, _i, _ll, _vp, _vp, _vp, _f, _f, _vp, _i] + [_vp] * 5 + [_i] * 3 + [_vp]),
    "ecg_bn_relu_pool_bwd_ws_floats": (_sz, [_i, _i, _i]),
    "ecg_bn_relu_pool_bwd": (_i, [_vp] * 10 + [_i, _i, _i, _i, _vp]),
    "ecg_bn_relu_pool_bwd_ld": (_i, [_vp] * 7 + [_i] + [_vp] * 3 + [_i, _i, _i, _i, _vp]),
    "ecg_conv1d_bn_relu_pool_eval_supported": (_i, [_i, _i, _i, _i]),
    "ecg_conv1d_bn_relu_pool_eval_fwd": (_i, [_vp] * 7 + [_f, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ecg_conv1d_bn_relu_pool_gap_eval_supported": (_i, [_i] * 5),
    "ecg_conv1d_bn_relu_pool_gap_eval_fwd": (_i, [_vp] * 7 + [_f, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ecg_bn_relu_pool_gap_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ecg_bn_relu_pool_gap_bwd": (_i, [_vp] * 10 + [_i, _i, _i, _i, _vp]),
    "ecg_bn_relu_pool_gap_bwd_ld": (_i, [_vp] * 7 + [_i] + [_vp] * 3 + [_i, _i, _i, _i, _vp]),
    "ecg_bn_apply_fwd": (_i, [_vp] * 6 + [_i, _i, _i, _vp]),
    "ecg_bn_bwd_ws_floats": (_sz, [_i, _i, _i]),
    "ecg_bn_bwd": (_i, [_vp] * 9 + [_i, _i, _i, _i, _vp]),
    "ecg_relu_fwd": (_i, [_vp, _vp, _sz, _vp]),
    "ecg_relu_bwd": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "ecg_maxpool2_fwd": (_i, [_vp, _vp, _i, _i, _vp]),
    "ecg_maxpool2_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "ecg_gap_fwd": (_i, [_vp, _vp, _i, _i, _vp]),
    "ecg_gap_bwd": (_i, [_vp, _vp, _i, _i, _vp]),
    "ecg_linear_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ecg_linear_bwd_ws_floats": (_sz, [_i, _i, _i]),
    "ecg_linear_bwd": (_i, [_vp] * 8 + [_i, _i, _i, _i, _vp]),
    "ecg_film_fwd": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "ecg_film_bwd": (_i, [_vp] * 5 + [_i, _i, _vp]),
    "ecg_bce_logits_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, ctypes.c_double, _vp]),
    "ecg_sigmoid_fwd": (_i, [_vp, _vp, _sz, _vp]),
    "ecg_transpose": (_i, [_vp, _vp, _i, _i, _vp]),
    "ecg_tail_fwd": (_i, [_vp] * 18 + [_i] * 7 + [_vp]),
    "ecg_tail_bwd_chain": (_i, [_vp] * 18 + [_i] * 8 + [_vp]),
    "ecg_linear_wgrad_grouped": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ecg_adamw_step": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _f, _f, _f, _f, _f, _f, _vp]),
    "ecg_adamw_step_graph": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _f, _f, _f, _f, _f, _f, _vp]),
    "ecg_wfdb16_physical": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ecg_wfdb16_zscore": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ecg_zscore_rows": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "ecg_host_gather_rows": (_i, [_vp, _sz, _vp, _i, _vp]),
}

_lock = threading.Lock()
_lib = None


class EcgHipError(RuntimeError):
    pass


def load():
    """dlopen libecg_hip.so and bind every symbol of the header.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise EcgHipError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C ptbxl-multimodal_amd/csrc`). CUDA tensors have no fallback path.")
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


def last_error():
    return load().ecg_last_error().decode("utf-8", "replace")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """hipStream_t of torch's current stream on the current device.  Called once per launch (~45 times a
    step): the raw accessors cost ~0.3 us, torch.cuda.current_stream() ~8 us (it builds a Stream object)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor that satisfies the ABI contract (or None)."""
    if t is None:
        return None
    if t.is_cuda and t.is_contiguous():
        return t.data_ptr()
    if not t.is_cuda:
        raise EcgHipError("a CPU tensor reached a HIP kernel wrapper: ecg_hip.functional's Functions take CUDA(HIP) "
                          "tensors only (the nn modules route CPU inputs to stock torch before getting here)")
    raise EcgHipError("ecg_hip kernels need contiguous tensors")


_F32 = torch.float32


def f32(t):
    if t is None:
        return None
    if t.dtype is _F32 and t.is_cuda and t.is_contiguous():
        return t.data_ptr()
    if t.dtype != torch.float32:
        raise EcgHipError(f"ecg_hip kernels compute in float32; got {t.dtype}")
    return ptr(t)


_events = None     # when a list: (name, int-args, start, end) per launch — see kernel_timing()


def call(name, *args):
    """Invoke an int-returning entry point; raise with ecg_last_error() on failure."""
    lib = load()
    if _events is None:
        rc = getattr(lib, name)(*args)
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        _events.append((name, tuple(a for a in args[:-1] if isinstance(a, int) and abs(a) < (1 << 31)), e0, e1))
    if rc != 0:
        raise EcgHipError(f"{name} failed (code {rc}): {last_error()}")


class kernel_timing:
    """Context manager: brackets every ABI launch with HIP events on the launch stream and
    returns {(entry point, shape ints): [ms, ...]} — the live per-kernel durations bench.py
    reports against the roofline.  Adds two event records per launch, so it is used in a
    separate pass, never inside the timed throughput region."""

    def __enter__(self):
        global _events
        _events = []
        self.result = {}
        return self

    def __exit__(self, *exc):
        global _events
        ev, _events = _events, None
        torch.cuda.synchronize()
        for name, sig, e0, e1 in ev:
            self.result.setdefault((name, sig), []).append(e0.elapsed_time(e1))
        return False


def ptr_table(tensors, any_dtype=False):
    """Host array of device pointers (NULL for None) for the grouped entry points."""
    arr = (ctypes.c_void_p * len(tensors))(*[(ptr if any_dtype else f32)(t) for t in tensors])
    return arr


def int_table(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


@functools.lru_cache(maxsize=4096)
def query(name, *args):
    """Invoke a size/count helper (no error channel).  They are pure functions of their integer arguments,
    so the answers are memoised: a train step asks ~30 of them with the same shapes every time."""
    return getattr(load(), name)(*args)
