"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUB_TARGET = 2048


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def sub(a):
    """Same strided subsample as tests/golden/make_golden.py::sub."""
    a = np.asarray(a).reshape(-1)
    return a[::max(1, a.size // SUB_TARGET)]


def to_np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def check_put(g, name, value, atol, rtol=0.0, what=""):
    """Compare `value` with a fixture entry written by make_golden.put()."""
    v = to_np(value)
    if name in g.files:
        np.testing.assert_allclose(v, g[name], atol=atol, rtol=rtol, err_msg=what or name)
    else:
        np.testing.assert_allclose(sub(v), g[name + "__sub"], atol=atol, rtol=rtol, err_msg=what or name)
        nrm = float(np.linalg.norm(v.astype(np.float64)))
        ref = float(g[name + "__norm"])
        assert abs(nrm - ref) <= 1e-4 * max(ref, 1e-12) + atol * np.sqrt(v.size), (name, nrm, ref)


def sd_from_npz(z, prefix=""):
    import torch
    return {k[len(prefix):]: torch.from_numpy(np.array(z[k])) for k in z.files
            if k.startswith(prefix) and not k.startswith("meta_")}


def paired(g, name, value):
    """(value, reference) as flat float64 arrays, subsampled like the fixture entry."""
    v = to_np(value).astype(np.float64)
    if name in g.files:
        return v.reshape(-1), np.asarray(g[name], dtype=np.float64).reshape(-1)
    return sub(v), np.asarray(g[name + "__sub"], dtype=np.float64)
