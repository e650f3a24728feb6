"""Host-side mirror of the reference's tinyqr::lm for batches of systems on the device.

Reference interface (tinyqr.h:461-470):
    std::vector<T> lm(const std::vector<T>& X, const std::vector<T>& y, T tol = 1e-12)
with X column-major n x p (n = y.size(), p = X.size() / n), one system per call. Here lm() takes
one system — X of shape (p, n), i.e. the same memory — or a batch (batch, p, n) with y (batch, n),
solved side by side on the GPU (n >= p, p <= 64). There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib


def lm(X, y, tol=1e-12, *, device=0, return_ms=False):
    """Least-squares coefficients beta (p,) or (batch, p): tinyqr::lm(X, y, tol) per system.
    X[..., j, i] is element (i, j) of the system's n x p matrix (column-major, as tinyqr takes it)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    single = X.ndim == 2
    if single:
        X, y = X[None], y[None]
    if X.ndim != 3 or y.ndim != 2 or y.shape != (X.shape[0], X.shape[2]):
        raise TypeError("X must be (batch, p, n) [column-major n x p systems] and y (batch, n)")
    batch, p, n = X.shape
    beta = np.empty((batch, p))
    ms = C.c_float()
    check(lib().nlsg_tinyqr_lm(X.ctypes.data_as(_capi.pd), y.ctypes.data_as(_capi.pd), batch, n, p,
                               float(tol), device, beta.ctypes.data_as(_capi.pd), C.byref(ms)))
    out = beta[0] if single else beta
    return (out, ms.value) if return_ms else out
