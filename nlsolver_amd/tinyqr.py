"""Host-side mirror of the reference's tinyqr::lm for batches of systems on the device.

Reference interface (tinyqr.h:461-470):
    std::vector<T> lm(const std::vector<T>& X, const std::vector<T>& y, T tol = 1e-12)
with X column-major n x p (n = y.size(), p = X.size() / n), one system per call. Here lm() takes
one system — X of shape (p, n), i.e. the same memory — or a batch (batch, p, n) with y (batch, n),
solved side by side on the GPU (n >= p, p <= 64). There is no CPU fallback.
qr_decomposition() and lm(..., reference_order=True) run the reference's own order of operations
on the device (nlsg_tinyqr_qr): Q, R and beta equal the reference's bit for bit, any p.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib


def _reference_order(X, y, tol, device, want_q, want_r, want_beta):
    """nlsg_tinyqr_qr: the reference's own order of operations on the device (a parity mode)."""
    batch, p, n = X.shape
    Q = np.empty((batch, p, n)) if want_q else None
    R = np.empty((batch, p, p)) if want_r else None
    beta = np.empty((batch, p)) if want_beta else None
    ptr = lambda a: a.ctypes.data_as(_capi.pd) if a is not None else None  # noqa: E731
    check(lib().nlsg_tinyqr_qr(ptr(X), ptr(y), batch, n, p, float(tol), device, ptr(Q), ptr(R), ptr(beta)))
    return Q, R, beta


def qr_decomposition(X, tol=1e-8, *, device=0):
    """tinyqr::qr_decomposition(X, n, p, tol) (tinyqr.h:291-310) per system on the device, in the
    reference's order of operations: returns (Q, R) with Q[..., i, j] = the reference's Q[i * n + j]
    (the thin Q, p rows of n) and R[..., j, i] = its R[j * p + i] = R(i, j) — bit for bit.
    X: (p, n) or (batch, p, n), the column-major n x p systems; n >= p, n + p <= 1280."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    single = X.ndim == 2
    if single:
        X = X[None]
    if X.ndim != 3:
        raise TypeError("X must be (p, n) or (batch, p, n) [column-major n x p systems]")
    Q, R, _ = _reference_order(X, None, tol, device, True, True, False)
    return (Q[0], R[0]) if single else (Q, R)


def lm(X, y, tol=1e-12, *, device=0, return_ms=False, reference_order=False):
    """Least-squares coefficients beta (p,) or (batch, p): tinyqr::lm(X, y, tol) per system.
    X[..., j, i] is element (i, j) of the system's n x p matrix (column-major, as tinyqr takes it).
    reference_order=True: the parity mode — Q formed, the reference's summation orders, beta equal to
    the reference's bit for bit (slower: one wave per system, serial rotations)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    single = X.ndim == 2
    if single:
        X, y = X[None], y[None]
    if X.ndim != 3 or y.ndim != 2 or y.shape != (X.shape[0], X.shape[2]):
        raise TypeError("X must be (batch, p, n) [column-major n x p systems] and y (batch, n)")
    batch, p, n = X.shape
    if reference_order:
        beta = _reference_order(X, y, tol, device, False, False, True)[2]
        return beta[0] if single else beta
    beta = np.empty((batch, p))
    ms = C.c_float()
    check(lib().nlsg_tinyqr_lm(X.ctypes.data_as(_capi.pd), y.ctypes.data_as(_capi.pd), batch, n, p,
                               float(tol), device, beta.ctypes.data_as(_capi.pd), C.byref(ms)))
    out = beta[0] if single else beta
    return (out, ms.value) if return_ms else out
