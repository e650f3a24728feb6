"""Host-side mirror of the reference's NelderMead class for device objectives, batched.

Reference interface (nlsolver.h:2099-2165):
    NelderMead<Callable, scalar_t>(f, step = -1, alpha = 1, gamma = 2, rho = 0.5, sigma = 0.5,
        eps = 1e-6, max_iter = 500, no_change_best_tol = 20, restarts = 0)
    minimize(x) / maximize(x) / minimize(x, upper, lower) / maximize(x, upper, lower)
(upper before lower — the opposite of PSO). x may be (n,) or (batch, n): independent starts,
one simplex per GPU workgroup.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import NMConfig, Status, check, lib


class NMEngine:
    """reference_order=True (NLSG_NM_REFERENCE_ORDER): the objective's terms and std_err's sums in index
    order, as the reference adds them — its own runs bit for bit at every dimension (the lane-tree sums
    can break a tie between two vertices the other way: same algorithm, another branch)."""

    def __init__(self, objective, batch, dim, *, minimize=True, bounded=False, step=-1.0, alpha=1.0,
                 gamma=2.0, rho=0.5, sigma=0.5, eps=1e-6, max_iter=500, no_change_best_tol=20,
                 restarts=0, device=0, stream=None, reference_order=False):
        cfg = NMConfig()
        cfg.flags = _capi.NM_REFERENCE_ORDER if reference_order else 0
        cfg.struct_size = C.sizeof(NMConfig)
        cfg.device = device
        cfg.stream = None if stream is None else (stream or 1)
        from .de import CustomObjective, rtc_library_path
        custom = objective if isinstance(objective, CustomObjective) else None
        cfg.objective = (_capi.OBJ_CUSTOM if custom else
                         _capi.OBJECTIVES[objective] if isinstance(objective, str) else objective)
        cfg.minimize, cfg.bounded = int(bool(minimize)), int(bool(bounded))
        cfg.batch, cfg.dim = batch, dim
        cfg.step, cfg.alpha, cfg.gamma, cfg.rho, cfg.sigma, cfg.eps = step, alpha, gamma, rho, sigma, eps
        cfg.max_iter, cfg.no_change_best_tol, cfg.restarts = max_iter, no_change_best_tol, restarts
        self.cfg = cfg
        self._h = C.c_void_p()
        if custom:
            check(lib().nlsg_rtc_load(rtc_library_path().encode()))
            obj = _capi.CustomObjectiveC(custom.term_body.encode(), custom.finish_body.encode(),
                                         int(custom.chain), 0)
            check(lib().nlsg_nm_create_custom(C.byref(cfg), C.byref(obj), C.byref(self._h)))
        else:
            check(lib().nlsg_nm_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().nlsg_nm_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def minimize(self, x, upper=None, lower=None):
        B, n = self.cfg.batch, self.cfg.dim
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (B, n)
        up = lo = None
        if self.cfg.bounded:
            up = np.ascontiguousarray(np.broadcast_to(upper, (n,)), dtype=np.float64)
            lo = np.ascontiguousarray(np.broadcast_to(lower, (n,)), dtype=np.float64)
        st = (Status * B)()
        eps = np.empty(B)
        check(lib().nlsg_nm_minimize(self._h, x.ctypes.data_as(_capi.pd),
                                     up.ctypes.data_as(_capi.pd) if up is not None else None,
                                     lo.ctypes.data_as(_capi.pd) if lo is not None else None,
                                     st, eps.ctypes.data_as(_capi.pd)))
        return x, list(st), eps

    def phase_cycles(self, x0):
        """One solve with the kernel's phase counters on: (batch, 8) shader-clock cycles per phase
        (scan, centroid, reflection, expansion / contraction, shrink, -, iterations, shrinks)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        out = np.zeros((self.cfg.batch, 8), dtype=np.uint64)
        check(lib().nlsg_nm_phase_cycles(self._h, x0.ctypes.data_as(_capi.pd),
                                         out.ctypes.data_as(_capi.pu)))
        return out

    def time_solve(self, x0, repeats=1):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        ms = C.c_float()
        check(lib().nlsg_nm_time_solve(self._h, x0.ctypes.data_as(_capi.pd), repeats, C.byref(ms)))
        return ms.value


class NelderMead:
    """Drop-in for nlsolver::NelderMead on a device objective (same ctor args/defaults).
    reference_order=None: reference order (NMEngine) wherever the reference's arithmetic exists on the
    device — Rosenbrock / Sphere / Styblinski-Tang, a custom objective given by its terms —: the
    reference's runs bit for bit. True / False force it."""

    def __init__(self, f, step=-1.0, alpha=1.0, gamma=2.0, rho=0.5, sigma=0.5, eps=1e-6,
                 max_iter=500, no_change_best_tol=20, restarts=0, *, device=0, reference_order=None):
        from .de import CustomObjective
        if reference_order is None:
            reference_order = (isinstance(f, str) and f in ("rosenbrock", "sphere", "styblinski_tang")) or \
                (isinstance(f, CustomObjective) and f.chain != 2)
        self.reference_order = bool(reference_order)
        self.f = f
        self.eps = eps  # mutated by every solve like the reference's member (nlsolver.h:2189)
        self.args = dict(step=step, alpha=alpha, gamma=gamma, rho=rho, sigma=sigma,
                         max_iter=max_iter, no_change_best_tol=no_change_best_tol,
                         restarts=restarts, device=device)

    def _solve(self, x, upper, lower, minimize):
        if not isinstance(x, np.ndarray) or x.dtype != np.float64 or x.ndim not in (1, 2):
            raise TypeError("x must be a float64 numpy array of shape (n,) or (batch, n)")
        xb = x.reshape(1, -1) if x.ndim == 1 else x
        bounded = upper is not None
        with NMEngine(self.f, xb.shape[0], xb.shape[1], minimize=minimize, bounded=bounded,
                      eps=self.eps, reference_order=self.reference_order, **self.args) as eng:
            out, st, eps = eng.minimize(xb, upper, lower)
        xb[...] = out
        if x.ndim == 1:
            self.eps = float(eps[0])
            return st[0]
        return st

    def minimize(self, x, upper=None, lower=None):
        return self._solve(x, upper, lower, True)

    def maximize(self, x, upper=None, lower=None):
        return self._solve(x, upper, lower, False)
