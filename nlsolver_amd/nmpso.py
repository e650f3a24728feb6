"""Host-side mirror of the reference's NelderMeadPSO class for device objectives, batched.

Reference interface (nlsolver.h:3546-3620):
    NelderMeadPSO<Callable, RNG, scalar_t>(f, generator, alpha = 1, gamma = 2, rho = 0.5,
        sigma = 0.5, inertia = 0.8, cognitive_coef = 1.8, social_coef = 1.8, eps = 1e-6,
        max_iter = 1000, no_change_best_iter = 20)
    minimize(x) / maximize(x) / minimize(x, lower, upper) / maximize(x, lower, upper)
(lower before upper, like PSO). x may be (n,) or (batch, n): independent instances, one GPU
workgroup each. The generator only seeds the instances (draws are keyed on the device).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import NMPSOConfig, Status, check, lib
from .de import DEFAULT_SEED, seed_from_generator


class NMPSOEngine:
    def __init__(self, objective, batch, dim, *, minimize=True, bounded=False, alpha=1.0, gamma=2.0,
                 rho=0.5, sigma=0.5, inertia=0.8, cognitive=1.8, social=1.8, eps=1e-6, max_iter=1000,
                 no_change_best_iter=20, seed=DEFAULT_SEED, inst_lo=0, device=0, stream=None):
        cfg = NMPSOConfig()
        cfg.struct_size = C.sizeof(NMPSOConfig)
        cfg.device = device
        cfg.stream = None if stream is None else (stream or 1)
        from .de import CustomObjective, rtc_library_path
        custom = objective if isinstance(objective, CustomObjective) else None
        cfg.objective = (_capi.OBJ_CUSTOM if custom else
                         _capi.OBJECTIVES[objective] if isinstance(objective, str) else objective)
        cfg.minimize, cfg.bounded = int(bool(minimize)), int(bool(bounded))
        cfg.batch, cfg.dim, cfg.inst_lo = batch, dim, inst_lo
        cfg.alpha, cfg.gamma, cfg.rho, cfg.sigma = alpha, gamma, rho, sigma
        cfg.inertia, cfg.cognitive, cfg.social, cfg.eps = inertia, cognitive, social, eps
        cfg.max_iter, cfg.no_change_best_iter, cfg.seed = max_iter, no_change_best_iter, seed
        self.cfg = cfg
        self._h = C.c_void_p()
        if custom:
            check(lib().nlsg_rtc_load(rtc_library_path().encode()))
            obj = _capi.CustomObjectiveC(custom.term_body.encode(), custom.finish_body.encode(),
                                         int(custom.chain), 0)
            check(lib().nlsg_nmpso_create_custom(C.byref(cfg), C.byref(obj), C.byref(self._h)))
        else:
            check(lib().nlsg_nmpso_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().nlsg_nmpso_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def minimize(self, x, lower=None, upper=None):
        """x: (batch, dim) starts; returns (best particles, [Status])."""
        n = self.cfg.dim
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        assert x.shape == (self.cfg.batch, n)
        lo = up = None
        if self.cfg.bounded:
            lo = np.ascontiguousarray(np.broadcast_to(lower, (n,)), dtype=np.float64)
            up = np.ascontiguousarray(np.broadcast_to(upper, (n,)), dtype=np.float64)
        st = (Status * self.cfg.batch)()
        check(lib().nlsg_nmpso_minimize(self._h, x.ctypes.data_as(_capi.pd),
                                        lo.ctypes.data_as(_capi.pd) if lo is not None else None,
                                        up.ctypes.data_as(_capi.pd) if up is not None else None, st))
        return x, list(st)

    def time_solve(self, x0, repeats=1):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        ms = C.c_float()
        check(lib().nlsg_nmpso_time_solve(self._h, x0.ctypes.data_as(_capi.pd), repeats, C.byref(ms)))
        return ms.value


class NelderMeadPSO:
    """Drop-in for nlsolver::NelderMeadPSO on a device objective (same ctor args and defaults)."""

    def __init__(self, f, generator=None, alpha=1.0, gamma=2.0, rho=0.5, sigma=0.5, inertia=0.8,
                 cognitive_coef=1.8, social_coef=1.8, eps=1e-6, max_iter=1000,
                 no_change_best_iter=20, *, device=0):
        self.f, self.generator = f, generator
        self.args = dict(alpha=alpha, gamma=gamma, rho=rho, sigma=sigma, inertia=inertia,
                         cognitive=cognitive_coef, social=social_coef, eps=eps, max_iter=max_iter,
                         no_change_best_iter=no_change_best_iter, device=device)

    def _solve(self, x, lower, upper, minimize):
        if not isinstance(x, np.ndarray) or x.dtype != np.float64 or x.ndim not in (1, 2):
            raise TypeError("x must be a float64 numpy array of shape (n,) or (batch, n)")
        xb = x.reshape(1, -1) if x.ndim == 1 else x
        if xb.shape[1] < 2:  # nlsolver.h:3627-3637: refused, "some invalid solver state"
            st = Status()
            st.f_value = 999999
            return st if x.ndim == 1 else [st] * xb.shape[0]
        with NMPSOEngine(self.f, xb.shape[0], xb.shape[1], minimize=minimize,
                         bounded=lower is not None, seed=seed_from_generator(self.generator),
                         **self.args) as eng:
            out, st = eng.minimize(xb, lower, upper)
        xb[...] = out
        return st[0] if x.ndim == 1 else st

    def minimize(self, x, lower=None, upper=None):
        return self._solve(x, lower, upper, True)

    def maximize(self, x, lower=None, upper=None):
        return self._solve(x, lower, upper, False)
