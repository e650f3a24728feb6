"""Host-side mirror of the reference's LevenbergMarquardt class, batched, for NLLS models
whose Gauss-Newton functors live on the device, and for built-in objectives minimised through the
reference's default functors (finite-difference gradient and Hessian, nlsolver.h:3494-3511).

Reference interface (nlsolver.h:3428-3463):
    LevenbergMarquardt<Callable, scalar_t, Grad, Hess>(f, lambda = 10, upward_mult = 10,
        downward_mult = 10, max_iter = 100, f_delta = 1e-12, g, h).minimize(x)
Here `f` is a TanhRegression model (f = sum r^2, Grad = 2 J^T r, Hess = 2 J^T J evaluated by the
kernels); minimize() takes one start (n,) or a batch (batch, n) — BASELINE config 4.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import LMConfig, Status, check, lib


class TanhRegression:
    """r_i(theta) = y_i - tanh(sum_j A_ij theta_j); A: (batch, m, n), y: (batch, m)."""
    nlsg_nlls_objective = _capi.OBJ_TANH_REGRESSION

    def __init__(self, A, y):
        self.A = np.ascontiguousarray(A, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        if self.A.ndim == 2:
            self.A, self.y = self.A[None], self.y[None]
        assert self.A.ndim == 3 and self.y.shape == self.A.shape[:2]

    def __call__(self, theta, problem=0):
        r = self.y[problem] - np.tanh(self.A[problem] @ np.asarray(theta, dtype=np.float64))
        return float(r @ r)


class LMEngine:
    """model: a TanhRegression, or the name / id of a built-in objective ("rosenbrock", "sphere",
    "styblinski_tang") or a CustomObjective, with batch= and n= (default functors: fin_diff +
    fin_diff_h)."""

    def __init__(self, model, *, lam=10.0, up=10.0, down=10.0, max_iter=100, f_delta=1e-12,
                 solver=_capi.LM_CHOLESKY, device=0, stream=None, batch=None, n=None):
        from .de import CustomObjective, rtc_library_path
        custom = model if isinstance(model, CustomObjective) else None
        fd = custom is not None or isinstance(model, (str, int))
        if fd:
            if batch is None or n is None:
                raise TypeError("an objective needs batch= and n=")
            B, m = batch, 0
            objective = (_capi.OBJ_CUSTOM if custom else
                         _capi.OBJECTIVES[model] if isinstance(model, str) else model)
        else:
            B, m, n = model.A.shape
            objective = model.nlsg_nlls_objective
        cfg = LMConfig()
        cfg.struct_size = C.sizeof(LMConfig)
        cfg.device = device
        cfg.stream = None if stream is None else (stream or 1)
        cfg.objective, cfg.solver = objective, solver
        cfg.batch, cfg.m, cfg.n = B, m, n
        cfg.lambda_, cfg.up, cfg.down, cfg.max_iter, cfg.f_delta = lam, up, down, max_iter, f_delta
        self.cfg = cfg
        self._h = C.c_void_p()
        if custom:
            check(lib().nlsg_rtc_load(rtc_library_path().encode()))
            obj = _capi.CustomObjectiveC(custom.term_body.encode(), custom.finish_body.encode(),
                                         int(custom.chain), 0)
            check(lib().nlsg_lm_create_custom(C.byref(cfg), C.byref(obj), C.byref(self._h)))
        else:
            check(lib().nlsg_lm_create(C.byref(cfg), C.byref(self._h)))
        if not fd:
            check(lib().nlsg_lm_set_data(self._h, model.A.ctypes.data_as(_capi.pd),
                                         model.y.ctypes.data_as(_capi.pd)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().nlsg_lm_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_solver(self, solver):
        """LM_CHOLESKY / LM_QR for the next solves; the model data stays on the device."""
        check(lib().nlsg_lm_set_solver(self._h, solver))
        self.cfg.solver = solver

    def minimize(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        assert theta.shape == (self.cfg.batch, self.cfg.n)
        st = (Status * self.cfg.batch)()
        lam = np.empty(self.cfg.batch)
        check(lib().nlsg_lm_minimize(self._h, theta.ctypes.data_as(_capi.pd), st,
                                     lam.ctypes.data_as(_capi.pd)))
        return theta, list(st), lam

    def time_solve(self, theta0, repeats=1):
        theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        ms = C.c_float()
        check(lib().nlsg_lm_time_solve(self._h, theta0.ctypes.data_as(_capi.pd), repeats, C.byref(ms)))
        return ms.value

    def time_eval_kernel(self, theta0, repeats=1):
        """Total ms of `repeats` launches of the evaluation kernel (f, g, H at theta0)."""
        theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        ms = C.c_float()
        check(lib().nlsg_lm_time_eval_kernel(self._h, theta0.ctypes.data_as(_capi.pd), repeats,
                                             C.byref(ms)))
        return ms.value


    def time_qr_kernel(self, theta0, repeats=1):
        """Total ms of `repeats` launches of the QR step kernel (after one evaluation at theta0)."""
        theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        ms = C.c_float()
        check(lib().nlsg_lm_time_qr_kernel(self._h, theta0.ctypes.data_as(_capi.pd), repeats,
                                           C.byref(ms)))
        return ms.value


class LevenbergMarquardt:
    """Drop-in for nlsolver::LevenbergMarquardt on a device NLLS model or a built-in objective
    (by name); x: (n,) or (batch, n). solver=None (as include/nlsolver_mi/nlsolver.h's
    device::summation() default): the default-functor model on Rosenbrock / Sphere / Styblinski-Tang solves
    in reference order (LM_CHOLESKY_REFERENCE_ORDER: the reference's run bit for bit, and — a probe
    per lane — the faster evaluation); everything else with LM_CHOLESKY."""

    def __init__(self, f, lam=10.0, upward_mult=10.0, downward_mult=10.0, max_iter=100,
                 f_delta=1e-12, g=None, h=None, *, solver=None, device=0):
        if g is not None or h is not None:
            raise TypeError("device models carry their functors (Gauss-Newton for NLLS models, "
                            "the reference's finite-difference defaults for objectives)")
        self.f = f
        self.args = dict(lam=lam, up=upward_mult, down=downward_mult, max_iter=max_iter,
                         f_delta=f_delta, solver=solver, device=device)

    def minimize(self, x):
        if not isinstance(x, np.ndarray) or x.dtype != np.float64 or x.ndim not in (1, 2):
            raise TypeError("x must be a float64 numpy array of shape (n,) or (batch, n)")
        xb = x.reshape(1, -1) if x.ndim == 1 else x
        shape = {} if hasattr(self.f, "A") else dict(batch=xb.shape[0], n=xb.shape[1])
        args = dict(self.args)
        if args["solver"] is None:
            from .de import CustomObjective
            has_it = (isinstance(self.f, str) and self.f in ("rosenbrock", "sphere", "styblinski_tang")) or \
                (isinstance(self.f, CustomObjective) and self.f.chain != 2)  # (given by its terms)
            ref = has_it
            args["solver"] = _capi.LM_CHOLESKY_REFERENCE_ORDER if ref else _capi.LM_CHOLESKY
        with LMEngine(self.f, **args, **shape) as eng:
            out, st, lam = eng.minimize(xb)
        xb[...] = out
        self.lambdas = lam  # the reference keeps lambda as a member across calls (:3436)
        return st[0] if x.ndim == 1 else st

    def maximize(self, x):  # nlsolver.h:3468 static_assert(minimize, ...)
        raise NotImplementedError("LevenbergMarquardt currently only supports minimization")
