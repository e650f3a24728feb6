"""Host-side mirror of the reference's BFGS class for device objectives, batched.

Reference interface (nlsolver.h:3169-3196):
    BFGS<Callable, scalar_t, Grad>(f, g = fin_diff, max_iter = 100, grad_eps = 5e-3, alpha = 1)
    solver_status minimize(std::vector<T>& x)          (one start per call)
Here `f` is a device objective: QuadDiagRank1 with its analytic gradient, or the name of a
built-in objective ("rosenbrock", "sphere", "styblinski_tang", "rastrigin"), for which the
reference's DEFAULT gradient runs on the device (fin_diff = finite_difference_gradient<.,.,1>,
nlsolver.h:1385-1413: four probes per coordinate, each counted as a function call). minimize()
accepts one start (n,) or a batch of independent starts (batch, n) — BASELINE config 3 — solved
in lock step on the GPU, and returns one status per start.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import BFGSConfig, Status, check, lib


class QuadDiagRank1:
    """f(x) = 1/2 sum d_i x_i^2 + 1/2 c (sum x)^2 - sum b_i x_i with its analytic gradient."""
    nlsg_objective = _capi.OBJ_QUAD_DIAG_RANK1

    def __init__(self, d, b, c):
        self.d = np.ascontiguousarray(d, dtype=np.float64)
        self.b = np.ascontiguousarray(b, dtype=np.float64)
        self.c = float(c)
        assert self.d.shape == self.b.shape and self.d.ndim == 1

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        return 0.5 * np.sum(self.d * x * x) + 0.5 * self.c * np.sum(x) ** 2 - np.sum(self.b * x)


class BFGSEngine:
    """symmetric=True: the rank-2 update restated so that H stays bitwise symmetric and only its
    upper 128 x 128 blocks are kept and streamed (NLSG_BFGS_SYMMETRIC, include/nlsg_c_api.h):
    56 % of the memory and traffic at dim = 1024; results agree with the literal update to rounding.
    reference_order=True: every sum in index order, as the reference's sequential loops take it
    (NLSG_BFGS_REFERENCE_ORDER): the reference's own runs bit for bit, default gradient included —
    with the default gradient also the faster kernels (a probe per lane); with a gradient functor
    1.5 x the tree kernels' time at dim = 1024."""

    def __init__(self, objective, batch, *, dim=None, max_iter=100, grad_eps=5e-3, alpha=1.0,
                 device=0, stream=None, symmetric=False, reference_order=False):
        cfg = BFGSConfig()
        cfg.flags = (_capi.BFGS_SYMMETRIC if symmetric else 0) | \
            (_capi.BFGS_REFERENCE_ORDER if reference_order else 0)
        self.symmetric = bool(symmetric)
        cfg.struct_size = C.sizeof(BFGSConfig)
        cfg.device = device
        cfg.stream = None if stream is None else (stream or 1)
        cfg.batch = batch
        cfg.max_iter, cfg.grad_eps, cfg.alpha = max_iter, grad_eps, alpha
        self._h = C.c_void_p()
        from .de import CustomObjective, rtc_library_path
        if isinstance(objective, CustomObjective):  # user objective + finite-difference gradient
            if dim is None:
                raise TypeError("a custom objective needs dim=")
            cfg.objective, cfg.dim, cfg.quad_c = _capi.OBJ_CUSTOM, dim, 0.0
            self.cfg = cfg
            check(lib().nlsg_rtc_load(rtc_library_path().encode()))
            obj = _capi.CustomObjectiveC(objective.term_body.encode(), objective.finish_body.encode(),
                                         int(objective.chain), 0)
            check(lib().nlsg_bfgs_create_custom(C.byref(cfg), C.byref(obj), C.byref(self._h)))
            return
        if isinstance(objective, str):  # built-in objective + finite-difference gradient
            if dim is None:
                raise TypeError("a built-in objective needs dim=")
            cfg.objective, cfg.dim, cfg.quad_c = _capi.OBJECTIVES[objective], dim, 0.0
            self.cfg = cfg
            check(lib().nlsg_bfgs_create(C.byref(cfg), None, None, C.byref(self._h)))
            return
        cfg.objective = objective.nlsg_objective
        cfg.dim, cfg.quad_c = objective.d.size, objective.c
        self.cfg = cfg
        check(lib().nlsg_bfgs_create(C.byref(cfg), objective.d.ctypes.data_as(_capi.pd),
                                     objective.b.ctypes.data_as(_capi.pd), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().nlsg_bfgs_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.cfg.batch, self.cfg.dim)
        return x

    def init(self, x0):
        check(lib().nlsg_bfgs_init(self._h, self._x(x0).ctypes.data_as(_capi.pd)))

    def step(self, iters=1):
        check(lib().nlsg_bfgs_step(self._h, iters))

    def unfinished(self):
        c = C.c_uint64()
        check(lib().nlsg_bfgs_unfinished(self._h, C.byref(c)))
        return c.value

    def identity_count(self):
        """Unfinished problems whose inverse Hessian is the identity right now (start / reset guard):
        the H passes skip their reads for those."""
        c = C.c_uint64()
        check(lib().nlsg_bfgs_identity_count(self._h, C.byref(c)))
        return c.value

    def download(self):
        B, n = self.cfg.batch, self.cfg.dim
        x = np.empty((B, n))
        st = (Status * B)()
        check(lib().nlsg_bfgs_download(self._h, x.ctypes.data_as(_capi.pd), st))
        return x, list(st)

    def download_state(self, hessian=True):
        B, n = self.cfg.batch, self.cfg.dim
        g = np.empty((B, n))
        H = np.empty((B, n, n)) if hessian else None
        check(lib().nlsg_bfgs_download_state(self._h, g.ctypes.data_as(_capi.pd),
                                             H.ctypes.data_as(_capi.pd) if hessian else None))
        return g, H

    def minimize(self, x):
        x = self._x(x)
        st = (Status * self.cfg.batch)()
        check(lib().nlsg_bfgs_minimize(self._h, x.ctypes.data_as(_capi.pd), st))
        return x, list(st)

    def hessian_bytes_per_iteration(self):
        """Algorithmic HBM bytes of the H passes per iteration and problem: read H (t = H y), read
        and write H (update + next direction) — the whole matrix, or its upper 128 x 128 blocks."""
        n = self.cfg.dim
        if not self.symmetric:
            return 3 * n * n * 8
        nb = (n + 127) // 128
        return 3 * (nb * (nb + 1) // 2) * 128 * 128 * 8

    def time_steps(self, iters):
        total, hess = C.c_float(), C.c_float()
        check(lib().nlsg_bfgs_time_steps(self._h, iters, C.byref(total), C.byref(hess)))
        return total.value, hess.value


class BFGS:
    """Drop-in for nlsolver::BFGS on a device objective; x may be (n,) or (batch, n).
    reference_order=None (as include/nlsolver_mi/nlsolver.h's device::summation() default): reference
    order wherever the reference's arithmetic exists on the device — the quadratic, the default
    gradient on Rosenbrock / Sphere / Styblinski-Tang or a custom objective given by its terms; literal
    update — i.e. the reference's runs bit for bit. With the default gradient those are also the
    faster kernels; the quadratic pays 1.18 x on large batches. True / False force it."""

    def __init__(self, f, g=None, max_iter=100, grad_eps=5e-3, alpha=1.0, *, device=0,
                 symmetric=False, reference_order=None):
        if g is not None:
            raise TypeError("device objectives carry their analytic gradient or use the default "
                            "finite-difference one; pass g=None")
        self.f = f
        self.args = dict(max_iter=max_iter, grad_eps=grad_eps, alpha=alpha, device=device,
                         symmetric=symmetric, reference_order=reference_order)

    def minimize(self, x):
        if not isinstance(x, np.ndarray) or x.dtype != np.float64 or x.ndim not in (1, 2):
            raise TypeError("x must be a float64 numpy array of shape (n,) or (batch, n); "
                            "it is updated in place")
        xb = x.reshape(1, -1) if x.ndim == 1 else x
        from .de import CustomObjective
        extra = dict(dim=xb.shape[1]) if isinstance(self.f, (str, CustomObjective)) else {}
        args = dict(self.args)
        if args["reference_order"] is None:
            fd = (isinstance(self.f, str) and self.f in ("rosenbrock", "sphere", "styblinski_tang")) or \
                (isinstance(self.f, CustomObjective) and self.f.chain != 2)  # (terms: index order is the body's own loop)
            quad = isinstance(self.f, QuadDiagRank1)
            args["reference_order"] = (fd or quad) and not args["symmetric"]
        with BFGSEngine(self.f, xb.shape[0], **extra, **args) as eng:
            out, st = eng.minimize(xb)
        xb[...] = out
        return st[0] if x.ndim == 1 else st

    def maximize(self, x):  # nlsolver.h:3199 static_assert(minimize, ...)
        raise NotImplementedError("BFGS currently only supports minimization (nlsolver.h:3199)")
