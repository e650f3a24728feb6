"""Host-side mirror of the reference's SANN class for device objectives, batched.

Reference interface (nlsolver.h:2744-2775):
    SANN<Callable, RNG, scalar_t>(f, generator, max_iter = 5000, temperature_iter = 10,
        temperature_max = 10.0)
    minimize(x) / maximize(x)
x may be (n,) or (batch, n): independent chains, one per GPU wave. The generator only seeds the
chains: draws are keyed by (seed, chain, step, slot) on the device (oracle: orc_sann_sync).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import SANNConfig, Status, check, lib
from .de import DEFAULT_SEED, seed_from_generator


class SANNEngine:
    def __init__(self, objective, batch, dim, *, minimize=True, max_iter=5000, temperature_iter=10,
                 temperature_max=10.0, seed=DEFAULT_SEED, chain_lo=0, device=0, stream=None):
        cfg = SANNConfig()
        cfg.struct_size = C.sizeof(SANNConfig)
        cfg.device = device
        cfg.stream = None if stream is None else (stream or 1)
        from .de import CustomObjective, rtc_library_path
        custom = objective if isinstance(objective, CustomObjective) else None
        cfg.objective = (_capi.OBJ_CUSTOM if custom else
                         _capi.OBJECTIVES[objective] if isinstance(objective, str) else objective)
        cfg.minimize = int(bool(minimize))
        cfg.batch, cfg.dim, cfg.chain_lo = batch, dim, chain_lo
        cfg.max_iter, cfg.temperature_iter = max_iter, temperature_iter
        cfg.temperature_max, cfg.seed = temperature_max, seed
        self.cfg = cfg
        self._h = C.c_void_p()
        if custom:
            check(lib().nlsg_rtc_load(rtc_library_path().encode()))
            obj = _capi.CustomObjectiveC(custom.term_body.encode(), custom.finish_body.encode(),
                                         int(custom.chain), 0)
            check(lib().nlsg_sann_create_custom(C.byref(cfg), C.byref(obj), C.byref(self._h)))
        else:
            check(lib().nlsg_sann_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().nlsg_sann_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def minimize(self, x):
        """x: (batch, dim) starts; returns (best points, [Status])."""
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        assert x.shape == (self.cfg.batch, self.cfg.dim)
        st = (Status * self.cfg.batch)()
        check(lib().nlsg_sann_minimize(self._h, x.ctypes.data_as(_capi.pd), st))
        return x, list(st)

    def time_solve(self, x0, repeats=1):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        ms = C.c_float()
        check(lib().nlsg_sann_time_solve(self._h, x0.ctypes.data_as(_capi.pd), repeats, C.byref(ms)))
        return ms.value


class SANN:
    """Drop-in for nlsolver::SANN on a device objective (same ctor args and defaults)."""

    def __init__(self, f, generator=None, max_iter=5000, temperature_iter=10, temperature_max=10.0,
                 *, device=0):
        self.f, self.generator = f, generator
        self.args = dict(max_iter=max_iter, temperature_iter=temperature_iter,
                         temperature_max=temperature_max, device=device)

    def _solve(self, x, minimize):
        if not isinstance(x, np.ndarray) or x.dtype != np.float64 or x.ndim not in (1, 2):
            raise TypeError("x must be a float64 numpy array of shape (n,) or (batch, n)")
        xb = x.reshape(1, -1) if x.ndim == 1 else x
        with SANNEngine(self.f, xb.shape[0], xb.shape[1], minimize=minimize,
                        seed=seed_from_generator(self.generator), **self.args) as eng:
            out, st = eng.minimize(xb)
        xb[...] = out
        return st[0] if x.ndim == 1 else st

    def minimize(self, x):
        return self._solve(x, True)

    def maximize(self, x):
        return self._solve(x, False)
