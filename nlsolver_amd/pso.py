"""Host-side mirror of the reference's PSO class for device objectives.

Reference interface (nlsolver.h:2498-2591):
    PSO<Callable, RNG, scalar_t, PSOType>(f, generator, inertia=0.8, cognitive_coef=1.8,
        social_coef=1.8, n_particles=10, max_iter=5000, best_val_no_change=50, eps=10e-4)
    minimize(x) / maximize(x)                       bounds = -+|x_i| (2553-2575)
    minimize(x, lower, upper) / maximize(x, lower, upper)   (note: lower first, 2577-2591)
Same positional arguments and defaults here. All compute runs in libnlsolver_hip.so.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import PSO_ACCELERATED, PSO_VANILLA, PSOConfig, Status, check, lib
from .de import DEFAULT_SEED, seed_from_generator


class PSOEngine:
    """RAII wrapper over the nlsg_pso_* C-ABI."""

    def __init__(self, objective, n_particles, dim, *, type=PSO_VANILLA, bounded=False,
                 minimize=True, inertia=0.8, cognitive=1.8, social=1.8, eps=10e-4, max_iter=5000,
                 best_val_no_change=50, seed=DEFAULT_SEED, device=0, stream=None, shard_lo=0,
                 shard_n=None):
        cfg = PSOConfig()
        cfg.struct_size = C.sizeof(PSOConfig)
        cfg.device = device
        cfg.stream = None if stream is None else (stream or 1)
        from .de import CustomObjective, rtc_library_path
        custom = objective if isinstance(objective, CustomObjective) else None
        cfg.objective = (_capi.OBJ_CUSTOM if custom else
                         _capi.OBJECTIVES[objective] if isinstance(objective, str) else objective)
        cfg.minimize, cfg.type, cfg.bounded = int(bool(minimize)), type, int(bool(bounded))
        cfg.n_particles, cfg.dim = n_particles, dim
        cfg.shard_lo = shard_lo
        cfg.shard_n = n_particles if shard_n is None else shard_n
        cfg.inertia, cfg.cognitive, cfg.social, cfg.eps = inertia, cognitive, social, eps
        cfg.max_iter, cfg.best_val_no_change, cfg.seed = max_iter, best_val_no_change, seed
        self.cfg = cfg
        self._h = C.c_void_p()
        if custom:
            check(lib().nlsg_rtc_load(rtc_library_path().encode()))
            obj = _capi.CustomObjectiveC(custom.term_body.encode(), custom.finish_body.encode(),
                                         int(custom.chain), 0)
            check(lib().nlsg_pso_create_custom(C.byref(cfg), C.byref(obj), C.byref(self._h)))
        else:
            check(lib().nlsg_pso_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().nlsg_pso_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _bounds(self, lower, upper):
        D = self.cfg.dim
        lo = np.ascontiguousarray(np.broadcast_to(lower, (D,)), dtype=np.float64)
        hi = np.ascontiguousarray(np.broadcast_to(upper, (D,)), dtype=np.float64)
        return lo, hi

    def init(self, lower, upper):
        lo, hi = self._bounds(lower, upper)
        check(lib().nlsg_pso_init(self._h, lo.ctypes.data_as(_capi.pd), hi.ctypes.data_as(_capi.pd)))

    def step(self, turns=1):
        check(lib().nlsg_pso_step(self._h, turns))

    def status(self):
        st = Status()
        check(lib().nlsg_pso_status(self._h, C.byref(st)))
        return st

    def best(self):
        x = np.empty(self.cfg.dim)
        f, idx = C.c_double(), C.c_uint64()
        check(lib().nlsg_pso_best(self._h, x.ctypes.data_as(_capi.pd), C.byref(f), C.byref(idx)))
        return x, f.value, idx.value

    def download(self):
        n, D = self.cfg.shard_n, self.cfg.dim
        vanilla = self.cfg.type == PSO_VANILLA
        pos, vel = np.empty((n, D)), (np.empty((n, D)) if vanilla else None)
        pbest, cur = np.empty(n), np.empty(n)
        check(lib().nlsg_pso_download(self._h, pos.ctypes.data_as(_capi.pd),
                                      vel.ctypes.data_as(_capi.pd) if vanilla else None,
                                      pbest.ctypes.data_as(_capi.pd), cur.ctypes.data_as(_capi.pd)))
        return pos, vel, pbest, cur

    def minimize(self, x, lower, upper, poll_every=0):
        lo, hi = self._bounds(lower, upper)
        st = Status()
        check(lib().nlsg_pso_minimize(self._h, x.ctypes.data_as(_capi.pd),
                                      lo.ctypes.data_as(_capi.pd), hi.ctypes.data_as(_capi.pd),
                                      poll_every, C.byref(st)))
        return st

    def time_move_kernel(self, launches):
        ms = C.c_float()
        check(lib().nlsg_pso_time_move_kernel(self._h, launches, C.byref(ms)))
        return ms.value

    def record_doubles(self):
        return lib().nlsg_pso_record_doubles(self._h)

    def turn_begin(self, send_dev_ptr):
        check(lib().nlsg_pso_turn_begin(self._h, send_dev_ptr))

    def turn_end(self, gathered_dev_ptr, world):
        check(lib().nlsg_pso_turn_end(self._h, gathered_dev_ptr, world))

    def comm_attach(self, unique_id, world, rank):
        """Collective: joins the library-side RCCL communicator (see nlsolver_amd.dist)."""
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        check(lib().nlsg_pso_comm_attach(self._h, buf, world, rank))

    def step_sharded(self, turns=1):
        check(lib().nlsg_pso_step_sharded(self._h, turns))

    def comm_ranks(self):
        """(world, rank) as the attached RCCL communicator reports them."""
        w, r = C.c_int32(), C.c_int32()
        check(lib().nlsg_pso_comm_ranks(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value


class PSO:
    """Drop-in for nlsolver::PSO on a device objective (same ctor args/defaults/overloads)."""

    def __init__(self, f, generator=None, inertia=0.8, cognitive_coef=1.8, social_coef=1.8,
                 n_particles=10, max_iter=5000, best_val_no_change=50, eps=10e-4, *,
                 type=PSO_VANILLA, device=0):
        self.f, self.generator, self.n_particles = f, generator, n_particles
        self.args = dict(inertia=inertia, cognitive=cognitive_coef, social=social_coef, eps=eps,
                         max_iter=max_iter, best_val_no_change=best_val_no_change, type=type,
                         device=device)

    def _solve(self, x, lower, upper, minimize):
        if not isinstance(x, np.ndarray) or x.dtype != np.float64 or x.ndim != 1:
            raise TypeError("x must be a 1-D float64 numpy array (updated in place)")
        bounded = lower is not None
        if not bounded:  # nlsolver.h:2553-2560: lower = -|x|, upper = |x|
            lower, upper = -np.abs(x), np.abs(x)
        seed = seed_from_generator(self.generator)
        with PSOEngine(self.f, self.n_particles, x.size, bounded=bounded, minimize=minimize,
                       seed=seed, **self.args) as eng:
            return eng.minimize(x, lower, upper)

    def minimize(self, x, lower=None, upper=None):
        return self._solve(x, lower, upper, True)

    def maximize(self, x, lower=None, upper=None):
        return self._solve(x, lower, upper, False)


PSOSolver = PSO  # README.md:99 alias
