"""Host-side mirror of the reference's DE class for device objectives.

Reference interface (nlsolver.h:2379-2411):
    DE<Callable, RNG, scalar_t, RecombinationStrategy>(f, generator, CR=0.9, F=0.8,
        eps=10e-4, pop_size=50, max_iter=1000, best_val_no_change=50)
    solver_status minimize(std::vector<T>& x) / maximize(std::vector<T>& x)
Same positional arguments, defaults and in/out `x` convention here; `f` is the
name of a built-in device objective (a device kernel cannot call a host functor,
DESIGN.md), `generator` supplies the 64-bit key of the counter-based RNG.
All compute happens in libnlsolver_hip.so on a gfx950 device.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import DE_BEST, DE_RANDOM, DEConfig, Status, check, lib

DEFAULT_SEED = 12374563468  # rng::splitmix seed, nlsolver.h:1265


def seed_from_generator(generator):
    """Two draws of a reference-style generator (T operator()() in [0,1]) -> u64 key."""
    if generator is None:
        return DEFAULT_SEED
    if isinstance(generator, int):
        return generator & (2**64 - 1)
    hi = min(int(generator() * 2.0**32), 2**32 - 1)
    lo = min(int(generator() * 2.0**32), 2**32 - 1)
    return (hi << 32) | lo


class CustomObjective:
    """A user objective f(x) = finish(sum_i term(x_i, x_{i+1}), D) given as C++ function bodies and
    compiled for the device when the engine is created (nlsg_custom_objective).

        CustomObjective("double t1 = 1 - xi; double t2 = xn - xi * xi; return t1 * t1 + 100 * t2 * t2;",
                        chain=True)                       # the Rosenbrock chain
        CustomObjective("return fabs(xi) * xi * xi;")     # sum |x_i|^3

    In the bodies: `xi`, `xn` (= x_{i+1}; chain objectives sum over i < D - 1) for `term`; `s`, `D`
    for `finish` (default "return s;").

    vector=True: the whole-vector form (NLSG_CUSTOM_VECTOR) for objectives that are not sums of
    such terms — `term_body` is the body of `double f(const X &x, uint64_t D)` with `x(i)`
    (coordinate i, same index in every lane), `x.size()` and `x.sum(g)` (lane-tree sum of
    g(x_i, i) over the coordinates):

        CustomObjective("double a = x(0) * x(0) + x(1) - 11, b = x(0) + x(1) * x(1) - 7;"
                        " return a * a + b * b;", vector=True)    # Himmelblau"""

    def __init__(self, term_body, *, chain=False, finish_body="return s;", vector=False):
        self.term_body, self.finish_body = term_body, finish_body
        self.chain = 2 if vector else int(bool(chain))  # nlsg_custom_objective.chain


def rtc_library_path():
    """hiprtc of the HIP runtime this process already uses (PyTorch ships its own)."""
    import os
    try:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libhiprtc.so")
        return cand if os.path.exists(cand) else ""
    except ImportError:
        return ""


class DEEngine:
    """Thin RAII wrapper over the nlsg_de_* C-ABI (one handle, one device, one stream)."""

    def __init__(self, objective, pop, dim, *, minimize=True, strategy=DE_RANDOM, CR=0.9, F=0.8,
                 eps=10e-4, max_iter=1000, best_val_no_change=50, seed=DEFAULT_SEED, device=0,
                 stream=None, shard_lo=0, shard_n=None, trace=False):
        cfg = DEConfig()
        cfg.struct_size = C.sizeof(DEConfig)
        cfg.device = device
        # None: the engine creates a private stream. An integer is a hipStream_t handle;
        # 0 is torch's default (null) stream, spelled hipStreamLegacy = 1 for the C-ABI.
        cfg.stream = None if stream is None else (stream or 1)
        custom = objective if isinstance(objective, CustomObjective) else None
        cfg.objective = (_capi.OBJ_CUSTOM if custom else
                         _capi.OBJECTIVES[objective] if isinstance(objective, str) else objective)
        cfg.minimize = int(bool(minimize))
        cfg.strategy = strategy
        cfg.trace = int(bool(trace))
        cfg.pop, cfg.dim = pop, dim
        cfg.shard_lo = shard_lo
        cfg.shard_n = pop if shard_n is None else shard_n
        cfg.CR, cfg.F, cfg.eps = CR, F, eps
        cfg.max_iter, cfg.best_val_no_change, cfg.seed = max_iter, best_val_no_change, seed
        self.cfg = cfg
        self._h = C.c_void_p()
        if custom:
            check(lib().nlsg_rtc_load(rtc_library_path().encode()))
            obj = _capi.CustomObjectiveC(custom.term_body.encode(), custom.finish_body.encode(),
                                         int(custom.chain), 0)
            check(lib().nlsg_de_create_custom(C.byref(cfg), C.byref(obj), C.byref(self._h)))
        else:
            check(lib().nlsg_de_create(C.byref(cfg), C.byref(self._h)))

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().nlsg_de_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- C-ABI calls ------------------------------------------------------
    def init(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        if x0.shape != (self.cfg.dim,):
            raise ValueError(f"x0 must have {self.cfg.dim} entries")
        check(lib().nlsg_de_init(self._h, x0.ctypes.data_as(_capi.pd)))

    def step(self, turns=1):
        check(lib().nlsg_de_step(self._h, turns))

    def status(self):
        st = Status()
        check(lib().nlsg_de_status(self._h, C.byref(st)))
        return st

    def best(self):
        x = np.empty(self.cfg.dim)
        f, idx = C.c_double(), C.c_uint64()
        check(lib().nlsg_de_best(self._h, x.ctypes.data_as(_capi.pd), C.byref(f), C.byref(idx)))
        return x, f.value, idx.value

    def download(self, trace=False):
        n, D = self.cfg.shard_n, self.cfg.dim
        pop, scores = np.empty((n, D)), np.empty(n)
        tr = np.empty((n, 5), dtype=np.uint64) if trace else None
        check(lib().nlsg_de_download(self._h, pop.ctypes.data_as(_capi.pd),
                                     scores.ctypes.data_as(_capi.pd),
                                     tr.ctypes.data_as(_capi.pu) if trace else None))
        return (pop, scores, tr) if trace else (pop, scores)

    def upload(self, pop, scores):
        pop = np.ascontiguousarray(pop, dtype=np.float64)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        assert pop.shape == (self.cfg.shard_n, self.cfg.dim) and scores.shape == (self.cfg.shard_n,)
        check(lib().nlsg_de_upload(self._h, pop.ctypes.data_as(_capi.pd),
                                   scores.ctypes.data_as(_capi.pd)))

    def minimize(self, x, poll_every=0):
        st = Status()
        check(lib().nlsg_de_minimize(self._h, x.ctypes.data_as(_capi.pd), poll_every, C.byref(st)))
        return st

    def time_generation_kernel(self, launches):
        ms = C.c_float()
        check(lib().nlsg_de_time_generation_kernel(self._h, launches, C.byref(ms)))
        return ms.value

    def time_turns(self, turns):
        ms = C.c_float()
        check(lib().nlsg_de_time_turns(self._h, turns, C.byref(ms)))
        return ms.value

    def record_doubles(self):
        return lib().nlsg_de_record_doubles(self._h)

    def turn_begin(self, send_dev_ptr):
        check(lib().nlsg_de_turn_begin(self._h, send_dev_ptr))

    def turn_end(self, gathered_dev_ptr, world):
        check(lib().nlsg_de_turn_end(self._h, gathered_dev_ptr, world))

    def turn_finalize(self, gathered_dev_ptr, world):
        check(lib().nlsg_de_turn_finalize(self._h, gathered_dev_ptr, world))

    def turn_generation(self):
        check(lib().nlsg_de_turn_generation(self._h))

    def can_speculate(self):
        return bool(lib().nlsg_de_can_speculate(self._h))

    def comm_attach(self, unique_id, world, rank):
        """Collective: joins the library-side RCCL communicator (see nlsolver_amd.dist)."""
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        check(lib().nlsg_de_comm_attach(self._h, buf, world, rank))

    def step_sharded(self, turns=1):
        check(lib().nlsg_de_step_sharded(self._h, turns))

    def comm_ranks(self):
        """(world, rank) as the attached RCCL communicator reports them."""
        w, r = C.c_int32(), C.c_int32()
        check(lib().nlsg_de_comm_ranks(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value


class DE:
    """Drop-in for nlsolver::DE on a device objective (same ctor args/defaults)."""

    def __init__(self, f, generator=None, crossover_prob=0.9, differential_weight=0.8, eps=10e-4,
                 pop_size=50, max_iter=1000, best_val_no_change=50, *, strategy=DE_RANDOM,
                 device=0):
        self.f, self.generator = f, generator
        self.args = dict(CR=crossover_prob, F=differential_weight, eps=eps, max_iter=max_iter,
                         best_val_no_change=best_val_no_change, strategy=strategy, device=device)
        self.pop_size = pop_size

    def _solve(self, x, minimize):
        if not isinstance(x, np.ndarray) or x.dtype != np.float64 or x.ndim != 1:
            raise TypeError("x must be a 1-D float64 numpy array (it is updated in place, "
                            "like std::vector<T>& in nlsolver.h:2404)")
        seed = seed_from_generator(self.generator)
        with DEEngine(self.f, self.pop_size, x.size, minimize=minimize, seed=seed,
                      **self.args) as eng:
            return eng.minimize(x)

    def minimize(self, x):
        return self._solve(x, True)

    def maximize(self, x):
        return self._solve(x, False)


DESolver = DE  # README.md:80 alias
