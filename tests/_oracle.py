"""ctypes binding of oracle/liboracle.so — the CHECKER (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module. The product package (nlsolver_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")

u64 = C.c_uint64
f64 = C.c_double
sz = C.c_size_t
pd = C.POINTER(C.c_double)
pu = C.POINTER(C.c_uint64)


class XorShift(C.Structure):
    _fields_ = [("x", u64 * 2)]


class EvalLog(C.Structure):
    _fields_ = [("xs", pd), ("fs", pd), ("capacity", sz), ("count", sz), ("D", sz)]


class Status(C.Structure):
    _fields_ = [("f_value", f64), ("iteration", u64), ("function_calls_used", u64),
                ("gradient_evals_used", u64), ("hessian_evals_used", u64)]


class DESync(C.Structure):
    _fields_ = [("obj", C.c_int), ("minimize", C.c_int), ("strategy", C.c_int),
                ("pop", sz), ("D", sz), ("n_shards", sz),
                ("CR", f64), ("F", f64), ("eps", f64),
                ("max_iter", sz), ("best_val_no_change", sz), ("seed", u64),
                ("cur", pd), ("nxt", pd), ("scores", pd),
                ("best_id", u64), ("iter", u64), ("val_no_change", u64), ("fcalls", u64),
                ("done", C.c_int), ("std_err", f64), ("trace", pu)]


class PSOSync(C.Structure):
    _fields_ = [("obj", C.c_int), ("minimize", C.c_int), ("type", C.c_int), ("bounded", C.c_int),
                ("n", sz), ("D", sz), ("n_shards", sz),
                ("inertia0", f64), ("cog", f64), ("soc", f64), ("eps", f64),
                ("max_iter", sz), ("best_val_no_change", sz), ("seed", u64),
                ("lower", pd), ("upper", pd),
                ("pos", pd), ("vel", pd), ("pbest_pos", pd), ("pbest_val", pd), ("cur_val", pd),
                ("gbest_x", pd), ("gbest_val", f64), ("gbest_idx", u64),
                ("iter", u64), ("val_no_change", u64), ("fevals", u64),
                ("done", C.c_int), ("std_err", f64), ("inertia", f64)]


class Quad(C.Structure):
    _fields_ = [("d", pd), ("b", pd), ("c", f64)]


class BfgsCounters(C.Structure):
    _fields_ = [("f_calls", u64), ("g_calls", u64), ("f_log", pd), ("f_cap", sz), ("f_count", sz),
                ("H_out", pd)]


class Nlls(C.Structure):
    _fields_ = [("kind", C.c_int), ("m", sz), ("n", sz), ("A", pd), ("y", pd), ("t", pd)]


def _ptr(a):
    return a.ctypes.data_as(pd)


def tanh_problem(lib, seed, problem, m, n):
    A, y, th0 = np.zeros((m, n)), np.zeros(m), np.zeros(n)
    lib.orc_lm_make_tanh_problem(seed, problem, m, n, _ptr(A), _ptr(y), _ptr(th0))
    return A, y, th0


def lm_solve(lib, A, y, x0, *, kind=1, t=None, lam=10.0, up=10.0, down=10.0, max_iter=100,
             f_delta=1e-12, solver=0, order=0):
    """Oracle LM on one problem; returns (status, x, final lambda, f_log)."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    y = np.ascontiguousarray(y, dtype=np.float64)
    A = np.ascontiguousarray(A, dtype=np.float64) if A is not None else np.zeros(1)
    tt = np.ascontiguousarray(t, dtype=np.float64) if t is not None else np.zeros(1)
    q = Nlls(kind, y.size, x.size, _ptr(A), _ptr(y), _ptr(tt))
    lam_c = C.c_double(lam)
    flog = np.zeros(max_iter + 2)
    st = lib.orc_lm_solve(C.byref(q), _ptr(x), C.cast(C.byref(lam_c), pd), up, down, max_iter,
                          f_delta, solver, order, _ptr(flog), flog.size)
    return st, x, lam_c.value, flog[:st.function_calls_used]


def quad_problem(n, c=0.01):
    """The G6 quadratic's parameters: d_i = 1 + 9 i/(n-1), b_i = sin(0.1 i)."""
    import math
    d = np.array([1.0 + 9.0 * i / (n - 1) if n > 1 else 1.0 for i in range(n)])
    b = np.array([math.sin(0.1 * i) for i in range(n)])
    return d, b, c


def bfgs_quad(lib, x0, *, max_iter=100, grad_eps=5e-3, alpha=1.0, tree=0, log=False, c=0.01,
              hessian=None):
    """Run the oracle's BFGS on the G6 quadratic; returns (status, x, f_log). `hessian`: an
    (n, n) array that receives the inverse Hessian after the last update."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    n = x.size
    d, b, c = quad_problem(n, c)
    q = Quad(_ptr(d), _ptr(b), c)
    cnt = BfgsCounters()
    flog = np.zeros(22 * (max_iter + 1) + 4) if log else None
    if log:
        cnt.f_log, cnt.f_cap = _ptr(flog), flog.size
    if hessian is not None:
        assert hessian.shape == (n, n) and hessian.flags.c_contiguous
        cnt.H_out = _ptr(hessian)
    st = lib.orc_bfgs_quad(C.byref(q), _ptr(x), n, max_iter, grad_eps, alpha, tree, C.byref(cnt))
    return st, x, (flog[:cnt.f_count] if log else None)


def bfgs_fd(lib, obj, x0, *, max_iter=100, grad_eps=5e-3, alpha=1.0, tree=0, log_cap=0):
    """Oracle BFGS on a built-in objective with the default finite-difference gradient;
    returns (status, x, f_log of at most log_cap values or None, number of objective calls)."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    cnt = BfgsCounters()
    flog = np.zeros(log_cap) if log_cap else None
    if log_cap:
        cnt.f_log, cnt.f_cap = _ptr(flog), flog.size
    st = lib.orc_bfgs_fd(OBJ[obj], _ptr(x), x.size, max_iter, grad_eps, alpha, tree, C.byref(cnt))
    return st, x, (flog[:min(cnt.f_count, log_cap)] if log_cap else None), cnt.f_count


def lm_fd(lib, obj, x0, *, lam=10.0, up=10.0, down=10.0, max_iter=100, f_delta=1e-12, order=0,
          log_cap=0):
    """Oracle LevenbergMarquardt with the default finite-difference functors on a built-in
    objective; returns (status, x, lambda_after, f_log or None)."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    lam_c = C.c_double(lam)
    flog = np.zeros(log_cap) if log_cap else None
    st = lib.orc_lm_fd(OBJ[obj], _ptr(x), x.size, C.byref(lam_c), up, down, max_iter, f_delta, order,
                       _ptr(flog) if log_cap else None, log_cap)
    return st, x, lam_c.value, flog


def sann_serial(lib, obj, x0, *, minimize=True, max_iter=5000, temp_iter=10, temp_max=10.0,
                log_cap=0):
    """Oracle SANN in the reference's arithmetic and draw order (fresh xorshift generator);
    returns (status, x, next draw of the generator, f_log or None)."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    gen = XorShift()
    lib.orc_xorshift_init(C.byref(gen))
    flog = np.zeros(log_cap) if log_cap else None
    st = lib.orc_sann_serial(OBJ[obj], int(minimize), _ptr(x), x.size, C.byref(gen), max_iter,
                             temp_iter, temp_max, _ptr(flog) if log_cap else None, log_cap)
    return st, x, lib.orc_xorshift_next(C.byref(gen)), flog


def sann_sync(lib, obj, x0, seed, chain, *, minimize=True, max_iter=5000, temp_iter=10,
              temp_max=10.0, log_cap=0):
    """Oracle SANN as the GPU runs it (counter-keyed draws, deterministic math, objective tree)."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    flog = np.zeros(log_cap) if log_cap else None
    st = lib.orc_sann_sync(OBJ[obj], int(minimize), _ptr(x), x.size, seed, chain, max_iter,
                           temp_iter, temp_max, _ptr(flog) if log_cap else None, log_cap)
    return st, x, flog


def _hyb_args(x, upper, lower, kw):
    n = x.size
    bound = upper is not None
    up = np.ascontiguousarray(np.broadcast_to(upper, (n,)), dtype=np.float64) if bound else None
    lo = np.ascontiguousarray(np.broadcast_to(lower, (n,)), dtype=np.float64) if bound else None
    coefs = [kw.get(k, d) for k, d in (("alpha", 1.0), ("gamma", 2.0), ("rho", 0.5), ("sigma", 0.5),
                                       ("inertia", 0.8), ("cog", 1.8), ("soc", 1.8))]
    tail = [kw.get("eps", 1e-6), kw.get("max_iter", 1000), kw.get("no_change", 20)]
    return bound, up, lo, coefs, tail


def nmpso_serial(lib, obj, x0, *, minimize=True, upper=None, lower=None, log_cap=0, **kw):
    """Oracle NelderMeadPSO in the reference's arithmetic and draw order (fresh xorshift);
    returns (status, x, next draw, f_log or None)."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    bound, up, lo, coefs, tail = _hyb_args(x, upper, lower, kw)
    gen = XorShift()
    lib.orc_xorshift_init(C.byref(gen))
    flog = np.zeros(log_cap) if log_cap else None
    st = lib.orc_nmpso_serial(OBJ[obj], int(minimize), int(bound), _ptr(x), x.size,
                              _ptr(up) if bound else None, _ptr(lo) if bound else None,
                              C.byref(gen), *coefs, *tail, _ptr(flog) if log_cap else None, log_cap)
    return st, x, lib.orc_xorshift_next(C.byref(gen)), flog


def nmpso_sync(lib, obj, x0, seed, instance, *, minimize=True, upper=None, lower=None, log_cap=0,
               **kw):
    """Oracle NelderMeadPSO as the GPU runs it (keyed draws, objective / std_err trees)."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    bound, up, lo, coefs, tail = _hyb_args(x, upper, lower, kw)
    flog = np.zeros(log_cap) if log_cap else None
    st = lib.orc_nmpso_sync(OBJ[obj], int(minimize), int(bound), _ptr(x), x.size,
                            _ptr(up) if bound else None, _ptr(lo) if bound else None, seed, instance,
                            *coefs, *tail, _ptr(flog) if log_cap else None, log_cap)
    return st, x, flog


def load():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    lib = C.CDLL(LIB)
    lib.orc_splitmix_next.restype = u64
    lib.orc_splitmix_next.argtypes = [pu]
    lib.orc_xorshift_init.argtypes = [C.POINTER(XorShift)]
    lib.orc_xorshift_next.restype = f64
    lib.orc_xorshift_next.argtypes = [C.POINTER(XorShift)]
    lib.orc_mix64.restype = u64
    lib.orc_mix64.argtypes = [u64]
    lib.orc_ctr_key.restype = u64
    lib.orc_ctr_key.argtypes = [u64, u64]
    lib.orc_u01.restype = f64
    lib.orc_u01.argtypes = [u64]
    for name in ("orc_objective_seq", "orc_objective_tree"):
        fn = getattr(lib, name)
        fn.restype = f64
        fn.argtypes = [C.c_int, pd, sz]
    for name in ("orc_std_err_serial", "orc_std_err_tree", "orc_block_tree_sum", "orc_tiled_sum"):
        fn = getattr(lib, name)
        fn.restype = f64
        fn.argtypes = [pd, sz]
    lib.orc_de_serial.restype = Status
    lib.orc_de_serial.argtypes = [C.c_int, C.c_int, C.c_int, pd, sz, C.POINTER(XorShift),
                                  f64, f64, f64, sz, sz, sz, C.POINTER(EvalLog)]
    lib.orc_de_sync_init.argtypes = [C.POINTER(DESync), pd]
    lib.orc_de_sync_step.argtypes = [C.POINTER(DESync)]
    lib.orc_de_sync_step_omp.argtypes = [C.POINTER(DESync), C.c_int]
    lib.orc_de_shard_record.argtypes = [C.POINTER(DESync), sz, sz, pd]
    lib.orc_de_apply_records.restype = C.c_int
    lib.orc_de_apply_records.argtypes = [C.POINTER(DESync), pd, C.c_int, pd]
    lib.orc_de_shard_generation.argtypes = [C.POINTER(DESync), sz, sz, C.c_int]
    lib.orc_de_commit.argtypes = [C.POINTER(DESync)]
    for name in ("orc_log", "orc_cos"):
        fn = getattr(lib, name)
        fn.restype = f64
        fn.argtypes = [f64]
    lib.orc_pso_serial.restype = Status
    lib.orc_pso_serial.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, pd, sz, pd, pd,
                                   C.POINTER(XorShift), f64, f64, f64, sz, sz, sz, f64,
                                   C.POINTER(EvalLog)]
    lib.orc_pso_sync_init.argtypes = [C.POINTER(PSOSync)]
    lib.orc_pso_sync_step.argtypes = [C.POINTER(PSOSync), C.c_int]
    lib.orc_pso_shard_record.argtypes = [C.POINTER(PSOSync), sz, sz, pd]
    lib.orc_pso_apply_records.restype = C.c_int
    lib.orc_pso_apply_records.argtypes = [C.POINTER(PSOSync), pd, C.c_int]
    lib.orc_pso_shard_move.argtypes = [C.POINTER(PSOSync), sz, sz, C.c_int]
    lib.orc_pso_commit.argtypes = [C.POINTER(PSOSync)]
    lib.orc_bfgs_quad.restype = Status
    lib.orc_bfgs_quad.argtypes = [C.POINTER(Quad), pd, sz, sz, f64, f64, C.c_int,
                                  C.POINTER(BfgsCounters)]
    lib.orc_nmpso_last_shrinks.restype = sz
    lib.orc_nmpso_last_shrinks.argtypes = []
    lib.orc_nmpso_serial.restype = Status
    lib.orc_nmpso_serial.argtypes = [C.c_int, C.c_int, C.c_int, pd, sz, pd, pd, C.POINTER(XorShift)] + \
        [f64] * 8 + [sz, sz, pd, sz]
    lib.orc_nmpso_sync.restype = Status
    lib.orc_nmpso_sync.argtypes = [C.c_int, C.c_int, C.c_int, pd, sz, pd, pd, u64, u64] + \
        [f64] * 8 + [sz, sz, pd, sz]
    lib.orc_sann_serial.restype = Status
    lib.orc_sann_serial.argtypes = [C.c_int, C.c_int, pd, sz, C.POINTER(XorShift), sz, sz, f64, pd, sz]
    lib.orc_sann_sync.restype = Status
    lib.orc_sann_sync.argtypes = [C.c_int, C.c_int, pd, sz, u64, u64, sz, sz, f64, pd, sz]
    lib.orc_lm_fd.restype = Status
    lib.orc_lm_fd.argtypes = [C.c_int, pd, sz, C.POINTER(C.c_double), f64, f64, sz, f64, C.c_int, pd,
                              sz]
    lib.orc_bfgs_fd.restype = Status
    lib.orc_bfgs_fd.argtypes = [C.c_int, pd, sz, sz, f64, f64, C.c_int, C.POINTER(BfgsCounters)]
    lib.orc_update_inverse_hessian.argtypes = [pd, pd, pd, pd, f64, sz, C.c_int]
    for name in ("orc_exp", "orc_tanh"):
        fn = getattr(lib, name)
        fn.restype = f64
        fn.argtypes = [f64]
    lib.orc_rnorm.restype = f64
    lib.orc_rnorm.argtypes = [u64]
    lib.orc_probe_math.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), sz]
    lib.orc_cholesky.argtypes = [pd, sz]
    lib.orc_update_with_hessian.argtypes = [pd, pd, pd, sz]
    lib.orc_qr_decomposition.argtypes = [pd, sz, sz, f64, pd, pd]
    lib.orc_tinyqr_lm.argtypes = [pd, pd, sz, sz, pd]
    lib.orc_tinyqr_lm_order.argtypes = [pd, pd, sz, sz, pd, C.c_int]
    lib.orc_tinyqr_lm_tol.argtypes = [pd, pd, sz, sz, pd, C.c_int, f64]
    lib.orc_lm_make_tanh_problem.argtypes = [u64, u64, sz, sz, pd, pd, pd]
    lib.orc_nm_run.restype = Status
    lib.orc_nm_run.argtypes = [C.c_int, C.c_int, C.c_int, pd, sz, pd, pd, f64, f64, f64, f64, f64,
                               pd, sz, sz, sz, C.c_int, C.POINTER(EvalLog)]
    lib.orc_lm_solve.restype = Status
    lib.orc_lm_solve.argtypes = [C.POINTER(Nlls), pd, pd, f64, f64, sz, f64, C.c_int, C.c_int, pd, sz]
    return lib


OBJ = {"rosenbrock": 0, "sphere": 1, "styblinski_tang": 2, "rastrigin": 3}


class DESyncRun:
    """Owns the numpy buffers of one synchronous-DE oracle run."""

    def __init__(self, lib, obj, pop, D, x0, *, minimize=True, strategy=1, n_shards=1,
                 CR=0.9, F=0.8, eps=0.0, max_iter=1000, best_val_no_change=50,
                 seed=12374563468, trace=False):
        self.lib = lib
        self.pop, self.D = pop, D
        self.bufs = [np.zeros((pop, D)), np.zeros((pop, D))]
        self.scores = np.zeros(pop)
        self.trace = np.zeros((pop, 5), dtype=np.uint64) if trace else None
        s = DESync()
        s.obj, s.minimize, s.strategy = OBJ[obj] if isinstance(obj, str) else obj, int(minimize), strategy
        s.pop, s.D, s.n_shards = pop, D, n_shards
        s.CR, s.F, s.eps = CR, F, eps
        s.max_iter, s.best_val_no_change, s.seed = max_iter, best_val_no_change, seed
        s.cur, s.nxt, s.scores = _ptr(self.bufs[0]), _ptr(self.bufs[1]), _ptr(self.scores)
        s.trace = self.trace.ctypes.data_as(pu) if trace else None
        self.s = s
        self.x0 = np.ascontiguousarray(x0, dtype=np.float64)
        lib.orc_de_sync_init(C.byref(s), _ptr(self.x0))

    def step(self, n=1, threads=0):
        for _ in range(n):
            if threads:
                self.lib.orc_de_sync_step_omp(C.byref(self.s), threads)
            else:
                self.lib.orc_de_sync_step(C.byref(self.s))

    @property
    def population(self):
        addr = C.addressof(self.s.cur.contents)
        for b in self.bufs:
            if b.ctypes.data == addr:
                return b
        raise RuntimeError("cur pointer lost")

    @property
    def best_x(self):
        return self.population[self.s.best_id].copy()


class OracleShardEngine:
    """CPU stand-in with the DEEngine interface used by nlsolver_amd.dist.ShardedDE
    (record_doubles / init / turn_begin / turn_end on host tensors). It owns a full-size
    oracle state but only ever computes the rows of its own shard — exactly what one
    rank does; used by the gloo tests to exercise the host-side exchange logic."""

    def __init__(self, lib, obj, pop, D, lo, n, **kw):
        self.lib, self.lo, self.n, self.D, self.pop = lib, lo, n, D, pop
        self.kw = dict(kw, n_shards=pop // n)
        self.obj = obj
        self.run = None

    def record_doubles(self):
        return self.D + 5

    def init(self, x0):
        self.run = DESyncRun(self.lib, self.obj, self.pop, self.D, x0, **self.kw)
        # forget every row this rank does not own
        mask = np.ones(self.pop, bool)
        mask[self.lo:self.lo + self.n] = False
        self.run.population[mask] = np.nan
        self.run.scores[mask] = np.nan

    def turn_begin(self, send_ptr):
        rec = np.ctypeslib.as_array(C.cast(send_ptr, pd), (self.D + 5,))
        if not self.run.s.done:
            self.lib.orc_de_shard_record(C.byref(self.run.s), self.lo, self.n, _ptr(rec))

    # the product's engine offers the turn in three pieces so that the generation can be
    # launched speculatively (strategy random); this stand-in mirrors that contract
    def can_speculate(self):
        return self.kw.get("strategy", 1) == 1

    def turn_generation(self):
        """Generation k+1 from the current state WITHOUT adopting it (non-destructive)."""
        s = self.run.s
        self._spec = None
        if s.done:
            return
        keep_scores = self.run.scores.copy()
        self.lib.orc_de_shard_generation(C.byref(s), self.lo, self.n, 1)
        self._spec = (keep_scores, self.run.scores.copy())
        self.run.scores[:] = keep_scores  # the finaliser must still see generation k

    def turn_finalize(self, gathered_ptr, world):
        s = self.run.s
        if s.done:
            return
        best_x = np.zeros(self.D)
        done = self.lib.orc_de_apply_records(C.byref(s), C.cast(gathered_ptr, pd), world, _ptr(best_x))
        if done:
            return  # the speculative generation is never adopted
        if self._spec is not None:
            self.run.scores[:] = self._spec[1]
            self.lib.orc_de_commit(C.byref(s))

    def turn_end(self, gathered_ptr, world):
        s = self.run.s
        if s.done:
            return
        best_x = np.zeros(self.D)
        done = self.lib.orc_de_apply_records(C.byref(s), C.cast(gathered_ptr, pd), world, _ptr(best_x))
        if done:
            self.best_x_cache = best_x
            return
        # strategy best reads the row of best_id: install the exchanged copy
        keep = None
        if not (self.lo <= s.best_id < self.lo + self.n):
            keep = self.run.population[s.best_id].copy()
            self.run.population[s.best_id] = best_x
        self.lib.orc_de_shard_generation(C.byref(s), self.lo, self.n, 1)
        if keep is not None:
            self.run.population[s.best_id] = keep
        self.lib.orc_de_commit(C.byref(s))

    def shard(self):
        return (self.run.population[self.lo:self.lo + self.n].copy(),
                self.run.scores[self.lo:self.lo + self.n].copy())


PSO_VANILLA, PSO_ACCELERATED = 0, 1


class PSOSyncRun:
    """Owns the numpy buffers of one synchronous-PSO oracle run."""

    def __init__(self, lib, obj, n, D, lower, upper, *, type=PSO_ACCELERATED, bounded=False,
                 minimize=True, n_shards=1, inertia=0.8, cog=1.8, soc=1.8, eps=0.0, max_iter=5000,
                 best_val_no_change=50, seed=12374563468):
        self.lib, self.n, self.D = lib, n, D
        self.lower = np.ascontiguousarray(np.broadcast_to(lower, (D,)), dtype=np.float64)
        self.upper = np.ascontiguousarray(np.broadcast_to(upper, (D,)), dtype=np.float64)
        self.pos = np.zeros((n, D))
        self.vel = np.zeros((n, D))
        self.pbest_pos = np.zeros((n, D))
        self.pbest_val = np.zeros(n)
        self.cur_val = np.zeros(n)
        self.gbest_x = np.zeros(D)
        s = PSOSync()
        s.obj = OBJ[obj] if isinstance(obj, str) else obj
        s.minimize, s.type, s.bounded = int(minimize), type, int(bounded)
        s.n, s.D, s.n_shards = n, D, n_shards
        s.inertia0, s.cog, s.soc, s.eps = inertia, cog, soc, eps
        s.max_iter, s.best_val_no_change, s.seed = max_iter, best_val_no_change, seed
        s.lower, s.upper = _ptr(self.lower), _ptr(self.upper)
        s.pos, s.vel, s.pbest_pos = _ptr(self.pos), _ptr(self.vel), _ptr(self.pbest_pos)
        s.pbest_val, s.cur_val, s.gbest_x = _ptr(self.pbest_val), _ptr(self.cur_val), _ptr(self.gbest_x)
        self.s = s
        lib.orc_pso_sync_init(C.byref(s))

    def step(self, n=1, threads=1):
        for _ in range(n):
            self.lib.orc_pso_sync_step(C.byref(self.s), threads)


def nm_run(lib, x0, *, obj="rosenbrock", minimize=True, upper=None, lower=None, step=-1.0,
           alpha=1.0, gamma=2.0, rho=0.5, sigma=0.5, eps=1e-6, max_iter=500, no_change=20,
           restarts=0, order=0, log_cap=0):
    """Oracle NelderMead minimize()/maximize(); returns (status, x, eps_after, log or None)."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    n = x.size
    bound = upper is not None
    up = np.ascontiguousarray(np.broadcast_to(upper if bound else 0.0, (n,)), dtype=np.float64)
    lo = np.ascontiguousarray(np.broadcast_to(lower if bound else 0.0, (n,)), dtype=np.float64)
    eps_c = C.c_double(eps)
    lg, lx, lf = None, None, None
    if log_cap:
        lx, lf = np.zeros((log_cap, n)), np.zeros(log_cap)
        lg = EvalLog(_ptr(lx), _ptr(lf), log_cap, 0, n)
    st = lib.orc_nm_run(OBJ[obj], int(minimize), int(bound), _ptr(x), n, _ptr(up), _ptr(lo), step,
                        alpha, gamma, rho, sigma, C.cast(C.byref(eps_c), pd), max_iter, no_change,
                        restarts, order, C.byref(lg) if lg else None)
    return st, x, eps_c.value, ((lx[:lg.count], lf[:lg.count]) if lg else None)


class OraclePSOShardEngine:
    """CPU stand-in for nlsolver_amd.PSOEngine on one shard (gloo tests)."""

    def __init__(self, lib, obj, n, D, lo, m, **kw):
        self.lib, self.lo, self.m, self.D, self.n = lib, lo, m, D, n
        self.obj, self.kw = obj, dict(kw, n_shards=n // m)

    def record_doubles(self):
        return self.D + 5

    def init(self, lower, upper):
        self.run = PSOSyncRun(self.lib, self.obj, self.n, self.D, lower, upper, **self.kw)
        mask = np.ones(self.n, bool)
        mask[self.lo:self.lo + self.m] = False
        self.run.pos[mask] = np.nan  # this rank never looks at rows it does not own
        self.run.cur_val[mask] = np.nan
        self.run.pbest_val[mask] = np.nan

    def turn_begin(self, send_ptr):
        rec = np.ctypeslib.as_array(C.cast(send_ptr, pd), (self.D + 5,))
        if not self.run.s.done:
            self.lib.orc_pso_shard_record(C.byref(self.run.s), self.lo, self.m, _ptr(rec))

    def turn_end(self, gathered_ptr, world):
        s = self.run.s
        if s.done:
            return
        if self.lib.orc_pso_apply_records(C.byref(s), C.cast(gathered_ptr, pd), world):
            return
        self.lib.orc_pso_shard_move(C.byref(s), self.lo, self.m, 1)
        self.lib.orc_pso_commit(C.byref(s))


PROBE = {"log": 0, "cos": 1, "exp": 2, "tanh": 3, "cos_2pi": 4, "u01": 5, "rnorm": 6}


def probe_math(lib, fn, bits):
    """The oracle's math primitive `fn` on an array of uint64 bit patterns (cf. nlsg_probe_math)."""
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    out = np.empty_like(bits)
    pu = C.POINTER(C.c_uint64)
    lib.orc_probe_math(PROBE[fn], bits.ctypes.data_as(pu), out.ctypes.data_as(pu), bits.size)
    return out
