"""One-off soak of the reference-order kernels against the serial oracle (order / tree 0, the restatement
pinned to the reference's runs): many sizes around the tile, chunk and wave boundaries, random starts,
every objective with reference arithmetic. Prints one line per solver; exits non-zero on a mismatch.
PYTHONPATH=. python scripts/ref_order_soak.py"""
import sys

import numpy as np

import nlsolver_amd
from nlsolver_amd._capi import LM_CHOLESKY_REFERENCE_ORDER
from tests import _oracle as O

oracle = O.load()
rng = np.random.default_rng(20261005)
objs = ["rosenbrock", "sphere", "styblinski_tang"]
bad = 0


def same(a, b):
    return a == b or (np.isnan(a) and np.isnan(b))


sizes = [1, 2, 3, 5, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 96, 127, 128, 129, 130, 191, 255, 256, 257, 300, 511, 513]
n_cases = 0
for n in sizes:
    obj = objs[n % 3]
    kw = dict(max_iter=3 if n > 128 else 6, grad_eps=0.0, alpha=1.0)
    x0 = 0.8 + 0.4 * (rng.random((2, n)) - 0.5)
    with nlsolver_amd.BFGSEngine(obj, 2, dim=n, reference_order=True, **kw) as eng:
        x, st = eng.minimize(x0.copy())
    for p in range(2):
        ref, xr, _, _ = O.bfgs_fd(oracle, obj, x0[p], tree=0, **kw)
        ok = (st[p].iteration, st[p].function_calls_used) == (ref.iteration, ref.function_calls_used) and \
            same(st[p].f_value, ref.f_value) and np.array_equal(x[p], xr, equal_nan=True)
        n_cases += 1
        if not ok:
            bad += 1
            print("BFGS-fd mismatch", obj, n, p)
print(f"bfgs default gradient: {n_cases} solves checked")

n_cases = 0
for n in [1, 2, 7, 16, 17, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 500, 1008, 1024]:
    kw = dict(max_iter=8, grad_eps=0.0, alpha=1.0)
    d, b, c = O.quad_problem(n)
    x0 = 1.0 + 0.5 * (rng.random((2, n)) - 0.5)
    with nlsolver_amd.BFGSEngine(nlsolver_amd.QuadDiagRank1(d, b, c), 2, reference_order=True, **kw) as eng:
        x, st = eng.minimize(x0.copy())
    for p in range(2):
        ref, xr, _ = O.bfgs_quad(oracle, x0[p], tree=0, **kw)
        ok = (st[p].iteration, st[p].function_calls_used) == (ref.iteration, ref.function_calls_used) and \
            same(st[p].f_value, ref.f_value) and np.array_equal(x[p], xr, equal_nan=True)
        n_cases += 1
        if not ok:
            bad += 1
            print("BFGS quadratic mismatch", n, p)
print(f"bfgs quadratic: {n_cases} solves checked")

n_cases = 0
for n in [1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 66, 100, 127, 128, 129, 130, 200]:
    obj = objs[n % 3]
    kw = dict(lam=10.0, max_iter=2 if n > 32 else 4, f_delta=0.0)
    x0 = 0.9 + 0.2 * (rng.random((2, n)) - 0.5)
    with nlsolver_amd.lm.LMEngine(obj, batch=2, n=n, solver=LM_CHOLESKY_REFERENCE_ORDER, **kw) as eng:
        x, st, lam = eng.minimize(x0.copy())
    for p in range(2):
        ref, xr, lam_r, _ = O.lm_fd(oracle, obj, x0[p], order=0, **kw)
        ok = (st[p].iteration, st[p].function_calls_used) == (ref.iteration, ref.function_calls_used) and \
            same(st[p].f_value, ref.f_value) and np.array_equal(x[p], xr, equal_nan=True) and same(lam[p], lam_r)
        n_cases += 1
        if not ok:
            bad += 1
            print("LM mismatch", obj, n, p)
print(f"lm default functors: {n_cases} solves checked")

n_cases = 0
for n in [1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 126, 127, 128, 129, 130, 200, 256, 257]:
    obj = objs[n % 3]
    kw = dict(step=-1.0, eps=0.0, max_iter=150 if n <= 32 else 80, no_change_best_tol=10**6, restarts=0)
    x0 = 0.5 + 1.0 * (rng.random((2, n)) - 0.5)
    with nlsolver_amd.NMEngine(obj, 2, n, reference_order=True, **kw) as eng:
        x, st, _ = eng.minimize(x0.copy())
    for p in range(2):
        ref, xr, _, _ = O.nm_run(oracle, x0[p], obj=obj, order=0, step=-1.0, eps=0.0, max_iter=kw["max_iter"],
                                 no_change=10**6, restarts=0)
        ok = (st[p].iteration, st[p].function_calls_used) == (ref.iteration, ref.function_calls_used) and \
            same(st[p].f_value, ref.f_value) and np.array_equal(x[p], xr, equal_nan=True)
        n_cases += 1
        if not ok:
            bad += 1
            print("NM mismatch", obj, n, p)
print(f"nelder-mead: {n_cases} solves checked")
print("mismatches:", bad)
sys.exit(1 if bad else 0)
