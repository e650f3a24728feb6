"""What the reference-order (bit-for-bit) modes cost against the tree-order kernels: default-gradient
BFGS on Rosenbrock-128D x 4096 starts, BFGS on the configs[2] quadratic (n 1024 x 512 starts), and
default-functor LM on Rosenbrock-16D x 4096. PYTHONPATH=. python scripts/ref_order_cost.py"""
import sys
import time

import numpy as np

import nlsolver_amd
from nlsolver_amd import _capi


def timed(make, x0, reps=3):
    with make() as eng:
        eng.minimize(x0.copy())
        t0 = time.perf_counter()
        for _ in range(reps):
            out = eng.minimize(x0.copy())
        return (time.perf_counter() - t0) / reps, out


rng = np.random.default_rng(1)
x0 = 0.8 + 0.4 * (rng.random((4096, 128)) - 0.5)
if len(sys.argv) > 1 and sys.argv[1] == "profile-bfgs-fd":  # (under rocprofv3: reference order only)
    timed(lambda: nlsolver_amd.BFGSEngine("rosenbrock", 4096, dim=128, max_iter=20, grad_eps=0.0,
                                          reference_order=True), x0)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "profile-bfgs-quad":
    nq = 1024
    dq = np.array([1.0 + 9.0 * i / (nq - 1) for i in range(nq)])
    bq = np.sin(0.1 * np.arange(nq))
    xq0 = 1.0 + 0.5 * (rng.random((512, nq)) - 0.5)
    timed(lambda: nlsolver_amd.BFGSEngine(nlsolver_amd.QuadDiagRank1(dq, bq, 0.01), 512, max_iter=20,
                                          grad_eps=0.0, reference_order=True), xq0)
    sys.exit(0)
for ref in (False, True):
    dt, out = timed(lambda: nlsolver_amd.BFGSEngine("rosenbrock", 4096, dim=128, max_iter=20, grad_eps=0.0,
                                                    reference_order=ref), x0)
    print(f"bfgs-fd rosenbrock-128 x 4096, 20 iterations, reference_order={ref}: {dt * 1e3:.1f} ms")
n = 1024
d = np.array([1.0 + 9.0 * i / (n - 1) for i in range(n)])
b = np.sin(0.1 * np.arange(n))
xq = 1.0 + 0.5 * (rng.random((512, n)) - 0.5)
for ref in (False, True):
    dt, out = timed(lambda: nlsolver_amd.BFGSEngine(nlsolver_amd.QuadDiagRank1(d, b, 0.01), 512, max_iter=20,
                                                    grad_eps=0.0, reference_order=ref), xq)
    print(f"bfgs quadratic n=1024 x 512, 20 iterations, reference_order={ref}: {dt * 1e3:.1f} ms")
xl = 0.9 + 0.2 * (rng.random((4096, 16)) - 0.5)
for solver, name in ((_capi.LM_CHOLESKY, "tree"), (_capi.LM_CHOLESKY_REFERENCE_ORDER, "reference")):
    dt, out = timed(lambda: nlsolver_amd.lm.LMEngine("rosenbrock", batch=4096, n=16, lam=10.0, max_iter=10,
                                                     f_delta=0.0, solver=solver), xl)
    print(f"lm-fd rosenbrock-16 x 4096, 10 iterations, {name} order: {dt * 1e3:.1f} ms")

# ONE start (what minimize() of the drop-in classes solves in reference order by default): a warm engine's
# whole solve, to its stop
print("one start:")
x1 = np.full((1, n), 1.0)
for ref in (False, True):
    dt, (xo, st) = timed(lambda: nlsolver_amd.BFGSEngine(nlsolver_amd.QuadDiagRank1(d, b, 0.01), 1, max_iter=100,
                                                         grad_eps=1e-10, reference_order=ref), x1, reps=5)
    print(f"  bfgs quadratic n=1024, {st[0].iteration} iterations, reference_order={ref}: {dt * 1e3:.2f} ms")
x1 = (0.9 + 0.0005 * np.arange(128)).reshape(1, -1)
for ref in (False, True):
    dt, (xo, st) = timed(lambda: nlsolver_amd.BFGSEngine("rosenbrock", 1, dim=128, max_iter=20, grad_eps=0.0,
                                                         reference_order=ref), x1, reps=5)
    print(f"  bfgs-fd rosenbrock-128, {st[0].iteration} iterations, reference_order={ref}: {dt * 1e3:.2f} ms")
for nn, iters in ((16, 6), (100, 2)):
    x1 = (0.95 + 0.0005 * np.arange(nn)).reshape(1, -1)
    for solver, name in ((_capi.LM_CHOLESKY, "tree"), (_capi.LM_CHOLESKY_REFERENCE_ORDER, "reference")):
        dt, out = timed(lambda: nlsolver_amd.lm.LMEngine("rosenbrock", batch=1, n=nn, lam=10.0, max_iter=iters,
                                                         f_delta=0.0, solver=solver), x1, reps=5)
        print(f"  lm-fd rosenbrock-{nn}, {iters} iterations, {name} order: {dt * 1e3:.2f} ms")

# past 64 parameters, a batch
xw = 0.95 + 0.1 * (rng.random((256, 128)) - 0.5)
for solver, name in ((_capi.LM_CHOLESKY, "tree"), (_capi.LM_CHOLESKY_REFERENCE_ORDER, "reference")):
    dt, out = timed(lambda: nlsolver_amd.lm.LMEngine("rosenbrock", batch=256, n=128, lam=10.0, max_iter=2,
                                                     f_delta=0.0, solver=solver), xw)
    print(f"lm-fd rosenbrock-128 x 256, 2 iterations, {name} order: {dt * 1e3:.1f} ms")
