"""Timing data points of the paths past the one-wave sizes (run with PYTHONPATH=.): LM with more
than 64 parameters (Gauss-Newton functors, default functors) and the Nelder-Mead / PSO hybrid
past 128 coordinates."""
import time

import numpy as np
import torch

import nlsolver_amd as m


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r


rng = np.random.default_rng(1)
for (mm, n, B, iters) in ((512, 128, 1024, 10), (1024, 256, 256, 6)):
    A = (2 * rng.random((B, mm, n)) - 1) / np.sqrt(n)
    star = 2 * rng.random((B, n)) - 1
    y = np.tanh(np.einsum("bmn,bn->bm", A, star))
    th0 = 0.5 * star + 0.05 * (2 * rng.random((B, n)) - 1)
    with m.LMEngine(m.TanhRegression(A, y), lam=10.0, max_iter=iters, f_delta=0.0) as eng:
        th, st, lam = eng.minimize(th0.copy())
        eng.time_solve(th0, 2)  # (clocks up)
        dt = eng.time_solve(th0, 3) / 3 * 1e-3  # HIP events around the whole solve, data resident
        ev = eng.time_eval_kernel(th0, 10) / 10
    print(f"LM tanh m={mm} n={n} batch={B}: {dt / iters * 1e3:8.3f} ms per iteration (evaluation kernel "
          f"{ev:.3f} ms), {B * iters / dt:.3e} iteration-problems/s, max f {max(s.f_value for s in st):.2e}",
          flush=True)
for (obj, n, B, iters) in (("rosenbrock", 100, 1, 2), ("rosenbrock", 100, 64, 2), ("sphere", 256, 8, 1)):
    x0 = 0.5 + 0.3 * (rng.random((B, n)) - 0.5)
    with m.lm.LMEngine(obj, batch=B, n=n, lam=1.0, max_iter=iters, f_delta=0.0) as eng:
        eng.minimize(x0.copy())
        dt, (x, st, lam) = timed(lambda: eng.minimize(x0.copy()))
    evals = B * (iters + 1) * (1 + 4 * n + 16 * n * n)
    print(f"LM default functors {obj} n={n} batch={B}: {dt / iters * 1e3:8.2f} ms per iteration, "
          f"{evals / dt:.3e} objective evaluations/s", flush=True)
for (n, B, iters) in ((256, 64, 20), (1024, 8, 5)):
    x0 = 0.5 + (rng.random((B, n)) - 0.5)
    with m.NMPSOEngine("rosenbrock", B, n, max_iter=iters, eps=0.0, no_change_best_iter=10**6, seed=3) as eng:
        eng.minimize(x0)
        dt, (x, st) = timed(lambda: eng.minimize(x0))
    print(f"NelderMeadPSO rosenbrock n={n} instances={B}: {dt / iters * 1e3:8.3f} ms per iteration, "
          f"{sum(s.function_calls_used for s in st) / dt:.3e} objective evaluations/s", flush=True)
